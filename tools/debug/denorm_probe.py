"""Does the fp16 MFMA honour denormal operands?  One product of a denormal with 1.0 through the plain similarity GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = torch.device('cuda')
for strip in (os.environ.get('LAFF_STRIP', '0'),):
    N, K = 40000, 512
    t = torch.zeros(N, 1, K, device=dev); v = torch.zeros(10000, 1, K, device=dev)
    t[:, 0, 0] = 2.0 ** -20; t[:, 0, 1] = 2.0 ** -24; t[:, 0, 2] = 3 * 2.0 ** -16
    v[:, 0, 0] = 1.0; v[:, 0, 1] = 1.0; v[:, 0, 2] = 2.0 ** -12
    T = ops.pack_rows(t, False, 1e-13, 'fp16', 1.0); V = ops.pack_rows(v, False, 1e-13, 'fp16', 1.0)
    S = ops.sim_gemm(T, V)
    want = 2.0 ** -20 + 2.0 ** -24 + 3 * 2.0 ** -28
    print('LAFF_STRIP', strip, 'got', S[0, 0].item(), S[-1, -1].item(), 'want', want, 'equal', S[0, 0].item() == want)
