"""CPU stand-ins for the per-rank kernels (HipBackend interface) + a small seeded problem: shared by the gloo tests of
laff_amd/dist.py and by bench.py's LAFF_BENCH_DRYRUN mode (the self-launch path exercised without a GPU)."""
import numpy as np
import torch

from oracle import laff_oracle as O


class Packed:
    def __init__(self, buf, N, K):
        self.buf, self.N, self.K, self.precision, self.prescale = buf, N, K, 'fp16', 1.0


class OracleBackend:
    """CPU stand-in with the HipBackend interface: fp16-rounded operands, fp32 GEMM."""

    def __init__(self, Wt, Wv):
        self.Wt, self.Wv = Wt, Wv

    def embed_text(self, f):
        return torch.from_numpy(O.l2norm(np.tanh(f['x'].numpy() @ self.Wt)))

    def embed_video(self, f):
        return torch.from_numpy(O.l2norm(np.tanh(f['x'].numpy() @ self.Wv)))

    def pack(self, E, layer=None):
        E = E.reshape(E.shape[0], -1)
        h = E.to(torch.float16).contiguous()
        return Packed(h.view(torch.uint8).reshape(-1), E.shape[0], E.shape[1])

    def pack_gathered(self, E):
        return self.pack(E)

    @staticmethod
    def _mat(p):
        return p.buf[:p.N * p.K * 2].view(torch.float16).reshape(p.N, p.K).float()

    def sim(self, T, V, heads):
        return self._mat(T) @ self._mat(V).T

    class State:
        pass

    def prepare(self, Et, Ev, T, V, gt, col0):
        """exact ground-truth scores (float64) of the texts whose video is in this shard, -inf elsewhere"""
        st = self.State()
        st.Et, st.Ev, st.T, st.V, st.gt, st.col0 = Et, Ev, T, V, gt, col0
        st.S64 = torch.from_numpy(O.txt2vis_matrix_f64(Et.reshape(Et.shape[0], -1).numpy(), Ev.reshape(Ev.shape[0], -1).numpy()))
        c = gt.long() - col0
        ok = (c >= 0) & (c < st.S64.shape[1])
        st.s_gt64 = torch.full((st.S64.shape[0],), float('-inf'), dtype=torch.float64)
        st.s_gt64[ok] = st.S64[torch.arange(st.S64.shape[0])[ok], c[ok]]
        return st

    def s_gt_of(self, st):
        return st.s_gt64

    def sim_ranked(self, st, want_scores=True):
        """fp16-operand score block + counts of the EXACT scores above the exact ground-truth score"""
        S = self.sim(st.T, st.V, 1)
        cols = torch.arange(S.shape[1])[None, :] + st.col0
        count = ((st.S64 > st.s_gt64[:, None]) & (cols != st.gt.long()[:, None])).sum(dim=1).to(torch.int32)
        return S, count

    # ---- 'video16' stand-ins (laff_amd.dist.evaluate_sharded_v16): the band is a constant that covers the fp16 operand rounding ----
    BAND = 4e-3

    def v16_rows(self, T):
        return T.buf[:T.N * T.K * 2].view(T.N, T.K * 2)

    def v16_operand(self, rows2d, N, like):
        return Packed(rows2d.reshape(-1)[:N * like.K * 2].clone(), N, like.K)

    def v16_band_video(self, Ev, V):
        return torch.zeros(Ev.shape[0])

    def v16_prepare_text(self, Et, T, Ev_all, gt_local):
        S64 = torch.from_numpy(O.txt2vis_matrix_f64(Et.reshape(Et.shape[0], -1).numpy(), Ev_all.reshape(Ev_all.shape[0], -1).numpy()))
        s = S64[torch.arange(S64.shape[0]), gt_local.long()].contiguous()
        return s, torch.full((Et.shape[0] + 4,), self.BAND / 2, dtype=torch.float32)

    def v16_gemm(self, T_all, V_local, heads, gt, col0, s_gt64, band_t, band_v, want_scores):
        S = self.sim(T_all, V_local, heads)
        st = self.State()
        st.V = V_local
        cols = torch.arange(S.shape[1])[None, :] + col0
        not_gt = cols != gt.long()[:, None]
        band = (band_t[:S.shape[0]].double() * 2)[:, None]
        d = S.double() - s_gt64[:, None]
        st.count = ((d > band) & not_gt).sum(dim=1).to(torch.int32)
        st.listed = torch.nonzero((d.abs() <= band) & not_gt)
        return S, st

    def v16_export(self, st, S, bounds, col0, cap):
        world = bounds.numel() - 1
        out = torch.full((world, cap, 2), -1, dtype=torch.int32)
        fill = torch.zeros(world + 1, dtype=torch.int32)
        b = bounds.long()
        for r, c in st.listed.tolist():
            o = int((b[1:world] <= r).sum())
            k = int(fill[o])
            fill[o] += 1
            if k < cap:
                out[o, k, 0], out[o, k, 1] = r - int(b[o]), c + col0
            else:
                fill[world] = 1
                st.count[0] = -(1 << 26)
        return out, fill

    def v16_resolve(self, Et, Ev_all, s_gt64, count, lst):
        S64 = torch.from_numpy(O.txt2vis_matrix_f64(Et.reshape(Et.shape[0], -1).numpy(), Ev_all.reshape(Ev_all.shape[0], -1).numpy()))
        assert lst[:4].tolist() == [0, 0, (lst.numel() - 4) // 2, 4]
        for r, c in lst[4:].view(-1, 2).tolist():
            if r >= 0 and S64[r, c] > s_gt64[r]:
                count[r] += 1
        return count

    def metrics(self, ranks):
        r = ranks.numpy().astype(np.float64)
        return O.eval_from_positions([[x] for x in r])


def problem(Nt=61, Nv=23, D=32, seed=5):
    g = np.random.default_rng(seed)
    zv = g.normal(0, 1, (Nv, 8)).astype(np.float32)
    gt = (np.arange(Nt) % Nv).astype(np.int32)
    xv = (zv @ g.normal(0, 1, (8, 16)) + 0.3 * g.normal(0, 1, (Nv, 16))).astype(np.float32)
    xt = (zv[gt] @ g.normal(0, 1, (8, 16)) + 0.3 * g.normal(0, 1, (Nt, 16))).astype(np.float32)
    Wt = g.normal(0, 0.3, (16, D)).astype(np.float32)
    Wv = g.normal(0, 0.3, (16, D)).astype(np.float32)
    return xt, xv, gt, Wt, Wv


