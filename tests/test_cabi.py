"""The C-ABI library loads on a CPU-only host and exports every symbol include/laff_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'laff_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(laff_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for s in ('laff_ctx_create', 'laff_fc_act_bn', 'laff_fuse', 'laff_frame_fuse', 'laff_frame_fuse_grouped_mask', 'laff_pack_rows', 'laff_sim_gemm',
              'laff_rank_count', 'laff_v2t_count', 'laff_v2t_count_exact', 'laff_rank_metrics', 'laff_last_error'):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from laff_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from laff_amd.build import build_library
        build_library(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), 'liblaff_hip.so does not export %s' % s
    assert lib.laff_abi_version() == _lib.ABI_VERSION


def test_binding_covers_every_declared_symbol():
    from laff_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_argument_errors_do_not_need_a_gpu():
    from laff_amd import _lib
    lib = _lib.load()
    n = ctypes.c_size_t()
    assert lib.laff_packed_bytes(10, 512, 3, ctypes.byref(n)) == 0 and n.value == 10 * 512 * 2 * 2
    assert lib.laff_packed_bytes(10, 512, 99, ctypes.byref(n)) == -1
    assert b'precision' in lib.laff_last_error()
    assert lib.laff_fuse(None, None, 1, 1, 1, 4, None, None, None, 0, None, None) == -1
    # the collectives and the exact V2T count refuse null handles before touching RCCL or the GPU
    assert lib.laff_comm_init(None, 0, 1, None, None) == -1 and b'laff_comm_init' in lib.laff_last_error()
    assert lib.laff_allgather_rows(None, None, None, 16) == -1
    assert lib.laff_allreduce_i32_sum(None, None, 4) == -1 and lib.laff_allreduce_f64_max(None, None, 4) == -1
    assert lib.laff_comm_destroy(None) == 0
    assert lib.laff_v2t_count_exact(None, None, 1, 1, 1, None, None, 1, None, None, 1, 4, None, None, None, None, None, 4) == -1


def test_coalesce_batches_concatenates_what_a_loader_yields():
    """model.retrieve() runs each tower once over the concatenated batches of any loader (laff_amd/model/model.py, coalesce_batches)."""
    import numpy as np
    import torch
    from laff_amd.model.model import coalesce_batches
    b1 = {'vis_feat_dict': {'a': torch.ones(2, 3), 'b': np.zeros((2, 1), np.float32)}, 'idxs': [0, 1], 'vis_ids': ('v0', 'v1'),
          'vis_frame_feat_dict': {}, 'extra': None}
    b2 = {'vis_feat_dict': {'a': 2 * torch.ones(1, 3), 'b': np.ones((1, 1), np.float32)}, 'idxs': [2], 'vis_ids': ('v2',),
          'vis_frame_feat_dict': {}, 'extra': None}
    out = coalesce_batches([b1, b2])
    assert out['idxs'] == [0, 1, 2] and out['vis_ids'] == ['v0', 'v1', 'v2'] and out['extra'] is None and out['vis_frame_feat_dict'] == {}
    assert out['vis_feat_dict']['a'].shape == (3, 3) and float(out['vis_feat_dict']['a'][2, 0]) == 2.0
    assert out['vis_feat_dict']['b'].tolist() == [[0.0], [0.0], [1.0]]
    assert coalesce_batches([]) is None
    with pytest.raises(ValueError):
        coalesce_batches([{'a': None}, {'b': None}])
    with pytest.raises(ValueError):
        coalesce_batches([None, torch.ones(1)])
    with pytest.raises(TypeError):
        coalesce_batches([3.0, 4.0])
    with pytest.raises(RuntimeError):
        coalesce_batches([torch.ones(2, 3), torch.ones(2, 4)])          # frame tensors padded to different lengths


def test_cpu_tensors_are_refused():
    import torch
    from laff_amd import ops
    with pytest.raises(RuntimeError, match='no CPU path'):
        ops.fc_act_bn(torch.zeros(4, 8), torch.zeros(16, 8))


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: with the shared library absent the first op raises (checked in a child process)."""
    import subprocess
    import sys
    code = ("import os, sys; os.environ['LAFF_HIP_LIB'] = %r; sys.path.insert(0, %r); "
            "from laff_amd import _lib\n"
            "try:\n    _lib.load()\nexcept RuntimeError as e:\n    assert 'no CPU fallback' in str(e); print('RAISED')\n"
            % (str(tmp_path / 'nope.so'), ROOT))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'RAISED' in out.stdout, out.stderr


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under laff_amd/ may import or execute it (tier rule 3)."""
    import glob
    for f in glob.glob(os.path.join(ROOT, 'laff_amd', '**', '*.py'), recursive=True):
        src = open(f).read()
        assert 'import oracle' not in src and 'from oracle' not in src and 'laff_oracle' not in src, f
    for f in glob.glob(os.path.join(ROOT, 'laff_amd', 'csrc', '*')):
        assert 'oracle' not in open(f).read().lower(), f


def test_fc_strip_kernels_keep_out_of_the_accumulator_registers(tmp_path):
    """fc_strip.hip names all 256 accumulator registers literally (the stationary strip).  hipcc must not have put anything of its own
    there -- neither a parked value (v_accvgpr_* outside the asm statements) nor a spill -- and must not have gone to scratch: compile
    the file with -save-temps and audit the ISA of its kernels (tools/debug/isa_audit.py)."""
    import re
    import subprocess
    import sys
    from laff_amd import build
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'debug'))
    import isa_audit
    src = os.path.join(build.CSRC, 'fc_strip.hip')
    r = subprocess.run([build.hipcc()] + build.FLAGS + ['-save-temps=obj', '-c', src, '-o', str(tmp_path / 'fc_strip.o')],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    asm = [str(tmp_path / f) for f in os.listdir(tmp_path) if f.endswith('gfx950.s')]
    assert len(asm) == 1
    stats = isa_audit.audit(asm[0], 'fc_strip_kernel', quiet=True)
    assert len(stats) == 3                                    # the three epilogue kinds
    for name, st in stats.items():
        assert st['acc_outside'] == 0 and st['scratch'] == 0, (name, st)
        assert st['mfma'] == 12 * 48, (name, st)           # the plain loop's four bodies + two tails of four (fc_strip.hip)
    text = open(asm[0]).read()
    for name in stats:
        meta = text[text.index('.name:           ' + name):]
        assert int(re.search(r'\.vgpr_spill_count: (\d+)', meta).group(1)) == 0
        assert int(re.search(r'\.sgpr_spill_count: (\d+)', meta).group(1)) == 0


def test_compiler_scheduled_kernels_keep_their_loads_in_flight(tmp_path):
    """Round 6 found hipcc waiting for every load by itself in the C++ kernels (a `global_load` + `s_waitcnt vmcnt(0)` per plane chunk in
    the fusion kernel, per pair of row pieces in the exact re-score) and compiling `__shfl_xor` reductions to the LDS crossbar.  The
    sources are written so that it cannot (raw loads into arrays first; lane swaps + DPP rotations): this holds the ISA to it.
      * fuse_reg_kernel<4, 2>: its eight whole-head plane loads (scalar base, `nt`) come with no vector-memory wait between them;
      * rank_resolve_kernel: somewhere eight 16-byte row loads are issued back to back, no spill;
      * their reductions are DPP row rotations (row_ror), not ds_bpermute_b32."""
    import re
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from laff_amd import build

    def asm_of(name):
        out = str(tmp_path / (name + '.s'))
        flags = [f for f in build.FLAGS if f != '-fPIC']
        r = subprocess.run([build.hipcc()] + flags + ['--cuda-device-only', '-S', os.path.join(build.CSRC, name + '.hip'), '-o', out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return open(out).read()

    with ThreadPoolExecutor(max_workers=2) as ex:
        fuse, rank = ex.map(asm_of, ['fuse', 'rank'])

    def body(text, mangled_prefix):
        m = re.search(r'^(%s\S*):' % re.escape(mangled_prefix), text, re.M)
        assert m, mangled_prefix
        rest = text[m.end():]
        return [ln.split(';')[0].strip() for ln in rest[:rest.index('.amdhsa_kernel')].split('\n')]

    f42 = [i for i in body(fuse, '_ZN4laff15fuse_reg_kernelILi4ELi2E') if i and not i.startswith('.')]
    # the longest run of nt plane loads with a scalar base that no `s_waitcnt vmcnt` interrupts
    best = run = 0
    for ins in f42:
        if re.match(r'global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\].* nt', ins):
            run += 1
            best = max(best, run)
        elif ins.startswith('s_waitcnt') and 'vmcnt' in ins:
            run = min(run, int(re.search(r'vmcnt\((\d+)\)', ins).group(1)))      # vmcnt(n): the n youngest requests stay in flight
    assert best >= 8, 'fuse_reg_kernel<4, 2>: at most %d plane loads in flight' % best
    assert sum(i.startswith('ds_bpermute') for i in f42) <= 1              # (the several-heads ticket broadcast of the rank side)
    assert sum('row_ror' in i for i in f42) >= 16                          # the reductions' DPP rotations
    res = [i for i in body(rank, '_ZN4laff19rank_resolve_kernel') if i and not i.startswith('.')]
    best = run = 0
    for ins in res:
        if ins.startswith('global_load_dwordx4'):
            run += 1
            best = max(best, run)
        elif ins.startswith('s_waitcnt') and 'vmcnt' in ins:
            run = min(run, int(re.search(r'vmcnt\((\d+)\)', ins).group(1)))
    assert best >= 8, 'rank_resolve_kernel: at most %d row loads issued together' % best
    assert sum(i.startswith('v_mov_b32_dpp') and 'row_ror' in i for i in res) >= 16     # the fp64 group sums travel by DPP
    meta = rank[rank.index('.name:           _ZN4laff19rank_resolve_kernel'):]
    assert int(re.search(r'\.vgpr_spill_count: (\d+)', meta).group(1)) == 0
    meta = fuse[fuse.index('.name:           _ZN4laff15fuse_reg_kernelILi4ELi2E'):]
    assert int(re.search(r'\.vgpr_spill_count: (\d+)', meta).group(1)) == 0


def _build_c_host(out_dir):
    """tests/c_host/laff_host.c: plain C11 against include/laff_hip.h, built with gcc (no hipcc, no torch) and linked to the library."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, 'tests', 'c_host', 'laff_host.c')
    exe = os.path.join(str(out_dir), 'laff_host')
    libdir = os.path.join(root, 'laff_amd', 'lib')
    cmd = ['gcc', '-std=c11', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(root, 'include'), src,
           '-o', exe, '-L' + libdir, '-llaff_hip', '-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_c_host_builds_against_the_header(tmp_path):
    """The C ABI is bindable from plain C: the host program of tests/c_host compiles with gcc -std=c11 -Werror against the header alone and
    links to liblaff_hip.so (it runs in the GPU suite: test_gpu_kernels.py::test_c_host_runs_the_exact_rank_tail)."""
    from laff_amd import build
    build.build_library(verbose=False)
    exe = _build_c_host(tmp_path)
    assert os.path.exists(exe)
