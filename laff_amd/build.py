"""Builds liblaff_hip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python -m laff_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'liblaff_hip.so')
SOURCES = ['gemm_nt.hip', 'sim_strip.hip', 'fc_strip.hip', 'fuse.hip', 'rank.hip', 'loss.hip', 'comm.hip', 'api.hip']
HEADERS = [os.path.join(CSRC, 'kernels.h'), os.path.join(CSRC, 'exact_cos.h'), os.path.join(CSRC, 'strip_util.h'), os.path.join(CSRC, 'wave_reduce.h'), os.path.join(os.path.dirname(HERE), 'include', 'laff_hip.h')]
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++20', '-fPIC', '--offload-arch=' + ARCH, '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash():
    """sha256 over the kernel sources + headers: what a measured artefact (profiles/*_traffic.json) is valid for."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def build_library(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    objs = []
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([cc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([cc, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv))
