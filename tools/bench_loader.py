#!/usr/bin/env python3
"""Row f-1 measurement (host side, no GPU): reading 10,000 x 512 float32 features out of a BigFile directory
  (a) the way the reference's VisionDataset does it -- one `BigFile.read_one` per video (open / seek / fromfile / Python list,
      /root/reference/bigfile.py:187-237, data_provider.py:457-479) -- timed on the REAL reference module when /root/reference
      exists (build container only), and
  (b) laff_amd's bulk path: one mmap gather per batch (`BigFile.read_matrix`).

    python tools/bench_loader.py
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laff_amd.bigfile import BigFile  # noqa: E402
from laff_amd.data import write_bigfile  # noqa: E402


def main():
    n, d, bs = 10000, 512, 256
    g = np.random.default_rng(0)
    mat = g.normal(0, 1, (n, d)).astype(np.float32)
    ids = ['video%d' % i for i in range(n)]
    order = g.permutation(n)
    with tempfile.TemporaryDirectory() as tmp:
        write_bigfile(tmp, ids, mat)
        bf = BigFile(tmp)
        t0 = time.perf_counter()
        out = np.empty((n, d), np.float32)
        for s in range(0, n, bs):
            bf.read_matrix([ids[i] for i in order[s:s + bs]], out=out[s:s + bs])
        t_bulk = time.perf_counter() - t0
        assert np.array_equal(out, mat[order])
        print('laff_amd bulk read_matrix, batches of %d : %8.1f ms  (%.2f GB/s)' % (bs, t_bulk * 1e3, n * d * 4 / t_bulk / 1e9))
        t0 = time.perf_counter()
        rows = [bf.read_one(ids[i]) for i in order]
        t_one = time.perf_counter() - t0
        print('laff_amd read_one per video             : %8.1f ms' % (t_one * 1e3))
        if os.path.isdir('/root/reference'):
            sys.path.insert(0, '/root/reference')
            sys.dont_write_bytecode = True
            os.environ.setdefault('HOME', '/tmp')
            import bigfile as ref_bigfile
            rb = ref_bigfile.BigFile(tmp)
            t0 = time.perf_counter()
            rows = [rb.read_one(ids[i]) for i in order]
            t_ref = time.perf_counter() - t0
            assert np.allclose(np.array(rows[:4], np.float32), mat[order[:4]])
            print('reference BigFile.read_one per video    : %8.1f ms  -> bulk path is %.0fx faster' % (t_ref * 1e3, t_ref / t_bulk))


if __name__ == '__main__':
    main()
