"""BigFile feature reader (/root/reference/bigfile.py:13-240): shape.txt + id.txt + feature.bin (float32 rows).

Same constructor, attributes (`names`, `name2index`, `ndims`, `nr_of_images`, `binary_file`) and
read / read_one / shape semantics (de-duplication, file-index order, unknown ids silently dropped, IndexError from
read_one on an unknown id).  Implementation is a memory map instead of per-row seek + array.fromfile, and adds
`read_matrix` (bulk gather -> one float32 matrix, optionally straight into pinned memory for an async H2D copy).
"""
import os

import numpy as np


class BigFile:
    def __init__(self, datadir, bin_file='feature.bin'):
        self.nr_of_images, self.ndims = list(map(int, open(os.path.join(datadir, 'shape.txt')).readline().split()))
        id_file = os.path.join(datadir, 'id.txt')
        self.names = open(id_file, 'r').read().strip().split('\n')
        if len(self.names) != self.nr_of_images:
            self.names = open(id_file, 'r').read().strip().split(' ')
        assert len(self.names) == self.nr_of_images
        self.name2index = dict(zip(self.names, range(self.nr_of_images)))
        self.binary_file = os.path.join(datadir, bin_file)
        self._mm = None
        print('[%s] %dx%d instances loaded from %s' % (self.__class__.__name__, self.nr_of_images, self.ndims, datadir))

    def _matrix(self):
        if self._mm is None:
            self._mm = np.memmap(self.binary_file, dtype=np.float32, mode='r', shape=(self.nr_of_images, self.ndims))
        return self._mm

    def _resolve(self, requested, isname=True):
        requested = set(requested)
        if isname:
            pairs = [(self.name2index[x], x) for x in requested if x in self.name2index]
        else:
            assert min(requested) >= 0
            assert max(requested) < len(self.names)
            pairs = [(x, self.names[x]) for x in requested]
        pairs.sort(key=lambda v: v[0])
        return pairs

    def read(self, requested, isname=True):
        pairs = self._resolve(requested, isname)
        if len(pairs) == 0:
            return [], []
        rows = self._matrix()[[p[0] for p in pairs]]
        return [p[1] for p in pairs], [r.tolist() for r in rows]

    def read_one(self, name):
        renamed, vectors = self.read([name])
        return vectors[0]

    def read_matrix(self, names, out=None):
        """Rows for `names` IN THE GIVEN ORDER (duplicates allowed, KeyError on unknown) as one float32 matrix."""
        idx = np.fromiter((self.name2index[n] for n in names), dtype=np.int64, count=len(names))
        if out is None:
            out = np.empty((len(idx), self.ndims), dtype=np.float32)
        np.take(self._matrix(), idx, axis=0, out=out)
        return out

    def shape(self):
        return [self.nr_of_images, self.ndims]
