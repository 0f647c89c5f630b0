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
        with open(os.path.join(datadir, 'shape.txt')) as fh:
            rows, width = (int(tok) for tok in fh.readline().split())
        with open(os.path.join(datadir, 'id.txt')) as fh:
            body = fh.read().strip()
        # one id per line, or (older feature directories) all ids on one line separated by blanks: the separator that yields
        # `rows` ids is the right one
        for sep in ('\n', ' '):
            ids = body.split(sep)
            if len(ids) == rows:
                break
        else:
            raise AssertionError('%s: id.txt holds %d ids, shape.txt says %d' % (datadir, len(ids), rows))
        self.nr_of_images, self.ndims, self.names = rows, width, ids
        self.name2index = {name: i for i, name in enumerate(ids)}     # a repeated id resolves to its LAST row, as dict(zip()) does
        self.binary_file = os.path.join(datadir, bin_file)
        self._mm = None
        print('[%s] %dx%d instances loaded from %s' % (type(self).__name__, rows, width, datadir))

    def _matrix(self):
        if self._mm is None:
            self._mm = np.memmap(self.binary_file, dtype=np.float32, mode='r', shape=(self.nr_of_images, self.ndims))
        return self._mm

    def _resolve(self, requested, isname=True):
        """De-duplicated (row, id) hits in file order; unknown ids are dropped, out-of-range rows are an AssertionError."""
        wanted = set(requested)
        if isname:
            known = self.name2index
            return sorted((known[n], n) for n in wanted if n in known)          # rows are unique: the id never decides the order
        rows = sorted(wanted)
        if rows and not (0 <= rows[0] and rows[-1] < len(self.names)):
            raise AssertionError('row index outside [0, %d)' % len(self.names))
        return [(r, self.names[r]) for r in rows]

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
