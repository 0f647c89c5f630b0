"""Bulk feature loading for the retrieval path.

The reference's VisionDataset / TextDataset (/root/reference/data_provider.py:380-618) do one open/seek/fromfile and a
Python-list round trip PER ITEM PER FEATURE through BigFile.read_one, then collate 64 items at a time.  Once the GPU pass
takes ~1 ms that dominates wall time.  These loaders keep the batch-dict layout `predict()` consumes
(`collate_vision` :38-73, `collate_text` :76-89) and the loader surface it touches (`.dataset.length`, `len(.dataset)`,
`.batch_size`, `len(loader)`), but gather whole batches of rows with one mmap take into pinned host memory followed by
an asynchronous H2D copy.  On-disk format unchanged (txt2bin.py:64-74: float32 rows, id.txt, shape.txt).
"""
import os

import numpy as np
import torch

from .bigfile import BigFile


def write_bigfile(datadir, ids, matrix):
    """feature.bin / id.txt / shape.txt as txt2bin.py writes them (newline separated ids)."""
    matrix = np.ascontiguousarray(matrix, dtype=np.float32)
    assert matrix.ndim == 2 and matrix.shape[0] == len(ids)
    os.makedirs(datadir, exist_ok=True)
    matrix.tofile(os.path.join(datadir, 'feature.bin'))
    with open(os.path.join(datadir, 'id.txt'), 'w') as f:
        f.write('\n'.join(ids) + '\n')
    with open(os.path.join(datadir, 'shape.txt'), 'w') as f:
        f.write('%d %d' % matrix.shape)


class NpyFeatures:
    """Features kept as a numpy file instead of a BigFile directory -- the two forms the reference reads: a pickled
    `{id: vector}` dict (`np.load(path, allow_pickle=True).item()`, trainer.py:144-148) or a plain (N, D) array next to an id list.
    Exposes the part of the BigFile surface the bulk loaders use (`names`, `ndims`, `nr_of_images`, `shape()`, `read_matrix`,
    `read`, `read_one`), so it can stand wherever a BigFile is expected in BulkVisLoader / BulkTxtLoader."""

    def __init__(self, path_or_array, ids=None):
        obj = np.load(path_or_array, allow_pickle=True) if isinstance(path_or_array, str) else path_or_array
        if isinstance(obj, np.ndarray) and obj.dtype == object and obj.shape == ():
            obj = obj.item()
        if isinstance(obj, dict):
            self.names = list(obj.keys())
            self._m = np.ascontiguousarray(np.stack([np.asarray(obj[k], dtype=np.float32).reshape(-1) for k in self.names]))
        else:
            if ids is None:
                raise ValueError('a plain (N, D) array needs the list of ids of its rows')
            self.names = list(ids)
            self._m = np.ascontiguousarray(np.asarray(obj, dtype=np.float32))
            if self._m.ndim != 2 or self._m.shape[0] != len(self.names):
                raise ValueError('array of shape %s does not match %d ids' % (self._m.shape, len(self.names)))
        self.name2index = dict(zip(self.names, range(len(self.names))))
        self.nr_of_images, self.ndims = self._m.shape

    def _matrix(self):
        return self._m

    def shape(self):
        return [self.nr_of_images, self.ndims]

    def read_matrix(self, names, out=None):
        idx = np.fromiter((self.name2index[n] for n in names), dtype=np.int64, count=len(names))
        if out is None:
            out = np.empty((len(idx), self.ndims), dtype=np.float32)
        np.take(self._m, idx, axis=0, out=out)
        return out

    def read(self, requested, isname=True):
        """BigFile.read semantics (bigfile.py:187-213): duplicates removed, unknown ids dropped, rows in storage order."""
        idx = sorted({self.name2index[r] for r in requested if r in self.name2index} if isname else set(requested))
        return [self.names[i] for i in idx], [self._m[i].tolist() for i in idx]

    def read_one(self, name):
        renamed, vectors = self.read([name])
        return vectors[0]


class _Dataset:
    def __init__(self, n, captions=None):
        self.length = n
        self.captions = captions or {}

    def __len__(self):
        return self.length


def _to_device(arr, device, pin):
    t = torch.from_numpy(arr)
    if device is None or torch.device(device).type == 'cpu':
        return t
    if pin:
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


class BulkVisLoader:
    """Video side: yields `collate_vision`-shaped dicts for `batch_size` videos at a time, tensors already on `device`."""

    def __init__(self, vis_feat_files, vis_ids, batch_size=8192, device='cuda', vis_frame_feat_dicts=None, max_frame=200,
                 pin_memory=True):
        self.files = dict(vis_feat_files or {})
        self.vis_ids = list(vis_ids)
        self.batch_size = int(batch_size)
        self.device, self.pin = device, pin_memory
        self.dataset = _Dataset(len(self.vis_ids))
        self.frame_files = dict(vis_frame_feat_dicts or {})
        self.max_frame = max_frame
        self._frames = {}
        for name, bf in self.frame_files.items():      # video id -> frame row indices in frame order (data_provider.py:432-449)
            groups = {}
            for row, fid in enumerate(bf.names):
                vid, num = fid.rsplit('_', 1)
                groups.setdefault(vid, []).append((int(num), row))
            self._frames[name] = {v: [r for _, r in sorted(lst)][:max_frame] for v, lst in groups.items()}

    def __len__(self):
        return (len(self.vis_ids) + self.batch_size - 1) // self.batch_size

    def whole(self):
        """The whole collection as ONE batch dict: model.retrieve() / predict() then run each tower once over the full matrices (one
        grouped FC launch, one fuse launch per side) instead of once per batch."""
        return next(self._batches(max(1, len(self.vis_ids))), None)

    def __iter__(self):
        return self._batches(self.batch_size)

    def _batches(self, batch_size):
        n = len(self.vis_ids)
        for s in range(0, n, batch_size):
            ids = self.vis_ids[s:s + batch_size]
            feats = {name: _to_device(bf.read_matrix(ids), self.device, self.pin) for name, bf in self.files.items()}
            frame_dict = {}
            if self.frame_files:
                first = next(iter(self._frames))
                lens = np.array([len(self._frames[first][v]) for v in ids], dtype=np.int64)
                fmax = int(lens.max())
                mask = (np.arange(fmax)[None, :] < lens[:, None]).astype(np.float32)
                frame_dict['mask_tensor'] = _to_device(mask, self.device, self.pin)
                for name, bf in self.frame_files.items():
                    out = np.zeros((len(ids), fmax, bf.ndims), dtype=np.float32)
                    mm = bf._matrix()
                    for i, v in enumerate(ids):
                        rows = self._frames[name][v]
                        out[i, :len(rows)] = mm[rows]
                    frame_dict[name] = _to_device(out, self.device, self.pin)
            yield {'vis_feat_dict': feats, 'idxs': list(range(s, s + len(ids))), 'vis_ids': tuple(ids),
                   'vis_frame_feat_dict': frame_dict, 'vis_origin_frame_tuple': (None,) * len(ids)}


class BulkTxtLoader:
    """Text side: yields (caption_feat_dict, idxs, cap_ids) like `collate_text`, in caption-file order.

    `text_feat_files`: {caption_feat_dict key (e.g. 'CLIP_encoding', 'bow_encoding'): BigFile keyed by caption id}.
    (The reference sorts each batch by token count for its GRU's packed sequences; with pre-extracted features the
    order is irrelevant and rows stay in file order -- `txt_ids` returned by predict() reflects that.)"""

    def __init__(self, capfile_or_pairs, text_feat_files, batch_size=16384, device='cuda', pin_memory=True):
        if isinstance(capfile_or_pairs, str):
            pairs = []
            with open(capfile_or_pairs) as reader:                 # data_provider.py:548-559
                for line in reader:
                    if line.strip() == '':
                        continue
                    parts = line.strip().split(None, 1)
                    pairs.append((parts[0], parts[1] if len(parts) > 1 else ''))
        else:
            pairs = list(capfile_or_pairs)
        self.cap_ids = [p[0] for p in pairs]
        self.files = dict(text_feat_files)
        self.batch_size = int(batch_size)
        self.device, self.pin = device, pin_memory
        self.dataset = _Dataset(len(pairs), dict(pairs))

    def __len__(self):
        return (len(self.cap_ids) + self.batch_size - 1) // self.batch_size

    def whole(self):
        """All captions as ONE (caption_feat_dict, idxs, cap_ids) batch (see BulkVisLoader.whole)."""
        return next(self._batches(max(1, len(self.cap_ids))), None)

    def __iter__(self):
        return self._batches(self.batch_size)

    def _batches(self, batch_size):
        n = len(self.cap_ids)
        for s in range(0, n, batch_size):
            ids = self.cap_ids[s:s + batch_size]
            cap = {'caption': [self.dataset.captions[i] for i in ids]}
            for key, bf in self.files.items():
                cap[key] = _to_device(bf.read_matrix(ids), self.device, self.pin)
            yield cap, list(range(s, s + len(ids))), tuple(ids)
