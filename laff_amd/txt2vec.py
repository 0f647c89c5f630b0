"""Text-side feature producers that are cheap on the device.

The reference turns every caption into a dense |vocab|-wide count vector on the host (`txt2vec.BowVec._encoding`,
/root/reference/txt2vec.py:56-63) and multiplies it with the FC weight as a dense GEMM (`BoWTxtEncoder`,
model/model.py:399-416), and averages word2vec rows fetched one BigFile read per caption (`W2Vec._encoding`,
txt2vec.py:97-104).  Here the host only tokenises (same rules: textlib.py:27-45) and maps words to ids; the arithmetic is a
gather-sum on the GPU (`laff_fc_gather_act_bn`): bow -> a CSR count matrix that goes straight into the TransformNet
(gather-sum of columns of W), w2v -> mean of table rows.

Stop words are DATA of the deployment (the reference ships `stopwords_en.txt`); pass them in (`stopwords=`), e.g.
`set(open('stopwords_en.txt').read().split())`.
"""
import pickle
import re

import numpy as np
import torch
import torch.nn as nn

_NON_ALNUM = re.compile(r"[^A-Za-z0-9]")


def tokenize(text, clean=True, remove_stopword=False, stopwords=()):
    """textlib.TextTool.tokenize(language='en') (textlib.py:27-45)."""
    sent = text
    if clean:
        sent = _NON_ALNUM.sub(' ', sent.replace('\r', ' ')).strip().lower()
    tokens = sent.split()
    if remove_stopword:
        tokens = [t for t in tokens if t not in stopwords]
    return tokens


class Vocabulary(object):
    """Same attributes as textlib.Vocabulary (textlib.py:69-102), so that the reference's vocab pickles load into it."""

    def __init__(self, encoding='bow'):
        self.word2idx, self.idx2word, self.encoding = {}, {}, encoding

    def add(self, word):
        if word not in self.word2idx:
            idx = len(self.word2idx)
            self.word2idx[word] = idx
            self.idx2word[idx] = word

    def find(self, word):
        return self.word2idx.get(word, -1)

    def __getitem__(self, index):
        return self.idx2word[index]

    def __len__(self):
        return len(self.word2idx)


class _VocabUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == 'Vocabulary':           # pickled as textlib.Vocabulary by the reference's build_vocab
            return Vocabulary
        return super().find_class(module, name)


def load_vocab(path):
    """A reference vocabulary pickle (`bow_nsw_5.pkl`, ...) without the reference's `textlib` module on the path."""
    with open(path, 'rb') as f:
        return _VocabUnpickler(f).load()


def _as_vocab(vocab):
    if isinstance(vocab, str):
        return load_vocab(vocab)
    if hasattr(vocab, 'find'):
        return vocab
    v = Vocabulary()
    for w in vocab:
        v.add(w)
    return v


def _csr(rows, ncols, device, dtype=torch.float32):
    """rows: list of (sorted unique column ids, values) -> torch CSR with int32 indices."""
    crow = np.zeros(len(rows) + 1, np.int32)
    crow[1:] = np.cumsum([len(c) for c, _ in rows])
    col = np.concatenate([np.asarray(c, np.int32) for c, _ in rows]) if rows else np.zeros(0, np.int32)
    val = np.concatenate([np.asarray(v, np.float32) for _, v in rows]) if rows else np.zeros(0, np.float32)
    return torch.sparse_csr_tensor(torch.from_numpy(crow).to(device), torch.from_numpy(col.astype(np.int32)).to(device),
                                   torch.from_numpy(val).to(device=device, dtype=dtype), size=(len(rows), ncols))


class BowVec(object):
    """txt2vec.BowVec (stopwords=None) / BowVecNSW (stopwords=set) with norm=0 (the only setting the path uses)."""

    def __init__(self, vocab, stopwords=None, clean=True):
        self.vocab = _as_vocab(vocab)
        self.ndims = len(self.vocab)
        self.stopwords = None if stopwords is None else frozenset(stopwords)
        self.clean = clean

    def __len__(self):
        return self.ndims

    def _ids(self, caption):
        words = tokenize(caption, self.clean, self.stopwords is not None, self.stopwords or ())
        counts = {}
        for w in words:
            i = self.vocab.find(w)
            if i >= 0:
                counts[i] = counts.get(i, 0) + 1
        ids = sorted(counts)
        return ids, [float(counts[i]) for i in ids]

    def encoding(self, caption):
        """Dense count vector (float64), as the reference returns."""
        vec = np.zeros(self.ndims)
        ids, cnt = self._ids(caption)
        vec[ids] = cnt
        return vec

    def csr(self, captions, device):
        """(B, ndims) count matrix in CSR: the input laff_fc_gather_act_bn takes."""
        return _csr([self._ids(c) for c in captions], self.ndims, device)


class W2Vec(object):
    """txt2vec.W2Vec / W2VecNSW: mean of the word2vec rows of the caption's distinct known words."""

    def __init__(self, words, table, stopwords=None, clean=True):
        """words: list of the table's row names; table: (V, ndims) float32 array/tensor (e.g. a BigFile's matrix)."""
        self.index = {w: i for i, w in enumerate(words)}
        self.table = torch.as_tensor(np.asarray(table) if not torch.is_tensor(table) else table, dtype=torch.float32).contiguous()
        self.ndims = int(self.table.shape[1])
        self.stopwords = None if stopwords is None else frozenset(stopwords)
        self.clean = clean
        self._dev = {}

    @classmethod
    def from_bigfile(cls, datadir, stopwords=None):
        from .bigfile import BigFile
        bf = BigFile(datadir)
        mat = np.fromfile(bf.binary_file, dtype=np.float32).reshape(bf.nr_of_images, bf.ndims)
        return cls(list(bf.names), mat, stopwords)

    def _ids(self, caption):
        words = tokenize(caption, self.clean, self.stopwords is not None, self.stopwords or ())
        ids = sorted({self.index[w] for w in words if w in self.index})       # BigFile.read dedups and drops unknown names
        return ids, [1.0 / max(1, len(ids))] * len(ids)

    def encoding(self, caption):
        ids, _ = self._ids(caption)
        if not ids:
            return np.zeros(self.ndims)
        return self.table[ids].numpy().astype(np.float64).mean(axis=0)

    def csr(self, captions, device):
        return _csr([self._ids(c) for c in captions], len(self.index), device)

    def device_table(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = self.table.to(device)
        return self._dev[key]

    def encode(self, captions, device):
        """(B, ndims) mean-pooled word vectors, computed on the GPU as a gather-sum over the table."""
        from . import ops
        if self.ndims % 4:
            raise NotImplementedError('word-vector width must be a multiple of 4 (got %d)' % self.ndims)
        return ops.fc_gather_act_bn(self.csr(captions, device), self.device_table(device))


class BoWTxtEncoder(nn.Module):
    """Drop-in for model.model.BoWTxtEncoder (model/model.py:399-416): `txt_net.encoder.bow_encoder = BoWTxtEncoder(t2v)`.
    Returns the bag-of-words matrix as CSR; TransformNet projects it with the gather-sum kernel."""

    def __init__(self, t2v_bow, device='cuda'):
        super().__init__()
        self.t2v_bow, self.device = t2v_bow, device

    def forward(self, caption_feat_dict, task3=False):
        return {'text_features': self.t2v_bow.csr(caption_feat_dict['caption'], self.device)}


class W2VTxtEncoder(nn.Module):
    """Drop-in for model.model.W2VTxtEncoder (model/model.py:419-434)."""

    def __init__(self, t2v_w2v, device='cuda'):
        super().__init__()
        self.t2v_w2v, self.device = t2v_w2v, device

    def forward(self, caption_feat_dict, task3=False):
        return {'text_features': self.t2v_w2v.encode(caption_feat_dict['caption'], self.device)}
