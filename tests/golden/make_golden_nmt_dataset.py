#!/usr/bin/env python3
"""Golden batches of the NMT corpus batcher, from the reference's own P/misc/dataloader/onmt_dataset_h5.py (imported with
harness stubs for onmt.Constants; the corpus arrays stand in for the h5py data sets).  Build container only.

    python tests/golden/make_golden_nmt_dataset.py
"""
import os
import sys
import types
import warnings

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np

P = "/root/reference/pivot_based_eccv2018"
HERE = os.path.dirname(os.path.abspath(__file__))


class H5Like(object):
    """what the batcher needs of an h5py data set: slicing, len, truthiness"""

    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def __getitem__(self, k):
        return self.a[k]

    def __len__(self):
        return len(self.a)

    def __bool__(self):
        return True


def main():
    for name in ("onmt", "onmt.Constants"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["onmt.Constants"].PAD = 0
    sys.modules["onmt"].Constants = sys.modules["onmt.Constants"]
    sys.path.insert(0, P)
    pkg = types.ModuleType("misc")
    pkg.__path__ = [os.path.join(P, "misc")]
    sys.modules["misc"] = pkg
    sub = types.ModuleType("misc.dataloader")
    sub.__path__ = [os.path.join(P, "misc", "dataloader")]
    sys.modules["misc.dataloader"] = sub
    warnings.simplefilter("ignore")
    from misc.dataloader.onmt_dataset_h5 import onmt_dataset_h5
    rng = np.random.default_rng(5)
    M, Ls, Lt, bs = 23, 12, 14, 6
    out = {"cfg": np.array([M, Ls, Lt, bs])}
    corpus = {}
    for split in ("train", "valid"):
        for side, L in (("src", Ls), ("tgt", Lt)):
            n = M if split == "train" else 7
            length = rng.integers(1, L + 1, n).astype(np.uint32)
            lab = np.zeros((n, L), dtype=np.uint32)
            for i in range(n):
                lab[i, :length[i]] = rng.integers(4, 50000, length[i])
            corpus["%s_%s_label" % (split, side)] = lab
            corpus["%s_%s_label_length" % (split, side)] = length
    for k, v in corpus.items():
        out["in::" + k] = v
    wrapped = {k: (H5Like(v) if not k.endswith("length") else v) for k, v in corpus.items()}
    for split in ("train", "valid"):
        ds = onmt_dataset_h5(wrapped, split, bs, 0)
        out["out::%s_numBatches" % split] = np.array([len(ds)])
        for b in range(len(ds)):
            batch = ds[b]
            out["out::%s_%d_src" % (split, b)] = batch.src.data.numpy()
            out["out::%s_%d_tgt" % (split, b)] = batch.tgt.data.numpy()
            out["out::%s_%d_lengths" % (split, b)] = batch.lengths.data.numpy()
            out["out::%s_%d_indices" % (split, b)] = np.array(batch.indices)
            out["out::%s_%d_batchSize" % (split, b)] = np.array([batch.batchSize])
    np.savez_compressed(os.path.join(HERE, "nmt_dataset.npz"), **out)
    print("wrote nmt_dataset.npz", len(out))


if __name__ == "__main__":
    main()
