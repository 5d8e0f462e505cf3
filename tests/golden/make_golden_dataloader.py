#!/usr/bin/env python3
"""Golden vectors for the input pipeline, produced by the REFERENCE's own DataLoader methods.

Runs only in the build container (needs /root/reference; nothing of the reference travels).  The reference module
P/misc/dataloader/dataloader.py is imported with harness-side stubs for what the image lacks (h5py, nltk, the onmt
package) and its real methods are called on a DataLoader object whose attributes are set by hand (its __init__ needs an
HDF5 file): `__getitem__` reads per-image .npz / .npy files written to a temp directory, `get_captions` and `get_batch`
run unmodified -- `get_batch` with a stand-in for the BlobFetcher (whose `.next()` is python 2) that calls the
reference's `__getitem__` and `_get_next_minibatch_inds`.

Each fixture holds the RAW arrays (in::), the settings, and what the reference returned (out::).

    python tests/golden/make_golden_dataloader.py
"""
import os
import sys
import types
import random
import tempfile
import argparse

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

P = "/root/reference/pivot_based_eccv2018"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    for name in ("nltk", "nltk.translate", "nltk.translate.bleu_score", "h5py", "onmt", "onmt.Markdown", "onmt.Models",
                 "onmt.modules", "onmt.Constants"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["nltk.translate.bleu_score"].SmoothingFunction = object
    sys.modules["onmt.Constants"].PAD = 0
    sys.modules["onmt"].Constants = sys.modules["onmt.Constants"]
    sys.path.insert(0, P)
    import builtins
    import functools
    builtins.reduce = functools.reduce                                  # get_batch is python 2 (:267)
    import misc.dataloader.dataloader as ref
    return ref


def make_raw(rng, n_images, D, Dfc, rmin, rmax, V, L, caps):
    raw = {"att": [], "box": [], "fc": [], "hw": []}
    for i in range(n_images):
        R = int(rng.integers(rmin, rmax + 1))
        raw["att"].append(np.abs(rng.standard_normal((R, D))).astype(np.float32))
        h, w = int(rng.integers(200, 640)), int(rng.integers(200, 640))
        x1 = rng.uniform(0, w * 0.6, R); y1 = rng.uniform(0, h * 0.6, R)
        x2 = x1 + rng.uniform(5, w * 0.4, R); y2 = y1 + rng.uniform(5, h * 0.4, R)
        box = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
        if R >= 3:
            box[2] = box[0]                                             # equal areas: the sort must be stable
        raw["box"].append(box)
        raw["fc"].append(rng.standard_normal(Dfc).astype(np.float32))
        raw["hw"].append((h, w))
    ncap = [int(c) for c in rng.integers(caps[0], caps[1] + 1, n_images)]
    M = sum(ncap)
    labels = np.zeros((M, L), dtype=np.uint32)
    for m in range(M):
        n = int(rng.integers(1, L + 1))
        labels[m, :n] = rng.integers(1, V + 1, n)
    end = np.cumsum(ncap)
    raw["labels"] = labels
    raw["label_start_ix"] = (end - np.array(ncap) + 1).astype(np.uint32)      # 1-indexed (prepro_labels.py:160-164)
    raw["label_end_ix"] = end.astype(np.uint32)
    return raw


def run_case(ref, name, seed, n_images, D, Dfc, rmin, rmax, V, L, caps, use_box, norm_att, norm_box, batch_size, S, n_batches):
    rng = np.random.default_rng(seed)
    raw = make_raw(rng, n_images, D, Dfc, rmin, rmax, V, L, caps)
    tmp = tempfile.mkdtemp(prefix="uic_golden_")
    for d in ("att", "box", "fc"):
        os.makedirs(os.path.join(tmp, d))
    info = {"images": [], "ix_to_word": {str(i + 1): "w%d" % i for i in range(V)}}
    for i in range(n_images):
        iid = 1000 + 7 * i
        np.savez(os.path.join(tmp, "att", "%d.npz" % iid), feat=raw["att"][i])
        np.save(os.path.join(tmp, "box", "%d.npy" % iid), raw["box"][i])
        np.savez(os.path.join(tmp, "fc", "%d.npz" % iid), feat=raw["fc"][i])
        info["images"].append({"id": iid, "file_path": "img/%d.jpg" % iid, "split": "train",
                               "height": raw["hw"][i][0], "width": raw["hw"][i][1]})

    dl = ref.DataLoader.__new__(ref.DataLoader)                        # __init__ opens HDF5 files: set what it would set
    dl.opt = argparse.Namespace()
    dl.use_blob_fetcher = 1
    dl.batch_size, dl.seq_per_img = batch_size, S
    dl.nmt_train_flag = dl.nmt_eval_flag = 0
    dl.type = True
    dl.fc_feat_size, dl.att_feat_size = Dfc, D + 5 * use_box
    dl.use_att, dl.use_box, dl.use_box_cls_prob = True, use_box, 0
    dl.norm_att_feat, dl.norm_box_feat = norm_att, norm_box
    dl.info = info
    dl.ix_to_word = info["ix_to_word"]
    dl.vocab_size = V
    dl.input_fc_dir, dl.input_att_dir, dl.input_box_dir = (os.path.join(tmp, d) for d in ("fc", "att", "box"))
    dl.h5_label_file = {"labels": raw["labels"]}
    dl.seq_length = L
    dl.label_start_ix, dl.label_end_ix = raw["label_start_ix"], raw["label_end_ix"]
    dl.num_images = n_images
    dl.split_ix = {"train": list(range(n_images)), "val": [], "test": []}
    dl.iterators = {"train": 0, "val": 0, "test": 0}

    class Fetcher(object):                                              # BlobFetcher.get without the py2 iterator
        def __init__(self, split):
            self.inner = ref.BlobFetcher(split, dl, split == "train")

        def get(self):
            ix, wrapped = self.inner._get_next_minibatch_inds()
            return list(dl[ix]) + [wrapped]

    dl._prefetch_process = {"train": Fetcher("train")}

    out = {"cfg": np.array([n_images, D, Dfc, V, L, use_box, norm_att, norm_box, batch_size, S, n_batches, seed])}
    for i in range(n_images):
        out["in::att_%d" % i] = raw["att"][i]
        out["in::box_%d" % i] = raw["box"][i]
        out["in::fc_%d" % i] = raw["fc"][i]
    out["in::hw"] = np.array(raw["hw"], dtype=np.int64)
    out["in::ids"] = np.array([im["id"] for im in info["images"]], dtype=np.int64)
    for k in ("labels", "label_start_ix", "label_end_ix"):
        out["in::" + k] = raw[k]
    for i in range(n_images):                                           # __getitem__ alone
        out["out::item_att_%d" % i] = dl[i][2]
    random.seed(seed)
    for b in range(n_batches):
        data = dl.get_batch("train")
        for k in ("fc_feats", "att_feats", "att_masks", "labels", "masks"):
            out["out::b%d_%s" % (b, k)] = np.asarray(data[k])
        out["out::b%d_ix" % b] = np.array([d["ix"] for d in data["infos"]], dtype=np.int64)
        out["out::b%d_id" % b] = np.array([d["id"] for d in data["infos"]], dtype=np.int64)
        for j, g in enumerate(data["gts"]):
            out["out::b%d_gts_%d" % (b, j)] = np.asarray(g)
        out["out::b%d_bounds" % b] = np.array([data["bounds"]["it_pos_now"], data["bounds"]["it_max"], int(data["bounds"]["wrapped"])])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    import shutil
    shutil.rmtree(tmp)
    print("wrote", name, {k: v.shape for k, v in out.items() if k.startswith("out::b0")})


def main():
    ref = load_reference()
    #            name                 seed  n   D    Dfc  rmin rmax V   L  caps    box na nb  bs S  nb
    run_case(ref, "dataloader_tiny",     11, 7,  16,  16,  3,   7,   40, 6, (1, 6), 1, 1, 1,  3, 2, 4)    # wraps + reshuffles
    run_case(ref, "dataloader_nobox",    12, 5,  24,  24,  2,   6,   40, 6, (2, 4), 0, 1, 0,  2, 3, 2)
    run_case(ref, "dataloader_nonorm",   13, 5,  10,  8,   1,   5,   40, 6, (5, 7), 1, 0, 0,  5, 5, 1)    # D % 4 != 0, R = 1
    run_case(ref, "dataloader_boxnorm0", 14, 4,  136, 8,   4,   9,   40, 6, (5, 5), 1, 1, 0,  4, 1, 1)    # 8 < D, two pairwise leaves
    run_case(ref, "dataloader_real",     15, 3,  2048, 2048, 10, 12, 9487, 16, (5, 5), 1, 1, 1, 3, 2, 1)


if __name__ == "__main__":
    main()
