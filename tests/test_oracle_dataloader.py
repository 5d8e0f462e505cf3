"""The CPU restatement of the batch assembly (oracle/dataloader.py) against what the reference's own DataLoader methods
returned for the same raw arrays (tests/golden/dataloader_*.npz, made by tests/golden/make_golden_dataloader.py).
Bit-exact: float32 numpy arithmetic in the reference's order."""
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import dataloader as O

CASES = ["dataloader_tiny", "dataloader_nobox", "dataloader_nonorm", "dataloader_boxnorm0", "dataloader_real"]
CFG = ["n_images", "D", "Dfc", "V", "L", "use_box", "norm_att", "norm_box", "batch_size", "S", "n_batches", "seed"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip(CFG, [int(x) for x in z["cfg"]]))
    return cfg, z


def oracle_item(cfg, z, i):
    box = z["in::box_%d" % i] if cfg["use_box"] else None
    h, w = (int(v) for v in z["in::hw"][i])
    return O.region_features(z["in::att_%d" % i], box, h, w, cfg["norm_att"], cfg["norm_box"])


@pytest.mark.parametrize("name", CASES)
def test_region_features_match_reference_getitem(name):
    cfg, z = load_case(name)
    for i in range(cfg["n_images"]):
        got = oracle_item(cfg, z, i)
        want = z["out::item_att_%d" % i]
        assert got.dtype == want.dtype == np.float32
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name", CASES)
def test_batches_match_reference_get_batch(name):
    cfg, z = load_case(name)
    S, L = cfg["S"], cfg["L"]
    split_ix = {"train": list(range(cfg["n_images"]))}
    iterators = {"train": 0}
    labels, start, end = z["in::labels"], z["in::label_start_ix"], z["in::label_end_ix"]
    random.seed(cfg["seed"])
    with np.errstate(over="ignore"):
        for b in range(cfg["n_batches"]):
            fc, att, rows, gts, infos, wrapped = [], [], [], [], [], False
            for _ in range(cfg["batch_size"]):
                ix, w = O.next_index(split_ix, iterators, "train", True)
                wrapped |= w
                fc.append(z["in::fc_%d" % ix])
                att.append(oracle_item(cfg, z, ix))
                rows.append(O.get_captions(labels, start, end, ix, S, L))
                gts.append(labels[start[ix] - 1: end[ix]])
                infos.append({"ix": ix, "id": int(z["in::ids"][ix])})
            data = O.merge_batch(fc, att, np.vstack(rows), gts, infos, S, L)
            for k in ("fc_feats", "att_feats", "att_masks", "labels", "masks"):
                want = z["out::b%d_%s" % (b, k)]
                assert data[k].shape == want.shape and np.array_equal(data[k], want), (b, k)
            assert [d["ix"] for d in data["infos"]] == list(z["out::b%d_ix" % b])
            for j, g in enumerate(data["gts"]):
                assert np.array_equal(g, z["out::b%d_gts_%d" % (b, j)])
            assert [iterators["train"], cfg["n_images"], int(wrapped)] == list(z["out::b%d_bounds" % b])


def test_pairwise_sum_order_is_what_the_device_kernel_assumes():
    """csrc/loader.hip restates numpy's pairwise summation (8 interleaved accumulators per <= 128-element block, halves
    split at a multiple of 8).  If a numpy release changed that order the golden vectors would still hold but the
    oracle would drift from them: keep the assumption itself under test."""
    f32 = np.float32

    def pw(a):
        n = len(a)
        if n < 8:
            r = f32(0.)
            for x in a:
                r = f32(r + x)
            return r
        if n <= 128:
            r = [a[i] for i in range(8)]
            i = 8
            while i < n - (n % 8):
                for j in range(8):
                    r[j] = f32(r[j] + a[i + j])
                i += 8
            res = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
            while i < n:
                res = f32(res + a[i])
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return f32(pw(a[:n2]) + pw(a[n2:]))

    rng = np.random.default_rng(0)
    for n in (1, 5, 7, 8, 9, 100, 128, 129, 136, 1000, 2048, 2053):
        x = np.abs(rng.standard_normal((4, n))).astype(f32)
        s = x * x
        want = np.add.reduce(s, axis=1)
        got = np.array([pw(list(r)) for r in s], dtype=f32)
        assert np.array_equal(got, want), n
