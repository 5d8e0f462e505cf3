"""Host side of the input pipeline without a GPU: the label store (HDF5 through the C library / .npz), the iteration
state and caption sampling of DataLoader.get_batch against the reference's own batches (the device assembly is stubbed
out: it is covered by tests/test_gpu_dataloader.py), and the exported symbol."""
import random

import numpy as np
import pytest

from dataset_files import loader_opt, write_dataset
from test_oracle_dataloader import CASES, load_case


def test_label_store_hdf5_round_trip(tmp_path):
    from unpaired_image_captioning_amd.misc.dataloader import label_store
    try:
        label_store.Hdf5Library()
    except ImportError:
        pytest.skip("no libhdf5 on this machine")
    rng = np.random.default_rng(0)
    arrays = {"labels": rng.integers(0, 9000, (11, 16)).astype(np.uint32), "label_start_ix": np.array([1, 4, 9], dtype=np.uint32),
              "label_end_ix": np.array([3, 8, 11], dtype=np.uint32), "label_length": rng.integers(1, 16, 11).astype(np.uint32)}
    path = str(tmp_path / "labels.h5")
    label_store.write_hdf5(path, arrays)
    with open(path, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"
    back = label_store.open_label_store(path)
    assert sorted(back) == sorted(arrays)
    for k, v in arrays.items():
        assert back[k].dtype == v.dtype and np.array_equal(back[k], v)
    nmt = {k: rng.integers(0, 50000, (5, 7)).astype(np.int64 if "length" not in k else np.uint32) for k in label_store.NMT_NAMES}
    label_store.write_hdf5(path, nmt)
    back = label_store.open_label_store(path, label_store.NMT_NAMES)
    assert all(np.array_equal(back[k], nmt[k]) and back[k].dtype == nmt[k].dtype for k in nmt)


def test_label_store_npz_and_missing_file(tmp_path):
    from unpaired_image_captioning_amd.misc.dataloader import label_store
    path = str(tmp_path / "labels.npz")
    np.savez(path, labels=np.ones((2, 3), dtype=np.uint32), label_start_ix=np.array([1]), label_end_ix=np.array([2]))
    back = label_store.open_label_store(path)
    assert back["labels"].shape == (2, 3)
    with pytest.raises(FileNotFoundError):
        label_store.open_label_store(str(tmp_path / "absent.h5"))


@pytest.mark.parametrize("name", CASES)
def test_iteration_and_caption_sampling_follow_the_reference(name, tmp_path, monkeypatch):
    from unpaired_image_captioning_amd.misc.dataloader import dataloader as M
    cfg, z = load_case(name)
    n = cfg["n_images"]
    label_path = write_dataset(str(tmp_path), [z["in::att_%d" % i] for i in range(n)], [z["in::box_%d" % i] for i in range(n)],
                               [z["in::fc_%d" % i] for i in range(n)], z["in::hw"], z["in::ids"], z["in::labels"],
                               z["in::label_start_ix"], z["in::label_end_ix"], cfg["V"], label_format="npz")
    opt = loader_opt(str(tmp_path), label_path, cfg["batch_size"], cfg["S"], cfg["Dfc"], cfg["D"] + 5 * cfg["use_box"],
                     cfg["use_box"], cfg["norm_att"], cfg["norm_box"])
    seen = {}

    def no_device(self, st):
        seen.update(st)
        return None, None, None
    monkeypatch.setattr(M.DataLoader, "_ship", no_device)
    loader = M.DataLoader(opt, device="cpu")
    random.seed(cfg["seed"])
    for b in range(cfg["n_batches"]):
        data = loader.get_batch("train")
        assert np.array_equal(data["labels"], z["out::b%d_labels" % b])
        assert np.array_equal(data["masks"], z["out::b%d_masks" % b]) and data["masks"].dtype == np.float32
        assert [d["ix"] for d in data["infos"]] == list(z["out::b%d_ix" % b])
        for j, g in enumerate(data["gts"]):
            assert np.array_equal(g, z["out::b%d_gts_%d" % (b, j)])
        bd = data["bounds"]
        assert [bd["it_pos_now"], bd["it_max"], int(bd["wrapped"])] == list(z["out::b%d_bounds" % b])
        want_regions = z["out::b%d_att_masks" % b][::cfg["S"]].sum(1).astype(int)
        assert [seen["counts"][i] for i in seen["order"]] == list(want_regions)
        # what the library's reader staged for the kernel: raw files back to back in fetch order, fc rows in batch order
        fetch = [data["infos"][seen["slot_of"][i]]["ix"] for i in range(cfg["batch_size"])]
        assert np.array_equal(seen["feat"][0].numpy(), np.concatenate([z["in::att_%d" % ix] for ix in fetch], 0))
        assert np.array_equal(seen["fc"][0].numpy(), np.stack([z["in::fc_%d" % d["ix"]] for d in data["infos"]]))
        assert np.array_equal(seen["meta"][0].numpy()[:cfg["batch_size"] + 1], seen["start"])
        if cfg["use_box"]:
            total = int(seen["start"][-1])
            assert np.array_equal(seen["box"][0].numpy()[:total * 4].reshape(-1, 4),
                                  np.concatenate([z["in::box_%d" % ix] for ix in fetch], 0))
            hw = seen["box"][0].numpy()[total * 4:].reshape(-1, 3)
            assert np.array_equal(hw[:, :2], z["in::hw"][fetch].astype(np.float32))


def test_loader_rejects_feature_files_that_are_not_float32(tmp_path):
    from unpaired_image_captioning_amd.misc.dataloader import dataloader as M
    cfg, z = load_case("dataloader_nobox")
    n = cfg["n_images"]
    att = [z["in::att_%d" % i].astype(np.float64) for i in range(n)]
    label_path = write_dataset(str(tmp_path), att, None, [z["in::fc_%d" % i] for i in range(n)], z["in::hw"], z["in::ids"],
                               z["in::labels"], z["in::label_start_ix"], z["in::label_end_ix"], cfg["V"], label_format="npz")
    opt = loader_opt(str(tmp_path), label_path, 2, 2, cfg["Dfc"], cfg["D"], 0, 1, 0)
    loader = M.DataLoader(opt, device="cpu")
    with pytest.raises(RuntimeError, match="float32"):
        loader.get_batch("train")


def test_padded_width_is_the_engines():
    from unpaired_image_captioning_amd.misc.dataloader.dataloader import padded_width
    assert [padded_width(d) for d in (2048, 2053, 8, 9, 136)] == [2048, 2176, 8, 128, 136]


def test_nmt_corpus_batches_equal_the_reference_batchers():
    """misc/dataloader/onmt_dataset_h5.py against the reference's batcher on the same corpus arrays
    (tests/golden/nmt_dataset.npz, made by tests/golden/make_golden_nmt_dataset.py)."""
    import os
    from conftest import GOLDEN
    from unpaired_image_captioning_amd.misc.dataloader.onmt_dataset_h5 import onmt_dataset_h5
    z = np.load(os.path.join(GOLDEN, "nmt_dataset.npz"))
    M, Ls, Lt, bs = (int(v) for v in z["cfg"])
    corpus = {k[4:]: z[k] for k in z.files if k.startswith("in::")}
    for split in ("train", "valid"):
        ds = onmt_dataset_h5(corpus, split, bs, 0)
        assert len(ds) == int(z["out::%s_numBatches" % split][0])
        for b in range(len(ds)):
            batch = ds[b]
            for field in ("src", "tgt", "lengths"):
                want = z["out::%s_%d_%s" % (split, b, field)]
                got = getattr(batch, field).numpy()
                assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), (split, b, field)
            assert batch.indices == list(z["out::%s_%d_indices" % (split, b)])
            assert batch.batchSize == int(z["out::%s_%d_batchSize" % (split, b)][0])
            assert np.array_equal(batch.words().numpy(), batch.src.numpy()[:, :, 0])
    with pytest.raises(AssertionError):
        ds[len(ds)]


def test_loader_serves_nmt_batches_beside_the_caption_batch(tmp_path, monkeypatch):
    import os
    from conftest import GOLDEN
    from unpaired_image_captioning_amd.misc.dataloader import dataloader as M
    cfg, z = load_case("dataloader_nobox")
    n = cfg["n_images"]
    label_path = write_dataset(str(tmp_path), [z["in::att_%d" % i] for i in range(n)], None, [z["in::fc_%d" % i] for i in range(n)],
                               z["in::hw"], z["in::ids"], z["in::labels"], z["in::label_start_ix"], z["in::label_end_ix"],
                               cfg["V"], label_format="npz")
    g = np.load(os.path.join(GOLDEN, "nmt_dataset.npz"))
    corpus_path = str(tmp_path / "nmt.npz")
    np.savez(corpus_path, **{k[4:]: g[k] for k in g.files if k.startswith("in::")})
    opt = loader_opt(str(tmp_path), label_path, 6, 2, cfg["Dfc"], cfg["D"], 0, 1, 0)
    opt.nmt_train_flag, opt.input_nmt_h5 = 1, corpus_path
    monkeypatch.setattr(M.DataLoader, "_ship", lambda self, st: (None, None, None))
    loader = M.DataLoader(opt, device="cpu")
    n_batches = int(g["out::train_numBatches"][0])
    for b in range(n_batches + 1):
        data = loader.get_batch("train")
        want = g["out::train_%d_tgt" % (b % n_batches)]
        assert np.array_equal(data["nmt"].tgt.numpy(), want)
        assert data["bounds"]["wrapper_nmt"] == (b % n_batches == n_batches - 1)


def test_library_reader_takes_npy_stored_npz_and_deflated_npz(tmp_path):
    """uic_loader_scan / uic_loader_read on the three ways the reference's scripts write feature files
    (scripts/make_bu_data.py:55-57: np.savez_compressed for att, np.save for fc and boxes; np.savez elsewhere)."""
    import ctypes as C
    from unpaired_image_captioning_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    arrays = [rng.standard_normal((37, 2048)).astype(np.float32), rng.standard_normal((5, 16)).astype(np.float32),
              np.abs(rng.standard_normal((1, 8))).astype(np.float32).round(1)]
    for kind in ("stored", "deflated"):
        paths = []
        for i, a in enumerate(arrays):
            p = str(tmp_path / ("%s_%d.npz" % (kind, i)))
            (np.savez if kind == "stored" else np.savez_compressed)(p, feat=a)
            paths.append(p.encode())
        arr = (C.c_char_p * len(paths))(*paths)
        info = np.zeros((len(paths), 6), dtype=np.int64)
        _lib.check(lib.uic_loader_scan(arr, len(paths), b"feat", info.ctypes.data, 4), "scan")
        assert [tuple(r[:3]) for r in info] == [(2,) + a.shape for a in arrays]
        assert set(info[:, 4]) == ({0} if kind == "stored" else {8})
        out = [np.full(a.shape, np.nan, dtype=np.float32) for a in arrays]
        dst = (C.c_void_p * len(out))(*[o.ctypes.data for o in out])
        _lib.check(lib.uic_loader_read(arr, len(paths), info.ctypes.data, dst, 4), "read")
        assert all(np.array_equal(o, a) for o, a in zip(out, arrays))
    vec = rng.standard_normal(2048).astype(np.float32)
    p = str(tmp_path / "v.npy")
    np.save(p, vec)
    arr = (C.c_char_p * 1)(p.encode())
    info = np.zeros((1, 6), dtype=np.int64)
    _lib.check(lib.uic_loader_scan(arr, 1, None, info.ctypes.data, 1), "scan")
    assert tuple(info[0, :5]) == (1, 2048, 1, 128, -1)
    out = np.empty(2048, dtype=np.float32)
    _lib.check(lib.uic_loader_read(arr, 1, info.ctypes.data, (C.c_void_p * 1)(out.ctypes.data), 1), "read")
    assert np.array_equal(out, vec)
    np.savez(str(tmp_path / "two.npz"), other=vec, feat=vec)
    arr = (C.c_char_p * 1)(str(tmp_path / "two.npz").encode())
    with pytest.raises(RuntimeError, match="first zip member is not feat.npy"):
        _lib.check(lib.uic_loader_scan(arr, 1, b"feat", info.ctypes.data, 1), "scan")
    arr = (C.c_char_p * 1)(str(tmp_path / "absent.npz").encode())
    with pytest.raises(RuntimeError, match="cannot read"):
        _lib.check(lib.uic_loader_scan(arr, 1, b"feat", info.ctypes.data, 1), "scan")


def _raw_deflate(data, level=6, strategy=None):
    import zlib
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY if strategy is None else strategy)
    return c.compress(data) + c.flush()


def _inflate_cases():
    rng = np.random.default_rng(1)
    return {
        "bottom-up-like (|N|, f32)": np.abs(rng.standard_normal((36, 2048))).astype(np.float32).tobytes(),
        "half zeros (relu, f32)": np.maximum(rng.standard_normal((20, 2048)), 0).astype(np.float32).tobytes(),
        "zeros": bytes(100000), "one byte": b"x", "empty": b"", "text": b"the quick brown fox " * 3000,
        "random bytes": rng.integers(0, 256, 150000, dtype=np.uint8).tobytes(),
        "small alphabet": rng.integers(0, 4, 100000, dtype=np.uint8).tobytes(),
    }


def test_the_loaders_own_inflate_agrees_with_zlib_on_every_block_type():
    """csrc/inflate_fast.h (what uic_loader_read decodes np.savez_compressed members with) against zlib, on stored, fixed-Huffman
    and dynamic-Huffman blocks, literal-only and match-heavy data, empty and one-byte streams; a stream of the wrong length or a
    truncated one must be DECLINED without a byte written outside the destination; a corrupted stream is either declined or
    decoded to exactly what zlib makes of it."""
    import ctypes as C
    import zlib
    from unpaired_image_captioning_amd import _lib
    lib = _lib.load()

    def run(src, m, fast):
        out = np.full(m + 8, 0xAB, dtype=np.uint8)
        rc = lib.uic_loader_inflate(src, len(src), out.ctypes.data, m, fast)
        return rc, out[:m].tobytes(), out[m:].tobytes()
    guard = bytes([0xAB]) * 8
    cases = _inflate_cases()
    settings = ((6, None), (1, None), (9, None), (0, None), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE))
    for name, data in cases.items():
        for level, strat in settings:
            src = _raw_deflate(data, level, strat)
            rc0, o0, _ = run(src, len(data), 0)
            rc1, o1, g1 = run(src, len(data), 1)
            assert rc0 == 0 and o0 == data, (name, level, strat)
            assert rc1 == 0 and o1 == data and g1 == guard, (name, level, strat, rc1)
            if data:
                rc, _, g = run(src, len(data) - 1, 1)
                assert rc == 1 and g == guard, (name, "one byte short")
            rc, _, g = run(src, len(data) + 1, 1)
            assert rc == 1 and g == guard, (name, "one byte long")
            if len(src) > 4:
                rc, _, g = run(src[:len(src) // 2], len(data), 1)
                assert rc == 1 and g == guard, (name, "truncated")
    data = cases["bottom-up-like (|N|, f32)"]
    src = bytearray(_raw_deflate(data))
    rng = np.random.default_rng(2)
    declined = 0
    for _ in range(200):
        bad = bytearray(src)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        rc, o, g = run(bytes(bad), len(data), 1)
        assert g == guard
        if rc == 0:
            assert o == zlib.decompress(bytes(bad), -15)
        declined += rc
    assert declined > 0


def test_two_streams_in_lock_step_decode_like_one_at_a_time():
    """uic_loader_inflate_pair: one thread, two deflate streams decoded side by side (how uic_loader_read takes deflated members:
    the two dependency chains share a core).  Every pairing of data kinds x block types; a broken stream must not disturb its
    partner."""
    import itertools
    import zlib
    from unpaired_image_captioning_amd import _lib
    lib = _lib.load()
    cases = _inflate_cases()

    def run_pair(s0, m0, s1, m1):
        o0, o1 = np.full(m0 + 8, 0xAB, dtype=np.uint8), np.full(m1 + 8, 0xAB, dtype=np.uint8)
        rc = lib.uic_loader_inflate_pair(s0, len(s0), o0.ctypes.data, m0, s1, len(s1), o1.ctypes.data, m1)
        return rc, o0[:m0].tobytes(), o1[:m1].tobytes(), o0[m0:].tobytes() + o1[m1:].tobytes()
    guard = bytes([0xAB]) * 16
    settings = ((6, None), (0, None), (6, zlib.Z_FIXED), (1, zlib.Z_RLE))
    streams = {(k, s): _raw_deflate(v, *s) for k, v in cases.items() for s in settings}
    for (a, sa), (b, sb) in itertools.product(streams, streams):
        rc, o0, o1, g = run_pair(streams[(a, sa)], len(cases[a]), streams[(b, sb)], len(cases[b]))
        assert rc == 0 and o0 == cases[a] and o1 == cases[b] and g == guard, (a, sa, b, sb, rc)
    a, b = "bottom-up-like (|N|, f32)", "half zeros (relu, f32)"
    good = streams[(a, (6, None))]
    broken = bytearray(streams[(b, (6, None))])
    broken[len(broken) // 3] ^= 0x55
    rc, o0, _, g = run_pair(good, len(cases[a]), bytes(broken), len(cases[b]))
    assert (rc & 1) == 0 and o0 == cases[a] and g == guard
    rc, _, o1, g = run_pair(bytes(broken), len(cases[b]), good, len(cases[a]))
    assert (rc & 2) == 0 and o1 == cases[a] and g == guard


def test_reader_pairs_deflated_members_and_falls_back_to_zlib(tmp_path, monkeypatch):
    """uic_loader_read over an odd number of deflated members mixed with stored ones and plain .npy files (pairs + one single +
    the rest), on one thread and on several: every array arrives, whichever decoder took it."""
    import ctypes as C
    from unpaired_image_captioning_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    arrays, paths = [], []
    for i in range(11):
        a = np.abs(rng.standard_normal((int(rng.integers(1, 40)), 256))).astype(np.float32)
        p = str(tmp_path / ("f%d" % i))
        if i % 3 == 2:
            np.savez(p, feat=a); p += ".npz"
        elif i == 4:
            np.save(p, a); p += ".npy"
        else:
            np.savez_compressed(p, feat=a); p += ".npz"
        arrays.append(a)
        paths.append(p.encode())
    arr = (C.c_char_p * len(paths))(*paths)
    info = np.zeros((len(paths), 6), dtype=np.int64)
    _lib.check(lib.uic_loader_scan(arr, len(paths), b"feat", info.ctypes.data, 2), "scan")
    assert (info[:, 4] == 8).sum() == 7
    for nt in (1, 3):
        out = [np.full(a.shape, np.nan, dtype=np.float32) for a in arrays]
        dst = (C.c_void_p * len(out))(*[o.ctypes.data for o in out])
        _lib.check(lib.uic_loader_read(arr, len(paths), info.ctypes.data, dst, nt), "read")
        assert all(np.array_equal(o, a) for o, a in zip(out, arrays))


def test_fc_vectors_as_npy_files(tmp_path, monkeypatch):
    """make_bu_data.py writes the fc vectors as <id>.npy; the loader takes either extension."""
    import os
    from unpaired_image_captioning_amd.misc.dataloader import dataloader as M
    cfg, z = load_case("dataloader_nobox")
    n = cfg["n_images"]
    label_path = write_dataset(str(tmp_path), [z["in::att_%d" % i] for i in range(n)], None, [z["in::fc_%d" % i] for i in range(n)],
                               z["in::hw"], z["in::ids"], z["in::labels"], z["in::label_start_ix"], z["in::label_end_ix"],
                               cfg["V"], label_format="npz")
    for i, iid in enumerate(z["in::ids"]):
        os.remove(str(tmp_path / "fc" / ("%d.npz" % iid)))
        np.save(str(tmp_path / "fc" / ("%d.npy" % iid)), z["in::fc_%d" % i])
        np.savez_compressed(str(tmp_path / "att" / ("%d.npz" % iid)), feat=z["in::att_%d" % i])      # as make_bu_data.py:55
    seen = {}
    monkeypatch.setattr(M.DataLoader, "_ship", lambda self, st: (seen.update(st), (None, None, None))[1])
    loader = M.DataLoader(loader_opt(str(tmp_path), label_path, 3, 2, cfg["Dfc"], cfg["D"], 0, 1, 0), device="cpu", read_ahead=False)
    data = loader.get_batch("val" if False else "train")
    assert np.array_equal(seen["fc"][0].numpy(), np.stack([z["in::fc_%d" % d["ix"]] for d in data["infos"]]))
    fetch = [data["infos"][seen["slot_of"][i]]["ix"] for i in range(3)]
    assert np.array_equal(seen["feat"][0].numpy(), np.concatenate([z["in::att_%d" % ix] for ix in fetch], 0))


def test_ranks_partition_the_global_batch(tmp_path, monkeypatch):
    """world_size 2: both ranks walk the same global sequence (same seed) and keep disjoint halves; together they hold
    exactly the batches one loader with twice the batch size produces (rows up to the per-rank sort by region count)."""
    from unpaired_image_captioning_amd.misc.dataloader import dataloader as M
    cfg, z = load_case("dataloader_tiny")
    n = cfg["n_images"]
    label_path = write_dataset(str(tmp_path), [z["in::att_%d" % i] for i in range(n)], [z["in::box_%d" % i] for i in range(n)],
                               [z["in::fc_%d" % i] for i in range(n)], z["in::hw"], z["in::ids"], z["in::labels"],
                               z["in::label_start_ix"], z["in::label_end_ix"], cfg["V"], label_format="npz")
    monkeypatch.setattr(M.DataLoader, "_ship", lambda self, st: (None, None, None))
    S = cfg["S"]

    def run(batch_size, rank, world):
        opt = loader_opt(str(tmp_path), label_path, batch_size, S, cfg["Dfc"], cfg["D"] + 5, 1, 1, 1)
        loader = M.DataLoader(opt, device="cpu", rank=rank, world_size=world)
        random.seed(5)
        out = []
        for _ in range(5):                                  # 5 x 4 images over 7: wraps and reshuffles twice
            d = loader.get_batch("train")
            out.append({info["ix"]: d["labels"][j * S:(j + 1) * S] for j, info in enumerate(d["infos"])})
        return out, loader
    whole, _ = run(4, 0, 1)
    r0, l0 = run(2, 0, 2)
    r1, l1 = run(2, 1, 2)
    assert l0.iterators == l1.iterators and l0.split_ix == l1.split_ix
    for b in range(5):
        assert len(r0[b]) + len(r1[b]) == len(whole[b]) or len(set(r0[b]) | set(r1[b])) == len(whole[b])
        merged = dict(r0[b])
        merged.update(r1[b])
        assert sorted(merged) == sorted(whole[b])
        for ix in whole[b]:
            assert np.array_equal(merged[ix], whole[b][ix])


def test_read_ahead_survives_interleaved_splits_and_batch_sizes(tmp_path, monkeypatch):
    """The read-ahead thread stages the files it GUESSES the next call wants; a call for another split or batch size must
    get its own files, not the guess."""
    from unpaired_image_captioning_amd.misc.dataloader import dataloader as M
    cfg, z = load_case("dataloader_tiny")
    n = cfg["n_images"]
    splits = ["train", "val", "train", "val", "train", "val", "train"][:n]
    label_path = write_dataset(str(tmp_path), [z["in::att_%d" % i] for i in range(n)], [z["in::box_%d" % i] for i in range(n)],
                               [z["in::fc_%d" % i] for i in range(n)], z["in::hw"], z["in::ids"], z["in::labels"],
                               z["in::label_start_ix"], z["in::label_end_ix"], cfg["V"], splits=splits, label_format="npz")
    seen = {}
    monkeypatch.setattr(M.DataLoader, "_ship", lambda self, st: (seen.update(st), (None, None, None))[1])
    loader = M.DataLoader(loader_opt(str(tmp_path), label_path, 1, 2, cfg["Dfc"], cfg["D"] + 5, 1, 1, 1), device="cpu")
    random.seed(0)
    for split, bs in (("train", 1), ("val", 1), ("train", 2), ("train", 1), ("val", 2), ("val", 1), ("train", 1)):
        data = loader.get_batch(split, batch_size=bs)
        fetch = [data["infos"][seen["slot_of"][i]]["ix"] for i in range(bs)]
        assert all(splits[ix] == split for ix in fetch)
        assert np.array_equal(seen["feat"][0].numpy(), np.concatenate([z["in::att_%d" % ix] for ix in fetch], 0)), (split, bs)
        assert np.array_equal(seen["fc"][0].numpy(), np.stack([z["in::fc_%d" % d["ix"]] for d in data["infos"]]))


def test_cpu_budget_follows_affinity_and_cgroup_quota(monkeypatch, tmp_path):
    """The reader teams are sized by what the process may USE: the affinity mask, cut down to a cgroup CPU quota (v2 `cpu.max`) --
    a team larger than the quota gets the whole process frozen once the period's quota is spent (profiles/LOG.md, round 6)."""
    import builtins
    import os
    from unpaired_image_captioning_amd.misc.dataloader import dataloader as dl
    real_open = builtins.open

    def fake(quota_text):
        def _open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                if quota_text is None:
                    raise OSError("no cgroup v2")
                p = tmp_path / "cpu.max"
                p.write_text(quota_text)
                return real_open(str(p), *a, **k)
            if str(path).startswith("/sys/fs/cgroup/cpu/"):
                raise OSError("no cgroup v1")
            return real_open(path, *a, **k)
        return _open
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)))
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert dl.cpu_budget() == 16
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert dl.cpu_budget() == 64
    monkeypatch.setattr(builtins, "open", fake("50000 100000\n"))
    assert dl.cpu_budget() == 1
    monkeypatch.setattr(builtins, "open", fake(None))
    assert dl.cpu_budget() == 64
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(4)))
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert dl.cpu_budget() == 4


def test_store_att_uncompressed_tool_keeps_the_arrays_and_drops_the_deflate(tmp_path):
    """tools/store_att_uncompressed.py: np.savez_compressed members -> stored ones, same arrays, readable by np.load (the
    reference's loader) and by the library's reader; a second run finds nothing to do."""
    import os
    import subprocess
    import sys
    import zipfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(9)
    arrays = {}
    for i in range(5):
        arrays[i] = np.abs(rng.standard_normal((int(rng.integers(5, 30)), 64))).astype(np.float32)
        np.savez_compressed(str(tmp_path / ("%d" % i)), feat=arrays[i])
    tool = os.path.join(root, "tools", "store_att_uncompressed.py")
    out = subprocess.run([sys.executable, tool, str(tmp_path), "--workers", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "5 of 5 files rewritten" in out.stdout, out.stdout + out.stderr
    for i, a in arrays.items():
        p = str(tmp_path / ("%d.npz" % i))
        assert all(m.compress_type == zipfile.ZIP_STORED for m in zipfile.ZipFile(p).infolist())
        assert np.array_equal(np.load(p)["feat"], a)
    out = subprocess.run([sys.executable, tool, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert "0 of 5 files rewritten" in out.stdout
