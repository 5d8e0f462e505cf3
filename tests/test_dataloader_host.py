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

    def no_device(self, raw, counts, order, slot_of, infos):
        seen["counts"], seen["order"] = counts, order
        return None, None, None
    monkeypatch.setattr(M.DataLoader, "_assemble", no_device)
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


def test_loader_rejects_feature_files_that_are_not_float32(tmp_path):
    from unpaired_image_captioning_amd.misc.dataloader import dataloader as M
    cfg, z = load_case("dataloader_nobox")
    n = cfg["n_images"]
    att = [z["in::att_%d" % i].astype(np.float64) for i in range(n)]
    label_path = write_dataset(str(tmp_path), att, None, [z["in::fc_%d" % i] for i in range(n)], z["in::hw"], z["in::ids"],
                               z["in::labels"], z["in::label_start_ix"], z["in::label_end_ix"], cfg["V"], label_format="npz")
    opt = loader_opt(str(tmp_path), label_path, 2, 2, cfg["Dfc"], cfg["D"], 0, 1, 0)
    loader = M.DataLoader(opt, device="cpu")
    with pytest.raises(TypeError, match="float32"):
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
    monkeypatch.setattr(M.DataLoader, "_assemble", lambda self, *a: (None, None, None))
    loader = M.DataLoader(opt, device="cpu")
    n_batches = int(g["out::train_numBatches"][0])
    for b in range(n_batches + 1):
        data = loader.get_batch("train")
        want = g["out::train_%d_tgt" % (b % n_batches)]
        assert np.array_equal(data["nmt"].tgt.numpy(), want)
        assert data["bounds"]["wrapper_nmt"] == (b % n_batches == n_batches - 1)
