"""CIDEr-D reward on the device (csrc/cider.hip through the C ABI) against the reference scorer's golden vectors and
the oracle; the self-critical Trainer step with the device reward."""
import argparse

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_oracle_ciderd import CASES, load_case
from oracle import ciderd as OC

pytestmark = pytest.mark.gpu


def scorer_for(df, ref_len):
    from unpaired_image_captioning_amd.misc.rewards import DeviceCiderD
    return DeviceCiderD(df, ref_len) if df is not None else DeviceCiderD(None, None)


@pytest.mark.parametrize("name", CASES)
def test_device_scores_and_reward_match_reference_golden(name):
    from unpaired_image_captioning_amd.misc import rewards
    z, gts, df, ref_len = load_case(name)
    N, L = z["gen"].shape
    S = int(z["seq_per_img"])
    sc = scorer_for(df, ref_len)
    hyp = torch.from_numpy(np.concatenate([z["gen"], z["greedy"]], 0)).cuda()
    s = sc.scores(hyp, gts, N, S).cpu().numpy()
    # f64, the reference's own summation order, host-made penalty table: equal to the last bits (tolerance: 4 ulp)
    assert np.abs(s - z["scores"]).max() <= 4 * np.finfo(np.float64).eps * max(1.0, np.abs(z["scores"]).max()), np.abs(s - z["scores"]).max()
    r = rewards.self_critical_reward_device(sc, torch.from_numpy(z["gen"]).cuda(), torch.from_numpy(z["greedy"]).cuda(), gts, 1.0)
    assert r.shape == (N, L) and r.dtype == torch.float32
    assert np.array_equal(r.cpu().numpy(), z["reward"].astype(np.float32)) or \
        np.abs(r.cpu().numpy() - z["reward"]).max() < 1e-6
    # string-keyed document frequencies, as the cached pickle holds them (scripts/prepro_ngrams.py:106)
    if df is not None:
        sc2 = scorer_for({tuple(str(t) for t in k): v for k, v in df.items()}, ref_len)
        assert np.array_equal(sc2.scores(hyp, gts, N, S).cpu().numpy(), s)


def test_device_scores_vs_oracle_at_training_size():
    """640 sampled + 640 greedy captions of 16 tokens against 128 images x 5 references, vocabulary 9487, a cached table
    of ~1e5 n-grams with collisions in the hash table: every score equals the oracle's."""
    g = np.random.default_rng(5)
    V, L, n_img, S = 9487, 16, 128, 5
    N = n_img * S

    def caps(n):
        r = np.zeros((n, L), dtype=np.int64)
        for i in range(n):
            ln = L if g.random() > 0.8 else int(g.integers(5, L))
            r[i, :ln] = np.minimum(g.zipf(1.3, ln), V)
        return r
    gts = [caps(5) for _ in range(n_img)]
    gen, greedy = caps(N), caps(N)
    for i in range(0, N, 3):
        gen[i, :8] = gts[i // S][i % 5][:8]
    corpus = [caps(5) for _ in range(3000)] + gts
    df = {}
    for img in corpus:
        for ng in set(k for r in img for k in OC.precook(OC.caption_words(r))):
            df[ng] = df.get(ng, 0.0) + 1.0
    assert len(df) > 50000
    ref_len = float(len(corpus))
    sc = scorer_for(df, ref_len)
    hyp = torch.from_numpy(np.concatenate([gen, greedy], 0)).cuda()
    s = sc.scores(hyp, gts, N, S).cpu().numpy()
    hyps = [OC.caption_words(r) for r in gen] + [OC.caption_words(r) for r in greedy]
    refs = [[OC.caption_words(r) for r in x] for x in gts]
    sub = list(range(0, 2 * N, 7))
    _, want = OC.ciderd_scores([hyps[i] for i in sub], [refs[i % N // S] for i in sub], df, ref_len)
    assert np.abs(s[sub] - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    assert (want > 0.5).any()


def _cider_sweep(n, seed):
    g = np.random.default_rng(seed)
    return [dict(V=int(g.integers(2, 40)), L=int(g.integers(1, 65)), Lr=int(g.integers(1, 65)), n_img=int(g.integers(1, 9)),
                 S=int(g.integers(1, 6)), corpus=bool(g.integers(0, 2)), log_ref=bool(g.integers(0, 2)), idx=i) for i in range(n)]


@pytest.mark.parametrize("cfg", _cider_sweep(20, 9), ids=lambda c: "cider%d" % c["idx"])
def test_device_scores_random_sweep_vs_oracle(cfg):
    """20 seeded random configurations: captions of 1..64 tokens (hypotheses and references of different widths), tiny
    vocabularies (many repeated n-grams and hash-table hits), 1..5 or 1..13 references, cached and 'corpus' document frequencies."""
    from unpaired_image_captioning_amd.misc import rewards
    g = np.random.default_rng(1000 + cfg["idx"])
    V, L, Lr, S = cfg["V"], cfg["L"], cfg["Lr"], cfg["S"]

    def caps(n, width):
        r = np.zeros((n, width), dtype=np.int64)
        for i in range(n):
            ln = width if g.random() > 0.6 else int(g.integers(0, width + 1))
            r[i, :ln] = g.integers(1, V + 1, ln)
        return r
    # (odd configurations: up to 13 references per image -- the kernel cooks references five at a time, one wave each, so images
    # with 6..13 of them take two or three rounds; the greedy rows given once per image take the scored-once path)
    max_refs = 14 if cfg["idx"] % 2 else 6
    gts = [caps(int(g.integers(1, max_refs)), Lr) for _ in range(cfg["n_img"])]
    N = cfg["n_img"] * S
    gen, greedy = caps(N, L), caps(N, L)
    df = ref_len = None
    if not cfg["corpus"]:
        corpus = [caps(int(g.integers(1, 6)), Lr) for _ in range(40)] + gts
        df = {}
        for img in corpus:
            for ng in set(k for r in img for k in OC.precook(OC.caption_words(r))):
                df[ng] = df.get(ng, 0.0) + 1.0
        ref_len = float(np.log(len(corpus))) if cfg["log_ref"] else float(len(corpus))
    sc = scorer_for(df, ref_len)
    r = rewards.self_critical_reward_device(sc, torch.from_numpy(gen).cuda(), torch.from_numpy(greedy).cuda(), gts, 1.0).cpu().numpy()
    want = OC.self_critical_reward(gen, greedy, gts, df, ref_len)
    assert r.shape == want.shape
    assert np.abs(r - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (cfg, np.abs(r - want).max())
    hyp = torch.from_numpy(np.concatenate([gen, greedy], 0)).cuda()
    s = sc.scores(hyp, gts, N, S).cpu().numpy()
    hyps = [OC.caption_words(x) for x in gen] + [OC.caption_words(x) for x in greedy]
    refs = [[OC.caption_words(x) for x in im] for im in gts]
    _, want_s = OC.ciderd_scores(hyps, [refs[i % N // S] for i in range(2 * N)], df, ref_len)
    assert np.abs(s - want_s).max() <= 1e-11 * max(1.0, np.abs(want_s).max()), (cfg, np.abs(s - want_s).max())
    if df is not None and S > 1:
        # one greedy row per image (what the eval-mode decode of identical replicas hands over): scored once, repeated -- the same
        # reward, bit for bit, as with the S copies written out
        g1 = greedy[::S].copy()
        rep = np.repeat(g1, S, 0)
        r_rep = rewards.self_critical_reward_device(sc, torch.from_numpy(gen).cuda(), torch.from_numpy(rep).cuda(), gts, 1.0)
        r_one = rewards.self_critical_reward_device(sc, torch.from_numpy(gen).cuda(), torch.from_numpy(g1).cuda(), gts, 1.0)
        assert torch.equal(r_rep, r_one)


def test_get_self_critical_reward_mirror_and_errors():
    from unpaired_image_captioning_amd.misc import rewards
    z, gts, df, ref_len = load_case("ciderd_cached")
    rewards.CiderD_scorer = None
    with pytest.raises(RuntimeError, match="init_scorer"):
        rewards.get_self_critical_reward(None, None, None, None, None, {"gts": gts}, torch.from_numpy(z["gen"]).cuda(),
                                         argparse.Namespace(cider_reward_weight=1, bleu_reward_weight=0))
    rewards.CiderD_scorer = scorer_for(df, ref_len)

    class FakeModel(object):
        mode = []
        def eval(self): self.mode.append("eval")
        def train(self): self.mode.append("train")
        def __call__(self, *a, **k):
            assert k.get("mode") == "sample" and self.mode[-1] == "eval"
            return torch.from_numpy(z["greedy"]).cuda(), None
    m = FakeModel()
    opt = argparse.Namespace(cider_reward_weight=1, bleu_reward_weight=0)
    r = rewards.get_self_critical_reward(m, None, None, None, None, {"gts": gts}, torch.from_numpy(z["gen"]).cuda(), opt)
    assert m.mode == ["eval", "train"] and r.shape == z["reward"].shape
    assert np.abs(r - z["reward"]).max() < 1e-6
    rewards.CiderD_scorer = None
    assert rewards.array_to_str(np.array([3, 5, 0, 7])) == "3 5 0" and rewards.array_to_str(np.array([3, 5])) == "3 5"


def test_trainer_self_critical_with_device_reward_equals_host_oracle_reward(tmp_path):
    """Trainer.train_self_critical with the device scorer (cached tokens read from a pickle as init_scorer does) takes the
    same steps as with the oracle's reward computed on the host from the same samples."""
    import pickle
    from unpaired_image_captioning_amd.trainer import Trainer
    from unpaired_image_captioning_amd.misc import rewards
    from test_gpu_topdown import make_opt, absmax
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    data = {k: I[k].numpy() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}
    g = np.random.default_rng(3)
    n_img = len(data["labels"]) // cfg["S"]
    data["gts"] = [np.concatenate([data["labels"][i * cfg["S"]:(i + 1) * cfg["S"], 1:cfg["L"] + 1],
                                   g.integers(0, cfg["V"] + 1, (2, cfg["L"]))], 0) for i in range(n_img)]
    df = {}
    for img in data["gts"]:
        for ng in set(k for r in img for k in OC.precook(OC.caption_words(r))):
            df[tuple(str(t) for t in ng)] = df.get(tuple(str(t) for t in ng), 0.0) + 1.0
    pkl = tmp_path / "toy-idxs.p"
    with open(pkl, "wb") as f:
        pickle.dump({"document_frequency": df, "ref_len": float(np.log(40.0))}, f)
    df_int = {tuple(int(t) for t in k): v for k, v in df.items()}

    def host_reward(d, sampled, greedy):
        return OC.self_critical_reward(sampled, greedy, d["gts"], df_int, float(np.log(40.0)))

    res = []
    for device_reward in (True, False):
        rewards.CiderD_scorer = None
        opt = make_opt(cfg, "f32", drop=0.5, seed=3)
        opt.i2t_learning_rate = 1e-3
        opt.seq_per_img = cfg["S"]
        opt.cached_tokens = str(pkl)
        tr = Trainer(opt)
        tr.i2t_model.load_state_dict(W)
        tr.build_optimizer()
        losses = [tr.train_self_critical(data, None if device_reward else host_reward) for _ in range(3)]
        res.append((losses, tr.i2t_avg_reward, {k: v.detach().cpu().clone() for k, v in tr.i2t_model.state_dict().items()}))
    rewards.CiderD_scorer = None
    (l0, a0, w0), (l1, a1, w1) = res
    np.testing.assert_allclose(l0, l1, rtol=0, atol=1e-6)
    assert abs(a0 - a1) < 1e-6 and any(abs(x) > 1e-4 for x in l0)
    for k in w0:
        assert absmax(w0[k], w1[k]) < 1e-6, k
