"""Pin the CPU oracle (oracle/topdown.py) against the golden vectors that
tests/golden/make_golden.py produced from the reference's own modules."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import topdown as O

TINY = ["topdown_tiny", "topdown_tiny_ragged", "topdown_tiny_nomask", "topdown_tiny_earlybreak",
        "topdown_tiny_bn1_eval", "topdown_tiny_bn2_train", "topdown_odd", "topdown_tiny_logit2", "topdown_tiny_logit3_bn1",
        "topdown_tiny_box", "topdown_tiny_box_bn1"]
TOL = 2e-6


def _close(a, b, tol=TOL):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol * max(1.0, b.abs().max().item()), err


@pytest.mark.parametrize("name", TINY)
def test_prepare_and_steps(name):
    cfg, W, I, Out, G, X = load_golden(name)
    am = I.get("att_masks")
    training = bool(cfg["bn_train"])
    Wc = {k: v.clone() for k, v in W.items()}
    fc, att, p_att, masks = O.prepare_feature(Wc, I["fc_feats"], I["att_feats"], am, None,
                                              cfg["use_bn"], training)
    _close(fc, Out["fc_embed"])
    _close(att, Out["att_embed"])
    _close(p_att, Out["p_att"])
    N, H = fc.shape[0], cfg["H"]
    state = (torch.zeros(2, N, H), torch.zeros(2, N, H))
    for t in range(3):
        logp, state, aux = O.logprobs_step(Wc, I["labels"][:, t], fc, att, p_att, masks, state)
        _close(aux["h_att"], Out["step%d_h_att" % t])
        _close(aux["c_att"], Out["step%d_c_att" % t])
        _close(aux["att_res"], Out["step%d_att_res" % t])
        _close(aux["h_lang"], Out["step%d_h_lang" % t])
        _close(aux["c_lang"], Out["step%d_c_lang" % t])
        _close(logp, Out["step%d_logp" % t], 1e-5)


@pytest.mark.parametrize("name", TINY)
def test_forward_loss_grads(name):
    cfg, W, I, Out, G, X = load_golden(name)
    am = I.get("att_masks")
    training = bool(cfg["bn_train"])
    Wc = {k: v.clone() for k, v in W.items()}
    loss, grads, logp = O.xe_loss_and_grads(Wc, I["fc_feats"], I["att_feats"], I["labels"], I["masks"],
                                            am, None, cfg["use_bn"], training)
    _close(logp, Out["logprobs"], 1e-5)
    assert abs(loss.item() - float(Out["loss"])) < 1e-5
    assert set(G) == set(grads), set(G) ^ set(grads)
    for k in G:
        _close(grads[k], G[k], 1e-5)
    for k, v in X.items():
        if k.startswith("bnstat::"):
            key = k.split("::", 1)[1]
            _close(Wc[key].double(), torch.as_tensor(v).double(), 1e-5)


def test_early_break_zero_fill():
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_earlybreak")
    logp = O.forward_logprobs(W, I["fc_feats"], I["att_feats"], I["labels"], I.get("att_masks"))
    ref = Out["logprobs"]
    zero_cols = (ref.abs().sum((0, 2)) == 0).nonzero().view(-1)
    assert zero_cols.numel() > 0                       # the fixture really exercises the break
    assert (logp[:, zero_cols] == 0).all()


@pytest.mark.parametrize("name", TINY)
def test_greedy_and_reward(name):
    cfg, W, I, Out, G, X = load_golden(name)
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    am = I.get("att_masks")
    for k, v in X.items():          # the reference decoded after its train-mode forward updated BN stats
        if k.startswith("bnstat::"):
            W[k.split("::", 1)[1]] = torch.as_tensor(v)
    seq, lp = O.sample(W, I["fc_feats"][idx], I["att_feats"][idx], None if am is None else am[idx],
                       cfg["L"], use_bn=cfg["use_bn"])
    assert torch.equal(seq, Out["greedy_seq"])
    _close(lp, Out["greedy_logp"], 1e-5)
    rl = O.reward_criterion(Out["greedy_logp"], Out["greedy_seq"], I["reward"])
    assert abs(rl.item() - float(Out["reward_loss"])) < 1e-6


def test_adam_trajectory():
    cfg, W, I, Out, G, X = load_golden("topdown_tiny")
    P = {k: v.clone() for k, v in W.items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(v) for k, v in P.items()}
    losses = []
    for step in range(1, 4):
        loss, grads, _ = O.xe_loss_and_grads(P, I["fc_feats"], I["att_feats"], I["labels"], I["masks"],
                                             I.get("att_masks"))
        losses.append(loss.item())
        O.adam_step(P, grads, m, v, step, 5e-4)
    np.testing.assert_allclose(losses, Out["adam_losses"].numpy(), rtol=0, atol=2e-5)
    _close(P["logit.bias"], Out["adam_final_logit_bias"], 1e-5)
    _close(P["core.attention.h2att.weight"], Out["adam_final_h2att_weight"], 1e-5)


def test_real_size_rows():
    """BASELINE config-2 shapes (R=36, D=2048, H=E=A=512, V1=9488), 4 caption rows."""
    cfg, W, I, Out, G, X = load_golden("topdown_real_n4")
    V, E, H, A, D, L = (cfg[k] for k in "VEHADL")
    wseed, dseed = [int(s) for s in torch.as_tensor(X["seeds"])]
    Wt = O.init_weights(V + 1, E, H, A, D, D, seed=wseed)
    b = O.synthetic_batch(cfg["n_img"], cfg["S"], cfg["R"], D, V, L, seed=dseed, ragged_regions=True)
    loss, grads, logp = O.xe_loss_and_grads(Wt, b["fc_feats"], b["att_feats"], b["labels"], b["masks"],
                                            b["att_masks"], None, 0, False)
    _close(logp[:, :, ::37], Out["logprobs_sub"], 2e-5)
    _close(logp.double().sum(2), Out["logprobs_rowsum"], 1e-5)
    assert abs(loss.item() - float(Out["loss"])) < 2e-5
    for k, g in G.items():
        _close(grads[k], g, 2e-5)
    for k, val in X.items():
        if k.startswith("gradnorm::"):
            n = grads[k.split("::", 1)[1]].double().norm().item(); val = torch.as_tensor(val)
            assert abs(n - float(val)) <= 1e-4 * max(1.0, float(val))
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    seq, lp = O.sample(Wt, b["fc_feats"][idx], b["att_feats"][idx], b["att_masks"][idx], L)
    assert torch.equal(seq, Out["greedy_seq"])
    _close(lp, Out["greedy_logp"], 2e-5)


def test_scheduled_sampling_matches_reference_draw_for_draw():
    """ss_prob = 0.5 (AttModel.py:130-143): with torch's CPU generator seeded as the golden harness seeded it, the oracle's
    uniform_/multinomial calls replay the reference's, so log-probs, loss and gradients must agree; and at least one
    input token really was replaced by a sampled one."""
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ss")
    ss_prob, ss_seed = float(X["ss"][0]), int(X["ss"][1])
    torch.manual_seed(ss_seed)
    loss, grads, logp = O.xe_loss_and_grads(W, I["fc_feats"], I["att_feats"], I["labels"], I["masks"], I["att_masks"],
                                            None, 0, True, ss={"prob": ss_prob})
    _close(logp, Out["logprobs"], 1e-5)
    assert abs(loss.item() - float(Out["loss"])) < 1e-5
    for k in G:
        _close(grads[k], G[k], 1e-5)
    torch.manual_seed(ss_seed)
    _, aux = O.forward_logprobs(W, I["fc_feats"], I["att_feats"], I["labels"], I["att_masks"], None, 0, True, True,
                                ss={"prob": ss_prob})
    used = aux["inputs"]
    assert (used != I["labels"][:, :used.shape[1]]).any()
    # teacher forcing (no ss) gives a different loss on this fixture
    loss_tf, _, _ = O.xe_loss_and_grads(W, I["fc_feats"], I["att_feats"], I["labels"], I["masks"], I["att_masks"])
    assert abs(loss_tf.item() - float(Out["loss"])) > 1e-4


BEAM_TAGS = ("b3", "b2c", "b3eos", "b4ppl")


@pytest.mark.parametrize("name", TINY)
def test_beam_search_matches_reference(name):
    """AttModel._sample_beam + CaptionModel.beam_search (beam 2-4, decoding_constraint, max_ppl, early-finishing beams via a
    raised EOS bias): token ids identical, per-step log-probs within 1e-5."""
    cfg, W, I, Out, G, X = load_golden(name)
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    am = I.get("att_masks")
    W = dict(W)
    for k, v in X.items():          # the reference decoded after its train-mode forward updated BN stats
        if k.startswith("bnstat::"):
            W[k.split("::", 1)[1]] = torch.as_tensor(v)
    n_early = 0
    for tag in BEAM_TAGS:
        bs, dc, mp, eos_bias = [float(x) for x in X["beam::%s_cfg" % tag]]
        Wb = dict(W)
        bk = O.logit_final_key(W, "bias")
        Wb[bk] = W[bk].clone()
        Wb[bk][0] += eos_bias
        seq, lp = O.sample_beam(Wb, I["fc_feats"][idx], I["att_feats"][idx], None if am is None else am[idx], cfg["L"],
                                int(bs), int(dc), int(mp), use_bn=cfg["use_bn"])
        ref_seq = torch.as_tensor(X["beam::%s_seq" % tag])
        assert torch.equal(seq, ref_seq), (tag, seq, ref_seq)
        _close(lp, torch.as_tensor(X["beam::%s_logp" % tag]), 1e-5)
        n_early += int((ref_seq == 0).any())
    assert n_early > 0          # the fixtures do contain beams that finished before the last step


@pytest.mark.parametrize("name", TINY)
def test_diverse_beam_search_matches_reference(name):
    """group_size > 1 (CaptionModel.py:36-45,100-177): the staggered groups with diversity penalties, restated in full;
    what _sample_beam returns is the best beam of group 0 -- identical to a plain search with beam_size // group_size
    beams, which is how the device path serves it."""
    cfg, W, I, Out, G, X = load_golden(name)
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    am = I.get("att_masks")
    W = dict(W)
    for k, v in X.items():
        if k.startswith("bnstat::"):
            W[k.split("::", 1)[1]] = torch.as_tensor(v)
    for tag in ("g2b4", "g3b6eos"):
        bs, gs, dc, mp, eos_bias, lam = [float(x) for x in X["beamg::%s_cfg" % tag]]
        Wb = dict(W)
        bk = O.logit_final_key(W, "bias")
        Wb[bk] = W[bk].clone()
        Wb[bk][0] += eos_bias
        args = (Wb, I["fc_feats"][idx], I["att_feats"][idx], None if am is None else am[idx], cfg["L"])
        seq, lp, beams = O.sample_beam(*args, int(bs), int(dc), int(mp), use_bn=cfg["use_bn"], group_size=int(gs),
                                       diversity_lambda=lam, return_beams=True)
        ref_seq = torch.as_tensor(X["beamg::%s_seq" % tag])
        assert torch.equal(seq, ref_seq), (tag, seq, ref_seq)
        _close(lp, torch.as_tensor(X["beamg::%s_logp" % tag]), 1e-5)
        seq1, lp1 = O.sample_beam(*args, int(bs) // int(gs), int(dc), int(mp), use_bn=cfg["use_bn"])
        assert torch.equal(seq1, seq) and torch.equal(lp1, lp)
        # the later groups did feel the penalty: their beams differ from group 0's somewhere in the fixture
        assert all(len(b) == int(bs) for b in beams)
        if eos_bias == 0:
            assert any(not torch.equal(b[0]["seq"], b[int(bs) // int(gs)]["seq"]) for b in beams)
