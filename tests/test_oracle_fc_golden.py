"""Pin oracle/fc.py (FC captioner, BASELINE config 1) against golden vectors from the reference's FCModel_NMT."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import fc as OF
from oracle import topdown as O


def _close(a, b, tol):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("name", ["fc_tiny", "fc_tiny_earlybreak", "fc_odd"])
def test_fc_forward_loss_grads_greedy(name):
    cfg, W, I, Out, G, X = load_golden(name)
    loss, grads, logp = OF.xe_loss_and_grads(W, I["fc_feats"], I["labels"], I["masks"])
    _close(logp, Out["logprobs"], 1e-5)
    assert abs(loss.item() - float(Out["loss"])) < 1e-5
    assert set(G) == set(grads)
    for k in G:
        _close(grads[k], G[k], 1e-5)
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    seq, lp = OF.sample(W, I["fc_feats"][idx], cfg["L"])
    assert torch.equal(seq, Out["greedy_seq"])
    _close(lp, Out["greedy_logp"], 1e-5)


def test_fc_early_break_fixture_really_breaks():
    cfg, W, I, Out, G, X = load_golden("fc_tiny_earlybreak")
    assert (Out["logprobs"].abs().sum((0, 2)) == 0).any()


def test_fc_config1_shapes():
    """BASELINE config 1: 16 images x 5 captions, seq_len 16, 2048-d fc feats, hidden 512, V+1 = 9488."""
    cfg, W, I, Out, G, X = load_golden("fc_cfg1")
    V, E, H, D, L = cfg["V"], cfg["E"], cfg["H"], cfg["D"], cfg["L"]
    wseed, dseed = [int(s) for s in torch.as_tensor(X["seeds"])]
    Wt = OF.init_weights(V + 1, E, H, D, seed=wseed)
    b = O.synthetic_batch(cfg["n_img"], cfg["S"], 3, D, V, L, seed=dseed)
    loss, grads, logp = OF.xe_loss_and_grads(Wt, b["fc_feats"], b["labels"], b["masks"])
    _close(logp[:, :, ::37], Out["logprobs_sub"], 2e-5)
    assert abs(loss.item() - float(Out["loss"])) < 2e-5
    for k, g in G.items():
        _close(grads[k], g, 2e-5)
    for k, v in X.items():
        if k.startswith("gradnorm::"):
            n = grads[k.split("::", 1)[1]].double().norm().item()
            assert abs(n - float(torch.as_tensor(v))) <= 1e-4 * max(1.0, float(torch.as_tensor(v)))
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    seq, lp = OF.sample(Wt, b["fc_feats"][idx], L)
    assert torch.equal(seq, Out["greedy_seq"])
    _close(lp, Out["greedy_logp"], 2e-5)


@pytest.mark.parametrize("name", ["fc_tiny", "fc_tiny_earlybreak", "fc_odd"])
def test_fc_beam_search_matches_reference(name):
    """FCModel_NMT._sample_beam + CaptionModel.beam_search: token ids identical, log-probs within 1e-5.  Two paths:
    the direct `_sample_beam(..., opt)` call honours the options; the public `_sample(..., opt)` call passes `opt` in the
    att_masks slot (P/models/FCModel_NMT.py:168), so the reference searches with beam_size 10 and no constraint whatever
    the caller asked for -- both behaviours are pinned."""
    cfg, W, I, Out, G, X = load_golden(name)
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    n_early = 0
    for tag in ("b3", "b2c", "b3eos", "b4ppl"):
        bs, dc, mp, eos_bias = [float(x) for x in X["beam::%s_cfg" % tag]]
        Wb = dict(W)
        Wb["logit.bias"] = W["logit.bias"].clone()
        Wb["logit.bias"][0] += eos_bias
        for prefix, args in (("beamd", (int(bs), int(dc), int(mp))), ("beam", (10, 0, 0))):
            seq, lp = OF.sample_beam(Wb, I["fc_feats"][idx], cfg["L"], *args)
            ref = torch.as_tensor(X["%s::%s_seq" % (prefix, tag)])
            assert torch.equal(seq, ref), (prefix, tag, seq, ref)
            assert (lp - torch.as_tensor(X["%s::%s_logp" % (prefix, tag)])).abs().max().item() < 1e-5
            n_early += int((ref == 0).any())
    assert n_early > 0
