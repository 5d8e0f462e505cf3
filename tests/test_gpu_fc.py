"""GPU parity of the FC captioner (BASELINE config 1: FCModel_NMT + maxout LSTMCore) against golden vectors from
the reference's own module and against the CPU oracle."""
import argparse

import pytest
import torch

from conftest import load_golden
from oracle import fc as OF
from oracle import topdown as O
from test_gpu_topdown import GRAD_TOL, LOGP_TOL, absmax, grads_close

pytestmark = pytest.mark.gpu


def make_opt(cfg, dtype, drop=0.0):
    return argparse.Namespace(vocab_size=cfg["V"], input_encoding_size=cfg["E"], rnn_type="LSTM", rnn_size=cfg["H"],
                              num_layers=1, drop_prob_lm=drop, seq_length=cfg["L"], fc_feat_size=cfg["D"],
                              caption_model="fc", compute_dtype=dtype)


def build(cfg, W, dtype, drop=0.0):
    from unpaired_image_captioning_amd import models
    from unpaired_image_captioning_amd.models.FCModel_NMT import FCModel_NMT
    model = models.setup(make_opt(cfg, dtype, drop))
    assert isinstance(model, FCModel_NMT) and list(model.state_dict().keys()) == list(W.keys())
    model.load_state_dict(W)
    return model.cuda()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["fc_tiny", "fc_tiny_earlybreak", "fc_odd"])
def test_fc_forward_loss_backward_vs_reference_golden(name, dtype):
    from unpaired_image_captioning_amd.misc.criterion import LanguageModelCriterion
    cfg, W, I, Out, G, X = load_golden(name)
    model = build(cfg, W, dtype)
    model.train()
    fc, labels, masks = I["fc_feats"].cuda(), I["labels"].cuda(), I["masks"].cuda()
    # the trainer's 5-argument call (P/trainer.py:164) and the reference's own signature give the same result
    logp = model(fc, None, None, labels, None)
    logp2 = model._forward(fc, None, labels)
    assert torch.equal(logp, logp2)
    assert absmax(logp, Out["logprobs"]) < LOGP_TOL[dtype]
    loss = LanguageModelCriterion()(logp, labels[:, 1:], masks[:, 1:])
    assert abs(loss.item() - float(Out["loss"])) < LOGP_TOL[dtype]
    loss.backward()
    grads_close({k: p.grad for k, p in model.named_parameters()}, G, GRAD_TOL[dtype])


@pytest.mark.parametrize("name", ["fc_tiny", "fc_tiny_earlybreak", "fc_odd"])
def test_fc_greedy_bit_exact_f32(name):
    cfg, W, I, Out, G, X = load_golden(name)
    model = build(cfg, W, "f32").eval()
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    seq, lp = model(I["fc_feats"][idx].cuda(), None, None, opt={"sample_max": 1}, mode="sample")
    assert tuple(seq.shape) == tuple(Out["greedy_seq"].shape)
    assert torch.equal(seq.cpu(), Out["greedy_seq"])
    assert absmax(lp, Out["greedy_logp"]) < 1e-3


def test_fc_multinomial_scored_by_oracle():
    cfg, W, I, Out, G, X = load_golden("fc_tiny")
    model = build(cfg, W, "f32").eval()
    fc = I["fc_feats"].cuda()
    seq, lp = model(fc, None, None, opt={"sample_max": 0}, mode="sample")
    # the reference feeds the RAW sampled token forward, so only rows that never finished early can be replayed from
    # the stored (masked) tokens; force a token stream without zeros instead
    forced = torch.randint(1, cfg["V"] + 1, (fc.shape[0], cfg["L"]))
    forced[0, 2] = 0                                   # one row finishes early
    seq2, lp2 = model(fc, None, None, opt={"sample_max": 0, "forced_tokens": forced.cuda()}, mode="sample")
    seq_o, lp_o = OF.sample(W, I["fc_feats"], cfg["L"], sample_max=0, forced_tokens=forced)
    assert torch.equal(seq2.cpu(), seq_o) and absmax(lp2, lp_o) < 1e-3
    assert int(seq.max()) <= cfg["V"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fc_training_mode_dropout_parity(dtype):
    from unpaired_image_captioning_amd import _lib as L
    from unpaired_image_captioning_amd.misc.criterion import LanguageModelCriterion
    cfg, W, I, Out, G, X = load_golden("fc_odd")
    model = build(cfg, W, dtype, drop=0.5)
    model.train()
    fc, labels, masks = I["fc_feats"].cuda(), I["labels"].cuda(), I["masks"].cuda()
    logp = model(fc, None, None, labels, None)
    seed = model._seed_counter
    loss = LanguageModelCriterion()(logp, labels[:, 1:], masks[:, 1:])
    loss.backward()
    lib = L.load()
    N, H, S = fc.shape[0], cfg["H"], labels.shape[1]

    def mask(t):
        out = torch.empty(N * H, device="cuda")
        L.check(lib.uic_dropout_mask(L.ptr(out), N * H, 0.5, seed, L.SITE_OUT0 + t, 0, L.stream()))
        return out.cpu().view(N, H)

    drop = {"out": torch.stack([mask(t) for t in range(S)])}
    loss_o, grads_o, logp_o = OF.xe_loss_and_grads(W, I["fc_feats"], I["labels"], I["masks"], drop)
    assert absmax(logp, logp_o) < LOGP_TOL[dtype]
    grads_close({k: p.grad for k, p in model.named_parameters()}, grads_o, GRAD_TOL[dtype])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fc_config1_shapes_vs_reference_golden(dtype):
    """BASELINE config 1: 16 images x 5 captions, seq_len 16, 2048-d fc feats, hidden 512, V+1 = 9488."""
    from unpaired_image_captioning_amd.misc.criterion import LanguageModelCriterion
    cfg, W, I, Out, G, X = load_golden("fc_cfg1")
    V, E, H, D, L = cfg["V"], cfg["E"], cfg["H"], cfg["D"], cfg["L"]
    wseed, dseed = [int(s) for s in torch.as_tensor(X["seeds"])]
    Wt = OF.init_weights(V + 1, E, H, D, seed=wseed)
    b = O.synthetic_batch(cfg["n_img"], cfg["S"], 3, D, V, L, seed=dseed)
    model = build(cfg, Wt, dtype)
    model.train()
    fc, labels, masks = b["fc_feats"].cuda(), b["labels"].cuda(), b["masks"].cuda()
    logp = model(fc, None, None, labels, None)
    assert absmax(logp[:, :, ::37], Out["logprobs_sub"]) < LOGP_TOL[dtype]
    loss = LanguageModelCriterion()(logp, labels[:, 1:], masks[:, 1:])
    assert abs(loss.item() - float(Out["loss"])) < LOGP_TOL[dtype]
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    grads_close(grads, G, GRAD_TOL[dtype])
    for k, v in X.items():
        if k.startswith("gradnorm::"):
            n = grads[k.split("::", 1)[1]].double().norm().item()
            assert abs(n - float(torch.as_tensor(v))) <= GRAD_TOL[dtype] * float(torch.as_tensor(v)), k
    if dtype == "f32":
        model.eval()
        idx = torch.arange(cfg["n_img"]) * cfg["S"]
        seq, lp = model(fc[idx], None, None, opt={"sample_max": 1}, mode="sample")
        assert torch.equal(seq.cpu(), Out["greedy_seq"])
        assert absmax(lp, Out["greedy_logp"]) < 1e-3


@pytest.mark.parametrize("name", ["fc_tiny", "fc_tiny_earlybreak", "fc_odd"])
def test_fc_beam_search_bit_exact_vs_reference_golden(name):
    """FCModel_NMT._sample_beam + CaptionModel.beam_search on the device, f32: the direct call honours the options; the
    public `_sample` call reproduces the reference's lost-options behaviour (beam 10, no constraint)."""
    cfg, W, I, Out, G, X = load_golden(name)
    model = build(cfg, W, "f32").eval()
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    fc = I["fc_feats"][idx].cuda()
    for tag in ("b3", "b2c", "b3eos", "b4ppl"):
        bs, dc, mp, eos_bias = [float(x) for x in X["beam::%s_cfg" % tag]]
        opts = {"sample_max": 1, "beam_size": int(bs), "decoding_constraint": int(dc), "max_ppl": int(mp)}
        with torch.no_grad():
            model.logit.bias[0] += eos_bias
        seq_d, lp_d = model._sample_beam(fc, None, None, opts)
        seq_p, lp_p = model(fc, None, None, opt=opts, mode="sample")
        with torch.no_grad():
            model.logit.bias[0] -= eos_bias
        assert torch.equal(seq_d.cpu(), torch.as_tensor(X["beamd::%s_seq" % tag])), tag
        assert absmax(lp_d, torch.as_tensor(X["beamd::%s_logp" % tag])) < 1e-3
        assert torch.equal(seq_p.cpu(), torch.as_tensor(X["beam::%s_seq" % tag])), tag
        assert absmax(lp_p, torch.as_tensor(X["beam::%s_logp" % tag])) < 1e-3


def _fc_sweep(n, seed):
    import numpy as np
    g = np.random.default_rng(seed)
    return [dict(V=int(g.integers(5, 500)), E=int(g.integers(1, 24)) * 8, H=int(g.integers(1, 24)) * 8, D=int(g.integers(1, 40)) * 8,
                 L=int(g.integers(1, 21)), n_img=int(g.integers(1, 12)), S=int(g.integers(1, 6)), drop=bool(g.integers(0, 2)), idx=i)
            for i in range(n)]


@pytest.mark.parametrize("cfg", _fc_sweep(24, 5), ids=lambda c: "fc%d" % c["idx"])
def test_fc_random_shape_sweep_vs_oracle(cfg):
    """24 seeded random FC-model configurations (f32): log-probs and every gradient against the oracle, with the kernels'
    own dropout masks when dropout is on."""
    from unpaired_image_captioning_amd import _lib as L
    from unpaired_image_captioning_amd.misc.criterion import LanguageModelCriterion
    W = OF.init_weights(cfg["V"] + 1, cfg["E"], cfg["H"], cfg["D"], seed=cfg["idx"])
    b = O.synthetic_batch(cfg["n_img"], cfg["S"], 2, cfg["D"], cfg["V"], cfg["L"], seed=50 + cfg["idx"])
    p = 0.5 if cfg["drop"] else 0.0
    model = build(cfg, W, "f32", drop=p)
    model.train()
    fc, labels, masks = b["fc_feats"].cuda(), b["labels"].cuda(), b["masks"].cuda()
    logp = model(fc, None, None, labels, None)
    seed = model._seed_counter
    loss = LanguageModelCriterion()(logp, labels[:, 1:], masks[:, 1:])
    loss.backward()
    drop = None
    if p:
        lib = L.load()
        N, H, S = fc.shape[0], cfg["H"], labels.shape[1]

        def mask(t):
            out = torch.empty(N * H, device="cuda")
            L.check(lib.uic_dropout_mask(L.ptr(out), N * H, p, seed, L.SITE_OUT0 + t, 0, L.stream()))
            return out.cpu().view(N, H)
        drop = {"out": torch.stack([mask(t) for t in range(S)])}
    loss_o, grads_o, logp_o = OF.xe_loss_and_grads(W, b["fc_feats"], b["labels"], b["masks"], drop)
    assert absmax(logp, logp_o) < LOGP_TOL["f32"]
    grads_close({k: q.grad for k, q in model.named_parameters()}, grads_o, 5e-3)
