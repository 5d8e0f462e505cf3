"""GPU parity of the pivot NMT step (Embeddings + packed bi-LSTM Encoder, input-feed
StackedLSTM Decoder with dot GlobalAttention, generator + NMTCriterion) against golden vectors from the reference's own
modules (tests/golden/nmt_*.npz) and against the CPU oracle on seeded inputs, through the C ABI
(uic_nmt_forward_loss / uic_nmt_backward) behind the reference-shaped python surface."""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import nmt as ON
from test_gpu_topdown import absmax, grads_close

pytestmark = pytest.mark.gpu

OUT_TOL = {"f32": 1e-3, "bf16": 2e-2}       # decoder outputs are tanh-bounded; attention weights are probabilities
GRAD_TOL = {"f32": 2e-3, "bf16": 1e-1}      # f32: max-entry error; bf16: L2 error (see grads_close)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    W = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    I = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in::")}
    Out = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out::")}
    G = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad::")}
    layers, H, B, S, T, Vs, Vt = [int(x) for x in z["cfg"]]
    cfg = dict(layers=layers, H=H, W=W["encoder.embeddings.word_lut.weight"].shape[1], B=B, S=S, T=T, Vs=Vs, Vt=Vt)
    return cfg, W, I, Out, G


def make_opt(cfg, dtype, dropout=0.0, seed=0):
    return argparse.Namespace(layers=cfg["layers"], rnn_size=cfg["H"], word_vec_size=cfg["W"], brnn=True, rnn_type="LSTM",
                              dropout=dropout, input_feed=1, position_encoding=False, coverage_attn=False, copy_attn=False,
                              context_gate=None, attention_type="dot", attn_transform="softmax", fertility=None,
                              predict_fertility=False, guided_fertility=None, supervised_fertility=None,
                              lambda_coverage=0, lambda_fertility=0, lambda_exhaust=0, batch_size=cfg["B"], gpus=[0],
                              compute_dtype=dtype, seed=seed)


def build(cfg, W, dtype, dropout=0.0, seed=0):
    """The construction sequence of P/trainer.py:80-94."""
    import torch.nn as nn
    from unpaired_image_captioning_amd.models import NMT_Models
    from unpaired_image_captioning_amd.misc import criterion
    opt = make_opt(cfg, dtype, dropout, seed)
    enc = NMT_Models.Encoder(opt, cfg["Vs"])
    dec = NMT_Models.Decoder(opt, cfg["Vt"])
    model = NMT_Models.NMTModel(opt, enc, dec, None, None, False)
    gen = nn.Sequential(nn.Linear(opt.rnn_size, cfg["Vt"]), nn.LogSoftmax(dim=-1))
    model.generator = gen
    if W is not None:
        assert list(model.state_dict().keys()) == list(W.keys())
        model.load_state_dict(W)
    model.cuda()
    crit = criterion.NMT_loss(opt, gen, criterion.NMTCriterion(cfg["Vt"], opt))
    return model, crit


def run(model, crit, I):
    Batch = argparse.Namespace
    batch = Batch(src=I["src"].cuda(), tgt=I["tgt"].cuda(), lengths=I["lengths"])
    outputs, attns, dec_state, ub = model(batch.src, batch.tgt, batch.lengths, None)
    loss = crit(None, batch, outputs, attns)
    return outputs, attns["std"], loss


def test_nmt_state_dict_matches_reference_contract():
    cfg, W, I, Out, G = load("nmt_tiny")
    model, _ = build(cfg, None, "f32")
    sd = model.state_dict()
    assert list(sd.keys()) == list(W.keys())
    for k in W:
        assert tuple(sd[k].shape) == tuple(W[k].shape), k


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["nmt_tiny", "nmt_tiny_1layer", "nmt_odd"])
def test_nmt_forward_loss_backward_vs_reference_golden(name, dtype):
    cfg, W, I, Out, G = load(name)
    model, crit = build(cfg, W, dtype)
    model.train()
    outputs, attn, loss = run(model, crit, I)
    assert absmax(outputs, Out["outputs"]) < OUT_TOL[dtype]
    assert absmax(attn, Out["attn"]) < OUT_TOL[dtype]
    n_words = int(Out["num_words"])
    assert abs(loss.item() - float(Out["loss"])) < OUT_TOL[dtype] * n_words
    assert crit.report_stats.n_words == n_words
    if dtype == "f32":
        assert crit.report_stats.n_correct == int(Out["num_correct"])
    assert abs(crit.report_stats.ppl() - float(np.exp(min(float(Out["loss"]) / n_words, 100)))) < 5e-2 * crit.report_stats.ppl()
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    assert all(g is not None for g in grads.values())
    grads_close(grads, G, GRAD_TOL[dtype])
    # nn.Embedding(padding_idx=PAD): no gradient reaches the PAD rows
    assert grads["encoder.embeddings.word_lut.weight"][0].abs().max().item() == 0
    assert grads["decoder.embeddings.word_lut.weight"][0].abs().max().item() == 0


def test_nmt_context_and_workspace_api_f32():
    """Raw C-ABI call: context (memory bank) output, padded source positions exactly zero (pad_packed_sequence)."""
    import ctypes as C
    from unpaired_image_captioning_amd import _lib as L
    cfg, W, I, Out, G = load("nmt_odd")
    lib = L.load()
    d = L.NmtDims(B=cfg["B"], S=cfg["S"], T=cfg["T"], H=cfg["H"], W=cfg["W"], layers=cfg["layers"], Vs=cfg["Vs"], Vt=cfg["Vt"],
                  dtype=L.F32, drop_p=0.0)
    nbytes = lib.uic_nmt_workspace_bytes(C.byref(d))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    Wd = {k: v.cuda().contiguous() for k, v in W.items()}
    w = L.nmt_weights(Wd, cfg["layers"])
    src = I["src"][:, :, 0].contiguous().cuda()
    tgt = I["tgt"].contiguous().cuda()
    lens_h = [int(x) for x in I["lengths"].reshape(-1).tolist()]
    lens = (C.c_int32 * cfg["B"])(*lens_h)
    lens_d = torch.tensor(lens_h, dtype=torch.int32, device="cuda")
    loss = torch.zeros(1, device="cuda")
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    context = torch.empty(cfg["S"], cfg["B"], cfg["H"], device="cuda")
    L.check(lib.uic_nmt_forward_loss(C.byref(d), C.byref(w), L.ptr(src), lens, L.ptr(lens_d), L.ptr(tgt), 1, 0, L.ptr(ws),
                                     L.ptr(loss), L.ptr(stats), None, None, L.ptr(context), L.stream()), "nmt_forward_loss")
    assert absmax(context, Out["context"]) < 1e-4
    for b, n in enumerate(lens_h):
        assert context[n:, b].abs().max().item() == 0 if n < cfg["S"] else True
    assert abs(loss.item() - float(Out["loss"])) < 1e-2
    assert stats.tolist() == [int(Out["num_correct"]), int(Out["num_words"])]
    # unsorted lengths are rejected like pack_padded_sequence does
    bad = (C.c_int32 * cfg["B"])(*sorted(lens_h))
    if sorted(lens_h) != lens_h:
        rc = lib.uic_nmt_forward_loss(C.byref(d), C.byref(w), L.ptr(src), bad, L.ptr(lens_d), L.ptr(tgt), 1, 0, L.ptr(ws),
                                      L.ptr(loss), None, None, None, None, L.stream())
        assert rc != 0 and b"sorted" in lib.uic_last_error_string()


def synthetic(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    B, S, T = cfg["B"], cfg["S"], cfg["T"]
    lengths = torch.sort(torch.randint(max(1, S // 3), S + 1, (B,), generator=g), descending=True)[0]
    lengths[0] = S
    src = torch.randint(4, cfg["Vs"], (S, B), generator=g)
    for b in range(B):
        src[lengths[b]:, b] = 0
    tl = torch.randint(max(3, T // 2), T + 1, (B,), generator=g)
    tl[0] = T
    tgt = torch.randint(4, cfg["Vt"], (T, B), generator=g)
    tgt[0] = 2                                     # BOS
    for b in range(B):
        tgt[tl[b] - 1, b] = 3                      # EOS
        tgt[tl[b]:, b] = 0
    return dict(src=src.unsqueeze(2), tgt=tgt, lengths=lengths.view(1, -1))


def random_weights(cfg, seed, scale=0.1):
    model, _ = build(cfg, None, "f32")
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.rand(v.shape, generator=g) * 2 - 1) * scale for k, v in model.state_dict().items()}


MID = dict(layers=2, H=256, W=192, B=24, S=21, T=17, Vs=900, Vt=1100)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_nmt_mid_size_vs_oracle(dtype):
    cfg = MID
    W = random_weights(cfg, 11)
    I = synthetic(cfg, 5)
    ref_loss, ref_g, aux = ON.loss_and_grads(W, I["src"], I["tgt"], I["lengths"])
    model, crit = build(cfg, W, dtype)
    model.train()
    outputs, attn, loss = run(model, crit, I)
    assert absmax(outputs, aux["outputs"]) < OUT_TOL[dtype]
    assert absmax(attn, aux["attn"]) < OUT_TOL[dtype]
    assert abs(loss.item() - ref_loss.item()) < OUT_TOL[dtype] * aux["num_words"]
    loss.backward()
    grads_close({k: p.grad for k, p in model.named_parameters()}, ref_g, GRAD_TOL[dtype])


def _nmt_sweep(n, seed):
    g = np.random.default_rng(seed)
    out = []
    for i in range(n):
        H = int(g.integers(1, 12)) * 16                       # rnn_size (even halves per direction, multiples of 8)
        out.append(dict(layers=int(g.integers(1, 4)), H=H, W=int(g.integers(1, 12)) * 8, B=int(g.integers(1, 20)),
                        S=int(g.integers(1, 16)), T=int(g.integers(3, 14)), Vs=int(g.integers(8, 300)), Vt=int(g.integers(8, 300)), idx=i))
    return out


@pytest.mark.parametrize("cfg", _nmt_sweep(24, 77), ids=lambda c: "nmt%d" % c["idx"])
def test_nmt_random_shape_sweep_vs_oracle(cfg):
    """24 seeded random NMT configurations (1-3 layers, odd batch sizes and lengths incl. one-word sources, hidden sizes
    that are multiples of 16 only): f32 forward, loss, counters and every gradient against the oracle."""
    W = random_weights(cfg, 100 + cfg["idx"])
    I = synthetic(cfg, 200 + cfg["idx"])
    ref_loss, ref_g, aux = ON.loss_and_grads(W, I["src"], I["tgt"], I["lengths"])
    model, crit = build(cfg, W, "f32")
    model.train()
    outputs, attn, loss = run(model, crit, I)
    assert absmax(outputs, aux["outputs"]) < OUT_TOL["f32"]
    assert absmax(attn, aux["attn"]) < OUT_TOL["f32"]
    assert abs(loss.item() - ref_loss.item()) < OUT_TOL["f32"] * max(1, aux["num_words"])
    assert crit.report_stats.n_words == aux["num_words"] and crit.report_stats.n_correct == aux["num_correct"]
    loss.backward()
    grads_close({k: p.grad for k, p in model.named_parameters()}, ref_g, 5e-3)


def test_nmt_large_vocabulary_criterion_bf16_vs_oracle():
    """A 20 003-word target vocabulary (80 KB logits rows: past the LDS-staged criterion kernel, onto the two-pass
    running-max kernel that also feeds NMT_loss.score's counters): loss, accuracy counters and gradients against the
    oracle; a generator bias makes some arg-maxes hit their targets."""
    cfg = dict(layers=1, H=64, W=64, B=6, S=7, T=6, Vs=50, Vt=20003)
    W = random_weights(cfg, 31)
    I = synthetic(cfg, 9)
    tgt = I["tgt"]
    W["generator.0.bias"][int(tgt[1, 0])] += 4.0                      # the first target of sentence 0 becomes the arg-max ...
    W["generator.0.bias"][20002] += 2.0                               # ... and the last vocabulary index a frequent runner-up
    ref_loss, ref_g, aux = ON.loss_and_grads(W, I["src"], I["tgt"], I["lengths"])
    model, crit = build(cfg, W, "bf16")
    model.train()
    outputs, attn, loss = run(model, crit, I)
    assert abs(loss.item() - ref_loss.item()) < OUT_TOL["bf16"] * aux["num_words"]
    st = crit.report_stats
    assert st.n_words == aux["num_words"] and abs(st.n_correct - aux["num_correct"]) <= 1 and aux["num_correct"] >= 1
    loss.backward()
    grads_close({k: p.grad for k, p in model.named_parameters()}, ref_g, GRAD_TOL["bf16"])


def export_masks(cfg, seed, p):
    """The multiplicative masks the kernels use, in the oracle's `drop` layout."""
    from unpaired_image_captioning_amd import _lib as L
    lib = L.load()
    B, S, Td, H, NL = cfg["B"], cfg["S"], cfg["T"] - 1, cfg["H"], cfg["layers"]

    def mask(site, n):
        out = torch.empty(n, device="cuda")
        L.check(lib.uic_dropout_mask(L.ptr(out), n, p, seed, site, 0, L.stream()))
        return out.cpu()

    enc = [mask(L.SITE_NMT_ENC0 + l, S * B * H).view(S, B, H) for l in range(NL - 1)]
    dec = [torch.stack([mask(L.SITE_NMT_DEC0 + l * 256 + t, B * H).view(B, H) for t in range(Td)]) for l in range(NL - 1)]
    out = torch.stack([mask(L.SITE_NMT_OUT0 + t, B * H).view(B, H) for t in range(Td)])
    return dict(enc=enc, dec=dec, out=out)


@pytest.mark.parametrize("name", ["nmt_tiny", "nmt_odd"])
def test_nmt_dropout_matches_oracle_with_exported_masks(name):
    cfg, W, I, Out, G = load(name)
    p = 0.3
    model, crit = build(cfg, W, "f32", dropout=p, seed=77)
    model.train()
    outputs, attn, loss = run(model, crit, I)
    drop = export_masks(cfg, model._last_seed, p)
    ref_loss, ref_g, aux = ON.loss_and_grads(W, I["src"], I["tgt"], I["lengths"], drop)
    assert absmax(outputs, aux["outputs"]) < 1e-3
    assert abs(loss.item() - ref_loss.item()) < 1e-3 * aux["num_words"]
    assert abs(loss.item() - float(Out["loss"])) > 1e-3          # dropout did change the result
    loss.backward()
    grads_close({k: q.grad for k, q in model.named_parameters()}, ref_g, 2e-3)
    # eval mode: no dropout, equals the golden
    model.eval()
    outputs, attn, loss = run(model, crit, I)
    assert absmax(outputs, Out["outputs"]) < 1e-3


def test_nmt_training_reduces_loss_bf16():
    cfg = dict(layers=2, H=128, W=128, B=16, S=12, T=11, Vs=200, Vt=220)
    I = synthetic(cfg, 3)
    torch.manual_seed(0)
    model, crit = build(cfg, None, "bf16", dropout=0.1, seed=5)
    for p in model.parameters():
        p.data.uniform_(-0.1, 0.1)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    model.train()
    first = None
    for step in range(60):
        opt.zero_grad()
        outputs, attn, loss = run(model, crit, I)
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
    assert loss.item() < 0.7 * first, (first, loss.item())
    assert crit.report_stats.accuracy() >= 0.0


# ---------------------------------------------------------------- Optim on the NMT model
def load_optim_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg, W, I, Out, G = load(name)
    final = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("final::")}
    lr, max_norm, warmup, noam, alpha, beta, eps = [float(x) for x in z["optcfg"]]
    return cfg, W, I, Out, final, dict(lr=lr, max_norm=max_norm, warmup=int(warmup), noam=bool(noam), alpha=alpha, beta=beta, eps=eps)


@pytest.mark.parametrize("name", ["nmt_optim_clip", "nmt_optim_noam"])
def test_nmt_optim_trajectory_vs_reference_golden(name):
    """4 steps of zero_grad / forward / NMT_loss / backward / Optim.step on one batch, against the same loop run with the
    reference's own Optim (clip_grad_norm 5 resp. noam schedule) -- losses, gradient norms, learning rates and the
    final weights."""
    from unpaired_image_captioning_amd.misc.optimizer import Optim
    cfg, W, I, Out, final, oc = load_optim_case(name)
    model, crit = build(cfg, W, "f32")
    o = model.opt
    o.nmt_train_flag, o.i2t_train_flag = 1, 0
    o.nmt_optim, o.nmt_learning_rate, o.nmt_max_grad_norm = "adam", oc["lr"], oc["max_norm"]
    o.nmt_decay_method, o.nmt_warmup_steps = ("noam" if oc["noam"] else ""), oc["warmup"]
    o.nmt_optim_alpha, o.nmt_optim_beta, o.nmt_optim_epsilon = oc["alpha"], oc["beta"], oc["eps"]
    optim = Optim(o)
    optim.set_parameters(None, model)
    model.train()
    losses, norms, lrs = [], [], []
    for it in range(len(Out["losses"])):
        optim.zero_grad()
        outputs, attn, loss = run(model, crit, I)
        loss.backward()
        norms.append(optim.nmt_arena.grad_norm())
        optim.step()
        losses.append(loss.item())
        lrs.append(optim.nmt_current_lr)
    np.testing.assert_allclose(lrs, Out["lrs"].numpy(), rtol=1e-6)
    np.testing.assert_allclose(losses, Out["losses"].numpy(), rtol=2e-4)
    np.testing.assert_allclose(norms, Out["grad_norms"].numpy(), rtol=5e-3)
    sd = model.state_dict()
    for k, ref in final.items():
        moved = (ref - W[k]).norm().item()
        err = (sd[k].cpu() - ref).norm().item()
        assert err <= 2e-2 * moved + 1e-6, (k, err, moved)


def test_trainer_nmt_half_runs_and_saves(tmp_path):
    """Trainer.build_nmt + train_nmt (P/trainer.py:80-94,175-193) and model_nmt.pth with the reference's keys."""
    from unpaired_image_captioning_amd.trainer import Trainer
    cfg = dict(layers=2, H=64, W=64, B=8, S=10, T=9, Vs=120, Vt=130)
    o = make_opt(cfg, "bf16", dropout=0.1, seed=3)
    o.nmt_train_flag, o.i2t_train_flag, o.checkpoint_path = 1, 0, str(tmp_path)
    o.nmt_learning_rate, o.nmt_max_grad_norm, o.param_init = 5e-3, 5, 0.1
    tr = Trainer(o)
    tr.build_nmt(cfg["Vs"], cfg["Vt"])
    I = synthetic(cfg, 9)
    batch = argparse.Namespace(src=I["src"].cuda(), tgt=I["tgt"].cuda(), lengths=I["lengths"])
    first = tr.train_nmt(batch)
    for _ in range(30):
        last = tr.train_nmt(batch)
    assert last < 0.8 * first, (first, last)
    assert tr.nmt_train_ppl > 1.0 and 0.0 <= tr.nmt_train_acc <= 100.0
    tr.save_models()
    sd = torch.load(os.path.join(str(tmp_path), "model_nmt.pth"))
    assert list(sd.keys()) == tr.nmt_model.param_names


def test_trainer_nmt_without_the_arena_fill_equals_the_eager_zero_grad(tmp_path):
    """Trainer.train_nmt skips the fill of the gradient arena (Optim.zero_grad(nmt_direct=True)): the in-place backward pass
    overwrites every gradient, embedding tables included.  Three steps with the arena POISONED before each backward pass must
    give the losses and weights of three steps with the eager fill, bit for bit; a backward pass that takes the accumulating
    path after a lazy zero_grad must clear the arena itself."""
    from unpaired_image_captioning_amd.trainer import Trainer
    cfg = dict(layers=2, H=64, W=64, B=8, S=10, T=9, Vs=120, Vt=130)
    I = synthetic(cfg, 9)
    batch = argparse.Namespace(src=I["src"].cuda(), tgt=I["tgt"].cuda(), lengths=I["lengths"])

    def fresh():
        o = make_opt(cfg, "bf16", dropout=0.1, seed=3)
        o.nmt_train_flag, o.i2t_train_flag, o.checkpoint_path = 1, 0, str(tmp_path)
        o.nmt_learning_rate, o.nmt_max_grad_norm, o.param_init = 5e-3, 5, 0.1
        tr = Trainer(o)
        tr.build_nmt(cfg["Vs"], cfg["Vt"])
        return tr

    eager = fresh()
    W0 = {k: v.clone() for k, v in eager.nmt_model.state_dict().items()}
    z = eager.optim.zero_grad
    eager.optim.zero_grad = lambda nmt_direct=False: z()                 # always the fill
    ref = [eager.train_nmt(batch) for _ in range(3)]
    lazy = fresh()
    lazy.nmt_model.load_state_dict(W0)
    zl = lazy.optim.zero_grad

    def poisoned(nmt_direct=False):
        zl(nmt_direct=nmt_direct)
        if nmt_direct:
            g = lazy.optim.nmt_arena.grad
            base = g.data_ptr()
            for v in lazy.optim.nmt_arena.grad_views.values():          # (the padding between tensors stays zero, as in real use)
                v.fill_(float("nan"))
    lazy.optim.zero_grad = poisoned
    got = [lazy.train_nmt(batch) for _ in range(3)]
    assert got == ref
    assert torch.equal(lazy.optim.nmt_arena.flat, eager.optim.nmt_arena.flat)
    # the accumulating path after a lazy zero_grad: the arena is cleared before autograd adds into it
    lazy.optim.zero_grad = zl
    lazy.optim.zero_grad(nmt_direct=True)
    for v in lazy.optim.nmt_arena.grad_views.values():
        v.fill_(1e6)
    lazy.nmt_model.unit_loss_gradient = False
    outputs, attn, _, _ = lazy.dp_nmt_model(batch.src, batch.tgt, batch.lengths, None)
    lazy.nmt_crit(None, batch, outputs, attn).backward()
    torch.cuda.synchronize()
    assert float(lazy.optim.nmt_arena.grad.abs().max()) < 1e5


def test_trainer_nmt_step_after_a_timeout_is_skipped_on_the_device_and_raises(tmp_path):
    """The status word the persistent pivot kernels (csrc/nmt_persist.hip) set when a bounded spin gives up, forced here: the
    clipped Adam update must be skipped on the device (uic_adam_step_clip_guarded), train_nmt must raise with the step counters
    rolled back, and the next call must train exactly as if the bad step had never been issued."""
    from unpaired_image_captioning_amd import _lib as Lb
    from unpaired_image_captioning_amd.trainer import Trainer
    cfg = dict(layers=2, H=64, W=64, B=8, S=10, T=9, Vs=120, Vt=130)
    I = synthetic(cfg, 9)
    batch = argparse.Namespace(src=I["src"].cuda(), tgt=I["tgt"].cuda(), lengths=I["lengths"])

    def fresh():
        o = make_opt(cfg, "bf16", dropout=0.0, seed=3)
        o.nmt_train_flag, o.i2t_train_flag, o.checkpoint_path = 1, 0, str(tmp_path)
        o.nmt_learning_rate, o.nmt_max_grad_norm, o.param_init = 5e-3, 5, 0.1
        tr = Trainer(o)
        tr.build_nmt(cfg["Vs"], cfg["Vt"])
        return tr

    ref = fresh()
    W0 = {k: v.clone() for k, v in ref.nmt_model.state_dict().items()}
    ref_losses = [ref.train_nmt(batch) for _ in range(2)]
    tr = fresh()
    tr.nmt_model.load_state_dict(W0)
    assert tr.train_nmt(batch) == ref_losses[0]
    ar = tr.optim.nmt_arena
    snap = (ar.flat.clone(), ar.exp_avg.clone(), ar.exp_avg_sq.clone(), tr.optim._nmt_steps, tr.optim._step)
    Lb.status_words()[0] = 0x7
    torch.cuda.synchronize()
    try:
        with pytest.raises(RuntimeError, match="timed out"):
            tr.train_nmt(batch)
        assert int(Lb.status_words()[0]) == 0                       # cleared by the raise
    finally:
        Lb.status_words().zero_()
    assert torch.equal(ar.flat, snap[0]) and torch.equal(ar.exp_avg, snap[1]) and torch.equal(ar.exp_avg_sq, snap[2])
    assert (tr.optim._nmt_steps, tr.optim._step) == snap[3:]
    assert tr.train_nmt(batch) == ref_losses[1]
    assert torch.equal(ar.flat, ref.optim.nmt_arena.flat)


@pytest.mark.parametrize("name", ["nmt_translate_tiny", "nmt_translate_odd", "nmt_translate_1layer", "nmt_translate_long"])
def test_nmt_translate_batch_vs_reference_golden(name):
    """NMTModel.translateBatch + onmt Beam (beam 15, up to 100 steps) on the device, f32: hypotheses token for token, the
    number of decoder steps, final scores, attention of the winning hypotheses."""
    cfg, W, I, Out, G = load(name)
    cfg["T"] = 2
    model, _ = build(cfg, W, "f32")
    model.eval()
    batch = argparse.Namespace(src=I["src"].cuda(), batchSize=cfg["B"])
    allHyp, allScores, allAttn, gold = model.translateBatch(batch)
    ref = Out["hyp"]
    assert len(allHyp) == cfg["B"] and len(allHyp[0][0]) == ref.shape[1]
    assert torch.equal(torch.tensor([h[0] for h in allHyp]), ref), (allHyp, ref)
    got_scores = torch.stack([s_[0] for s_ in allScores]).cpu().double()
    assert (got_scores - Out["scores"]).abs().max().item() < 2e-3 * max(1.0, Out["scores"].abs().max().item())
    for b in range(cfg["B"]):
        a = allAttn[b][0].cpu()
        assert (a - Out["attn"][b, :, :a.shape[1]]).abs().max().item() < 1e-3
        assert Out["attn"][b, :, a.shape[1]:].abs().max().item() == 0 if a.shape[1] < cfg["S"] else True
    assert float(gold.abs().sum()) == 0


def test_nmt_translate_bf16_mid_size_vs_oracle():
    """bf16, 16 sentences, 500-word vocabulary: the device's hypotheses re-scored by the oracle (bf16 can swap near-tied
    candidates, so compare the final beam scores instead of demanding identical tokens)."""
    cfg = dict(layers=2, H=128, W=128, B=16, S=12, T=2, Vs=400, Vt=500)
    W = random_weights(cfg, 21, scale=0.3)
    W["generator.0.bias"][3] += 3.0
    I = synthetic(dict(cfg, T=6), 8)
    model, _ = build(cfg, W, "bf16")
    model.eval()
    batch = argparse.Namespace(src=I["src"].cuda(), batchSize=cfg["B"])
    allHyp, allScores, allAttn, gold = model.translateBatch(batch)
    hyp_o, scores_o, attn_o = ON.translate_batch(W, I["src"])
    n_same = sum(int(allHyp[b][0] == [int(t) for t in hyp_o[b]]) for b in range(cfg["B"])) if len(allHyp[0][0]) == hyp_o.shape[1] else 0
    got = torch.stack([s_[0] for s_ in allScores]).cpu()
    assert (got - scores_o).abs().max().item() < 0.15 * max(1.0, scores_o.abs().max().item()), (got, scores_o)
    assert n_same >= cfg["B"] // 2, n_same


@pytest.mark.parametrize("cfg", _nmt_sweep(10, 4242), ids=lambda c: "tr%d" % c["idx"])
def test_nmt_translate_random_sweep_vs_oracle(cfg):
    """10 seeded random configurations through translateBatch (beam 15, f32): the oracle's hypotheses token for token for
    most sentences (random weights produce near-tied candidates that may swap) and its final scores for all of them."""
    cfg = dict(cfg, Vt=max(cfg["Vt"], 20), T=2)
    W = random_weights(cfg, 300 + cfg["idx"], scale=0.3)
    W["generator.0.bias"][3] += 2.5                                   # EOS likely enough for sentences to end at varied steps
    I = synthetic(dict(cfg, T=6), 400 + cfg["idx"])
    model, _ = build(cfg, W, "f32")
    model.eval()
    batch = argparse.Namespace(src=I["src"].cuda(), batchSize=cfg["B"])
    allHyp, allScores, allAttn, gold = model.translateBatch(batch, max_steps=12)
    hyp_o, scores_o, attn_o = ON.translate_batch(W, I["src"], max_steps=12)
    got = torch.stack([s_[0] for s_ in allScores]).cpu()
    assert (got - scores_o).abs().max().item() < 2e-3 * max(1.0, scores_o.abs().max().item()), (got, scores_o)
    if len(allHyp[0][0]) == hyp_o.shape[1]:
        n_same = sum(int(allHyp[b][0] == [int(t) for t in hyp_o[b]]) for b in range(cfg["B"]))
        assert n_same >= (cfg["B"] + 1) // 2, (n_same, cfg["B"])


# ---------------------------------------------------------------- BASELINE configs[2] at its real size
def recipe_weights(cfg, seed, scale):
    """tests/golden/make_golden_nmt.py::recipe_weights on this side: U(-scale, scale) from one torch CPU generator, tensor
    after tensor in state_dict order."""
    model, _ = build(cfg, None, "f32")
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.rand(v.shape, generator=g) * 2 - 1) * scale for k, v in model.state_dict().items()}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_nmt_real_width_vs_reference_golden(dtype):
    """2 layers, rnn_size 512, vocabularies of 50 004 words (the widths of BASELINE configs[2]) on four sentences: the
    reference's own NMTModel / NMTCriterion produced outputs, attention, loss, accuracy counters, the norm of every gradient,
    decoder.attn.linear_in.weight's gradient and the generator / embedding gradient rows of the words that occur
    (tests/golden/nmt_real_b4.npz; the 300 MB of weights are rebuilt from the recipe stored with it)."""
    z = np.load(os.path.join(GOLDEN, "nmt_real_b4.npz"))
    layers, H, B, S, T, Vs, Vt = [int(x) for x in z["cfg"]]
    cfg = dict(layers=layers, H=H, W=H, B=B, S=S, T=T, Vs=Vs, Vt=Vt)
    seed, scale = int(z["recipe"][0]), float(z["recipe"][1])
    W = recipe_weights(cfg, seed, scale)
    assert list(W.keys()) == [str(k) for k in z["keys"]]
    I = dict(src=torch.from_numpy(z["in::src"]), tgt=torch.from_numpy(z["in::tgt"]), lengths=torch.from_numpy(z["in::lengths"]))
    model, crit = build(cfg, W, dtype)
    model.train()
    outputs, attn, loss = run(model, crit, I)
    assert absmax(outputs, torch.from_numpy(z["out::outputs"])) < OUT_TOL[dtype]
    assert absmax(attn, torch.from_numpy(z["out::attn"])) < OUT_TOL[dtype]
    nw = int(z["out::num_words"])
    assert abs(loss.item() - float(z["out::loss"])) < OUT_TOL[dtype] * nw
    st = crit.report_stats
    assert st.n_words == nw and abs(st.n_correct - int(z["out::num_correct"])) <= (0 if dtype == "f32" else 1)
    loss.backward()
    g = {k: p.grad.detach().float().cpu().double() for k, p in model.named_parameters()}
    big = max(float(z["gnorm::" + k]) for k in g)
    tol = 2e-4 if dtype == "f32" else 4e-2
    for k, v in g.items():
        ref = float(z["gnorm::" + k])
        assert abs(v.norm().item() - ref) <= tol * max(ref, 1e-3 * big), (k, v.norm().item(), ref)

    def close(got, ref, what):
        ref = torch.from_numpy(ref).double()
        assert ((got - ref).norm() / max(ref.norm().item(), 1e-3 * big)).item() < tol, what
    close(g["decoder.attn.linear_in.weight"], z["grad::decoder.attn.linear_in.weight"], "linear_in")
    close(g["generator.0.weight"][torch.from_numpy(z["rows::generator"])], z["gradrows::generator.0.weight"], "generator rows")
    close(g["encoder.embeddings.word_lut.weight"][torch.from_numpy(z["rows::enc_lut"])], z["gradrows::encoder.embeddings.word_lut.weight"], "enc rows")
    close(g["decoder.embeddings.word_lut.weight"][torch.from_numpy(z["rows::dec_lut"])], z["gradrows::decoder.embeddings.word_lut.weight"], "dec rows")


CONFIGS2 = dict(layers=2, H=512, W=512, B=64, S=30, T=31, Vs=50004, Vt=50004)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_nmt_configs2_full_size_vs_oracle(dtype):
    """BASELINE configs[2] at full size (batch 64, 2 layers, 512, vocabularies 50 004, lengths up to 30): decoder outputs,
    attention, loss, accuracy counters and every gradient against oracle/nmt.py on the same inputs (the oracle's 1 920 x 50 004
    generator GEMM and its backward take a few seconds on the host)."""
    cfg = CONFIGS2
    W = recipe_weights(cfg, 5, 0.08)
    I = synthetic(cfg, 6)
    nt = torch.get_num_threads()
    torch.set_num_threads(min(16, nt))
    ref_loss, ref_g, aux = ON.loss_and_grads(W, I["src"], I["tgt"], I["lengths"])
    torch.set_num_threads(nt)
    model, crit = build(cfg, W, dtype)
    model.train()
    outputs, attn, loss = run(model, crit, I)
    assert absmax(outputs, aux["outputs"]) < OUT_TOL[dtype]
    assert absmax(attn, aux["attn"]) < OUT_TOL[dtype]
    assert abs(loss.item() - ref_loss.item()) < OUT_TOL[dtype] * aux["num_words"]
    st = crit.report_stats
    assert st.n_words == aux["num_words"] and abs(st.n_correct - aux["num_correct"]) <= (0 if dtype == "f32" else 2)
    loss.backward()
    big = max(float(v.norm()) for v in ref_g.values())
    tol = 2e-5 if dtype == "f32" else 6e-2            # measured on MI355X: 2.3e-6 / 3.7e-2
    worst = 0.0
    for k, p in model.named_parameters():
        g, r = p.grad.detach().float().cpu().double(), ref_g[k].double()
        err = ((g - r).norm() / max(r.norm().item(), 1e-3 * big)).item()
        worst = max(worst, err)
        assert err < tol, (k, err)
    print("configs[2] full size %s: loss %.4f (oracle %.4f), worst per-tensor L2 gradient error %.3e" % (dtype, loss.item(), ref_loss.item(), worst))


@pytest.mark.parametrize("shape", [dict(B=64, S=30, T=31), dict(B=5, S=7, T=9), dict(B=128, S=64, T=6), dict(B=37, S=1, T=4), dict(B=161, S=20, T=5)])
def test_persistent_launches_equal_the_launch_chain(shape):
    """bf16, rnn_size 512: the packed bidirectional encoder layers (P/models/NMT_Models.py:95-135), NMT_Models.Decoder.forward's
    target-step loop (:228-262) and their BPTT as ONE persistent launch each (csrc/nmt_persist.hip: 2 + 1 forward, 1 + 2 backward)
    against the per-step launches they replace (2 per source step and layer, layers + 2 per target step, 7 + 4 backward) --
    decoder outputs, attention, loss and
    every gradient (the backward pass reads what either form left in the workspace: gates, states, dropped copies, contexts), with
    training-mode dropout 0.3 (same sites, same masks): configs[2]'s batch (8 rows per XCD group, the weight-stationary kernel), 5
    rows (groups without rows), 128 rows (full 16-row tiles) with the longest source the kernels take (64), a one-word source, and
    161 rows (the generic kernel: 21 rows per group, weights re-read every step).  The two forms differ by
    bf16 summation order only; the placement-independent SAFE protocol must give the persistent launch's results bit for bit."""
    from unpaired_image_captioning_amd import _lib as Lb
    cfg = dict(layers=2, H=512, W=512, Vs=300, Vt=260, **shape)
    W = random_weights(cfg, 11, 0.08)
    I = synthetic(cfg, 12)
    res = {}
    for mode, flags in (("chain", Lb.REC_FWD_CHAIN), ("persistent", 0), ("safe", Lb.REC_SAFE)):
        model, crit = build(cfg, W, "bf16", dropout=0.3, seed=5)
        model.train()
        model.engine.recurrence = flags
        before = Lb.persistent_status()
        outputs, attn, loss = run(model, crit, I)
        after = Lb.persistent_status()
        launches = (after[1] - before[1], after[2] - before[2])
        n = 3 if cfg["B"] <= 128 else 1              # forward: one launch per encoder layer (batch <= 128) + the decoder's
        assert launches == {"chain": (0, 0), "persistent": (n, 0), "safe": (0, n)}[mode], (mode, launches)
        loss.backward()
        res[mode] = (outputs.detach().float().cpu(), attn.detach().float().cpu(), loss.item(),
                     {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()})
    oc, ac, lc, gc = res["chain"]
    op, ap, lp, gp = res["persistent"]
    os_, as_, ls, gs = res["safe"]
    assert torch.equal(op, os_) and torch.equal(ap, as_) and lp == ls
    for k in gp:
        assert torch.equal(gp[k], gs[k]), k
    nw = int((I["tgt"][1:] != 0).sum())
    assert (op - oc).abs().max().item() < OUT_TOL["bf16"] and (ap - ac).abs().max().item() < OUT_TOL["bf16"]
    assert abs(lp - lc) < 2e-3 * nw
    big = max(float(v.norm()) for v in gc.values())
    for k in gc:
        err = ((gp[k].double() - gc[k].double()).norm() / max(gc[k].double().norm().item(), 1e-3 * big)).item()
        assert err < 5e-2, (k, err)
