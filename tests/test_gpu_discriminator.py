"""CNN sentence discriminator (csrc/discriminator.hip, BASELINE configs[3]) against oracle/discriminator.py.  PARITY UNPINNED:
the reference has no discriminator code; the oracle restates this package's own spec, so these tests pin the kernels to the
spec, not to the reference.  Tolerances as for the other rows: logits 1e-3 (f32) / 1e-2 (bf16)."""
import argparse
import ctypes as C

import pytest
import torch

from oracle import discriminator as O

pytestmark = pytest.mark.gpu

LOGIT_TOL = {"f32": 1e-3, "bf16": 1e-2}
GRAD_TOL = {"f32": 5e-6, "bf16": 1e-2}      # per-tensor L2, relative to max(|tensor|, 1e-3 x largest tensor norm); measured 1.5e-6 / 3.2e-3


def build(V, L, E, F, widths, dtype, W, drop=0.0, seed=3):
    from unpaired_image_captioning_amd.models import SentenceDiscriminator
    opt = argparse.Namespace(vocab_size=V, seq_length=L, input_encoding_size=E, disc_num_filters=F, disc_filter_sizes=widths,
                             disc_dropout=drop, compute_dtype=dtype, seed=seed)
    m = SentenceDiscriminator(opt).cuda()
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.copy_(W[k])
    return m


def tokens_and_labels(N, L, V, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(L // 2, L + 1, (N,), generator=g)
    tok = torch.randint(1, V + 1, (N, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None]).long()
    lab = (torch.rand(N, generator=g) > 0.5).float()
    return tok, lab


def drop_mask(N, Ft, p, seed):
    from unpaired_image_captioning_amd import _lib as L_
    out = torch.empty(N * Ft, device="cuda")
    L_.check(L_.load().uic_dropout_mask(L_.ptr(out), N * Ft, p, seed, L_.SITE_DISC, 0, L_.stream()))
    return out.view(N, Ft).cpu()


CASES = [
    ("tiny", dict(V=50, L=8, E=32, F=16, widths=(1, 2, 3), N=6)),                 # odd sizes: every GEMM on the fallback paths
    ("mid", dict(V=300, L=16, E=128, F=128, widths=(1, 2, 3, 4), N=64)),          # TN / LDS-DMA eligible shapes
    ("config3", dict(V=9487, L=16, E=512, F=128, widths=(1, 2, 3, 4), N=64)),     # BASELINE configs[3]: batch 64 per GPU
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name,c", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("drop", [0.0, 0.25])
def test_forward_loss_backward_vs_oracle(name, c, dtype, drop):
    W = O.init_weights(c["V"] + 1, c["E"], c["F"], c["widths"], seed=11)
    tok, lab = tokens_and_labels(c["N"], c["L"], c["V"], seed=5)
    m = build(c["V"], c["L"], c["E"], c["F"], c["widths"], dtype, W, drop)
    m.train()
    seed = 77
    mask = drop_mask(c["N"], m.Ft, drop, seed) if drop > 0 else None
    # bf16: the oracle rounds the same stored operands (oracle/discriminator.py::forward), so both sides pick the same
    # max-over-time winners; the plain f32 oracle bounds the logits as well (north_star: 1e-2)
    ref_loss, ref_grads, ref_logits = O.loss_and_grads(W, tok, lab, c["widths"], mask, bf16=dtype == "bf16")
    if dtype == "bf16":
        exact = O.forward(W, tok, c["widths"], mask)
        assert (ref_logits - exact).abs().max().item() < LOGIT_TOL[dtype] * max(1.0, float(exact.abs().max()))
    logits = m(tok.cuda(), seed=seed)
    loss = m.bce(logits, lab.cuda())
    loss.backward()
    scale = max(1.0, float(ref_logits.abs().max()))
    assert (logits.detach().cpu() - ref_logits).abs().max().item() < LOGIT_TOL[dtype] * scale
    assert abs(loss.item() - ref_loss.item()) < LOGIT_TOL[dtype] * scale
    floor = 1e-3 * max(float(v.norm()) for v in ref_grads.values())
    worst = 0.0
    for k, p in m.named_parameters():
        r = ref_grads[k].double()
        err = ((p.grad.cpu().double() - r).norm() / max(r.norm().item(), floor)).item()
        worst = max(worst, err)
        assert err < GRAD_TOL[dtype], (k, err)
    print("disc %s %s drop %.2f: worst per-tensor gradient error %.2e" % (name, dtype, drop, worst))


def test_scores_are_eval_mode_probabilities_and_repeat():
    c = CASES[1][1]
    W = O.init_weights(c["V"] + 1, c["E"], c["F"], c["widths"], seed=2)
    tok, _ = tokens_and_labels(c["N"], c["L"], c["V"], seed=9)
    m = build(c["V"], c["L"], c["E"], c["F"], c["widths"], "f32", W, drop=0.25)
    ref = torch.sigmoid(O.forward(W, tok, c["widths"]))
    a, b = m.scores(tok.cuda()), m.scores(tok.cuda())
    assert torch.equal(a, b)
    assert (a.cpu() - ref).abs().max().item() < 1e-4
    with pytest.raises(RuntimeError):
        m.scores(tok)                                  # CPU tensor: no fallback


def test_training_separates_real_from_shuffled_captions():
    """A few Adam steps on 'real' (sorted token rows) vs 'fake' (random rows): the loss must fall well below ln 2 -- the
    forward/backward pair is a usable training signal end to end (flat arena + uic_adam_step as for the other models)."""
    from unpaired_image_captioning_amd.misc.optimizer import FlatArena
    V, L, N = 200, 12, 128
    g = torch.Generator().manual_seed(1)
    real = torch.sort(torch.randint(1, V + 1, (N, L), generator=g), dim=1)[0]
    fake = torch.randint(1, V + 1, (N, L), generator=g)
    tok = torch.cat([real, fake]).cuda()
    lab = torch.cat([torch.ones(N), torch.zeros(N)]).cuda()
    opt = argparse.Namespace(vocab_size=V, seq_length=L, input_encoding_size=64, disc_num_filters=32, disc_filter_sizes=(1, 2, 3),
                             disc_dropout=0.25, compute_dtype="bf16", seed=4)
    from unpaired_image_captioning_amd.models import SentenceDiscriminator
    torch.manual_seed(0)
    m = SentenceDiscriminator(opt).cuda()
    m.train()
    arena = FlatArena(m)
    first = None
    for step in range(1, 61):
        arena.zero_grad()
        loss = m.bce(m(tok), lab)
        loss.backward()
        arena.adam(2e-3, (0.9, 0.999), 1e-8, step)
        first = loss.item() if first is None else first
    assert first > 0.5 and loss.item() < 0.25, (first, loss.item())
    m.eval()
    s = m.scores(tok)
    assert s[:N].mean().item() > 0.8 and s[N:].mean().item() < 0.2


def test_two_forward_passes_before_one_backward_keep_their_own_activations():
    """D(real) and D(fake) evaluated separately and summed into one loss, and a no-grad scores() call in between: every
    forward pass saves its activations in a workspace of its own, so the gradients equal those of the two passes run and
    differentiated one after the other."""
    V, L, N = 120, 10, 32
    g = torch.Generator().manual_seed(3)
    real = torch.randint(1, V + 1, (N, L), generator=g).cuda()
    fake = torch.randint(1, V + 1, (N, L), generator=g).cuda()
    opt = argparse.Namespace(vocab_size=V, seq_length=L, input_encoding_size=64, disc_num_filters=32, disc_filter_sizes=(1, 2, 3),
                             disc_dropout=0.0, compute_dtype="f32", seed=4)
    from unpaired_image_captioning_amd.models import SentenceDiscriminator
    torch.manual_seed(0)
    m = SentenceDiscriminator(opt).cuda()
    m.train()
    ones, zeros = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")

    def grads_of(loss):
        m.zero_grad()
        loss.backward()
        return [p.grad.detach().clone() for p in m.parameters()]
    ga = grads_of(m.bce(m(real, seed=1), ones))
    gb = grads_of(m.bce(m(fake, seed=1), zeros))
    a = m(real, seed=1)
    b = m(fake, seed=1)                       # same batch size: would overwrite a's saved activations in a shared buffer
    m.scores(fake)
    both = grads_of(m.bce(a, ones) + m.bce(b, zeros))
    for x, y, z in zip(ga, gb, both):
        assert (z - (x + y)).abs().max().item() <= 1e-6 * max(1.0, float((x + y).abs().max())), float((z - (x + y)).abs().max())


def test_trainer_adversarial_round_runs():
    """BASELINE configs[3] in miniature: the generator samples captions, the discriminator takes one BCE step on ground-truth
    vs sampled rows, and the self-critical step adds w (D(sampled) - D(greedy)) to its reward (reward_fn = 0, so the policy
    gradient comes from the discriminator alone).  Checks the plumbing end to end: finite losses, D's parameters move, the
    generator's parameters move."""
    import numpy as np
    from conftest import load_golden
    from test_gpu_topdown import make_opt
    from unpaired_image_captioning_amd.trainer import Trainer
    cfg, W, I, Out, G, X = load_golden("topdown_tiny")
    opt = make_opt(cfg, "f32")
    opt.i2t_learning_rate = 5e-3
    opt.disc_reward_weight = 1.0
    opt.disc_num_filters, opt.disc_filter_sizes, opt.disc_learning_rate = 16, (1, 2, 3), 1e-2
    tr = Trainer(opt)
    tr.i2t_model.load_state_dict(W)
    tr.build_optimizer()
    D = tr.build_discriminator()
    data = {k: I[k].numpy() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}
    real = I["labels"][:, 1:1 + opt.seq_length]
    tr.i2t_model.eval()
    with torch.no_grad():
        fake, _ = tr.i2t_model(I["fc_feats"].cuda(), None, I["att_feats"].cuda(), I["att_masks"].cuda(), opt={'sample_max': 0}, mode='sample')
    tr.i2t_model.train()
    d0 = [p.detach().clone() for p in D.parameters()]
    g0 = tr.arena.flat.clone()
    losses = [tr.train_discriminator(real, fake) for _ in range(20)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert any(not torch.equal(a, b) for a, b in zip(d0, D.parameters()))
    loss = tr.train_self_critical(data, lambda data, s, g: np.zeros(s.shape, np.float32))
    assert np.isfinite(loss) and not torch.equal(g0, tr.arena.flat)
