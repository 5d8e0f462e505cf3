#!/usr/bin/env python3
"""Bit-reproducibility soak of the two training steps: the same step (same weights, batch and dropout seed) a few hundred times
beside an HBM-bound and an MFMA-bound neighbour on another stream; every gradient tensor must equal the first run's bit for bit.
The steps are deterministic by construction (sorted embedding gradient, fixed-order split-K reductions), so ANY difference is a
timing-dependent bug -- a counted wait that is one short, a hand-off without its fence, store data overwritten early.
    gpurun -- python tools/step_repro_soak.py [--iters 300]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--safe", action="store_true", help="the persistent kernels' placement-independent SAFE protocol (UIC_REC_SAFE)")
a = ap.parse_args()
import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer, xe_step
from unpaired_image_captioning_amd import _lib as L

side = torch.cuda.Stream()
hog_a = torch.randn(32 << 20, device="cuda")
hog_b = torch.empty_like(hog_a)
X = torch.randn(4096, 4096, device="cuda").bfloat16()


def neighbours(it):
    with torch.cuda.stream(side):
        if it % 2 == 0:
            hog_b.copy_(hog_a)
        if it % 3 != 0:
            torch.matmul(X, X)


# ---- captioner XE step (configs[1], 640 rows)
c = CFG
tr = Trainer(make_opt("bf16", 1234))
tr.build_optimizer()
model = tr.i2t_model
model.train()
if a.safe:
    model.engine.recurrence |= L.REC_SAFE
batch = {k: v.cuda() for k, v in synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=7).items()}
ref = None
bad = {}
for it in range(a.iters):
    neighbours(it)
    model._seed_counter = 4242
    loss, g = xe_step(model, batch, fused=True)
    torch.cuda.synchronize()
    if ref is None:
        ref = (loss.clone(), {k: v.clone() for k, v in g.items()})
        continue
    if not torch.equal(loss, ref[0]):
        bad["loss"] = bad.get("loss", 0) + 1
    for k, v in g.items():
        if not torch.equal(v, ref[1][k]):
            bad[k] = bad.get(k, 0) + 1
print("captioner XE step: %d runs beside busy neighbours, tensors that ever differed from run 0: %s" % (a.iters, bad or "none"), flush=True)
print("persistent status", L.persistent_status())

# ---- pivot NMT step (configs[2])
opt = argparse.Namespace(layers=2, rnn_size=512, word_vec_size=512, brnn=True, rnn_type="LSTM", dropout=0.3, input_feed=1,
                         position_encoding=False, coverage_attn=False, copy_attn=False, context_gate=None, attention_type="dot",
                         attn_transform="softmax", fertility=None, predict_fertility=False, guided_fertility=None,
                         supervised_fertility=None, lambda_coverage=0, lambda_fertility=0, lambda_exhaust=0, batch_size=64,
                         compute_dtype="bf16", seed=1, nmt_train_flag=1, i2t_train_flag=0, nmt_learning_rate=1e-3,
                         nmt_max_grad_norm=5, param_init=0.1)
tn = Trainer(opt)
V = 50004
tn.build_nmt(V, V)
gen = torch.Generator().manual_seed(3)
B, S, T = 64, 30, 32
lengths = torch.sort(torch.randint(5, S + 1, (B,), generator=gen), descending=True)[0]; lengths[0] = S
src = torch.randint(4, V, (S, B), generator=gen)
for b in range(B):
    src[lengths[b]:, b] = 0
tl = torch.randint(7, T + 1, (B,), generator=gen); tl[0] = T
tgt = torch.randint(4, V, (T, B), generator=gen); tgt[0] = 2
for b in range(B):
    tgt[tl[b] - 1, b] = 3; tgt[tl[b]:, b] = 0
nb = argparse.Namespace(src=src.unsqueeze(2).cuda(), tgt=tgt.cuda(), lengths=lengths.view(1, -1))
ref = None
bad = {}
m = tn.nmt_model
if a.safe:
    m.engine.recurrence = int(getattr(m.engine, "recurrence", 0)) | L.REC_SAFE
for it in range(a.iters):
    neighbours(it)
    m._seed_counter = 777
    tn.optim.zero_grad()
    m.unit_loss_gradient = True
    outputs, attn, dec_state, ub = tn.dp_nmt_model(nb.src, nb.tgt, nb.lengths, None)
    loss = tn.nmt_crit(None, nb, outputs, attn)
    loss.backward()
    torch.cuda.synchronize()
    g = tn.optim.nmt_arena.grad
    if ref is None:
        ref = (loss.detach().clone(), g.clone())
        continue
    if not torch.equal(loss.detach(), ref[0]):
        bad["loss"] = bad.get("loss", 0) + 1
    if not torch.equal(g, ref[1]):
        bad["grad arena"] = bad.get("grad arena", 0) + 1
        if bad["grad arena"] <= 3:
            d = (g != ref[1]).nonzero().flatten()
            print("   NMT run %d: %d differing elements, first at %d" % (it, d.numel(), int(d[0])))
print("pivot NMT step: %d runs beside busy neighbours, what ever differed from run 0: %s" % (a.iters, bad or "none"), flush=True)
print("persistent status", L.persistent_status())
