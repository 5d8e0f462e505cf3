#!/usr/bin/env python3
"""Persistent decode launch (rnn_persist.hip, decode mode) against the per-step launch chain at BASELINE configs[1] size:
agreement of tokens / log-probs (greedy, multinomial, forced replay), pass times, phase stamps of the decode tail.
    python tools/decode_probe.py [--dbg] [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dbg", action="store_true")
ap.add_argument("--rows", type=int, default=0, help="images (default: the bench config's 128)")
args = ap.parse_args()

import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd import models
from unpaired_image_captioning_amd.synthetic import synthetic_batch

c = CFG
n_img = args.rows or c["n_img"]
model = models.setup(make_opt("bf16", 1234)).cuda()
# a logit layer with some contrast, or every row decodes the same flat distribution
with torch.no_grad():
    lw = model.logit.weight if isinstance(model.logit, torch.nn.Linear) else model.logit[-1].weight
    lw.mul_(25.0)
batch = {k: v.cuda() for k, v in synthetic_batch(n_img, c["S"], c["R"], c["D"], c["V"], c["L"], seed=1).items()}
fc, att, am = batch["fc_feats"], batch["att_feats"], batch.get("att_masks")
eng = model.engine
Lsteps = c["L"]


def run(mode_flags, sample_max, training, forced=None, keep=False):
    eng.recurrence = mode_flags
    pd = {k: v.detach() for k, v in model.param_dict().items()}
    out = eng.sample(pd, fc, att, am, Lsteps, sample_max=sample_max, seed=777, forced=forced, training=training, keep_forward=keep)
    if keep:
        eng.release(out[2])
    return out[0], out[1]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


st0 = L.persistent_status()
for name, sm, tr, keep in (("greedy, eval", 1, False, False), ("multinomial, eval", 0, False, False),
                           ("multinomial, train mode (dropout), kept forward", 0, True, True)):
    seq_c, lp_c = run(L.REC_FWD_CHAIN, sm, tr, keep=keep)
    seq_p, lp_p = run(0, sm, tr, keep=keep)
    same = (seq_c == seq_p).all(1).float().mean().item()
    live = seq_c == seq_p
    print("%-50s rows with identical captions %.3f   tokens identical %.4f   max |d logp| where identical %.2e   mean length %.2f / %.2f" %
          (name, same, live.float().mean().item(), (lp_c - lp_p)[live].abs().max().item(), (seq_c > 0).sum(1).float().mean().item(),
           (seq_p > 0).sum(1).float().mean().item()))
    # forced replay of the chain's tokens through the persistent launch: log-probs of the same tokens
    if not sm:
        seq_f, lp_f = run(0, 0, tr, forced=seq_c, keep=keep)
        alive = torch.ones_like(seq_c, dtype=torch.bool)
        alive[:, 1:] = (seq_c[:, :-1] > 0).cumprod(1).bool()
        print("%-50s forced replay: tokens equal %s   max |d logp| (live positions) %.2e" %
              ("", bool((seq_f == seq_c).all()), (lp_f - lp_c)[alive].abs().max().item()))
    t_c = timeit(lambda: run(L.REC_FWD_CHAIN, sm, tr, keep=keep), args.iters)
    t_p = timeit(lambda: run(0, sm, tr, keep=keep), args.iters)
    print("%-50s per-step chain %.3f ms   persistent %.3f ms   (%d rows x %d steps)" % ("", t_c, t_p, seq_c.shape[0], Lsteps))
st1 = L.persistent_status()
print("persistent launches (XCD-local, SAFE): %d, %d   timeouts: %d" % (st1[1] - st0[1], st1[2] - st0[2], st1[0]))

if args.dbg:
    names = ["att_lstm", "bar", "h2att", "bar", "attention", "bar", "lang_lstm", "bar+logits", "bar+sample", "bar+xt gemm"]
    for label, sm, tr in (("greedy, eval", 1, False), ("multinomial, train mode, kept forward", 0, True)):
        eng.recurrence = L.REC_STAMPS
        pd = {k: v.detach() for k, v in model.param_dict().items()}
        seq, lp, ws = eng.sample(pd, fc, att, am, Lsteps, sample_max=sm, seed=777, training=tr, keep_forward=True)
        torch.cuda.synchronize()
        T = Lsteps + 1
        dbg = eng.workspace_tensor(ws, "rnn_dbg", (256, T, 16), torch.int64).cpu().double()
        eng.release(ws)
        eng.recurrence = 0
        d = dbg[:, 1:Lsteps - 1, :]
        ph = [(d[:, :, i + 1] - d[:, :, i]).mean().item() / 100.0 for i in range(10)]
        step = (dbg[:, 2:Lsteps, 0] - dbg[:, 1:Lsteps - 1, 0]).mean().item() / 100.0
        print("decode step phases, %s (us, mean over workgroups and steps 1..%d; step %.2f us):" % (label, Lsteps - 2, step))
        print("   " + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, ph)))
        sub = [("barrier wait", 7, 15), ("A staging", 15, 11), ("GEMM", 11, 12), ("bias, stores, partials", 12, 14), ("combine", 14, 8)]
        print("   logits phase: " + "  ".join("%s %.2f" % (n, (d[:, :, b] - d[:, :, a]).mean().item() / 100.0) for n, a, b in sub))
