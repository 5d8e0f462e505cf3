#!/usr/bin/env python3
"""Host time of the fused training step's C call (uic_topdown_xe_train_step enqueues ~200 launches and returns without
synchronising) next to the GPU time of the step: if the two are close the step is launch-bound on the host.
    python tools/host_time.py [--steps 30]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--early", action="store_true", help="UIC_REC_EARLY_GRADS: the order of the gradient work that finishes most bytes early")
ap.add_argument("--all-positions", action="store_true", help="without the list of unmasked positions (uic_topdown_batch.live_rows)")
args = ap.parse_args()

import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd import models, trainer
from unpaired_image_captioning_amd.synthetic import synthetic_batch

lib = L.load()
c = CFG
model = models.setup(make_opt(args.dtype, 1234)).cuda()
model.train()
if args.early:
    model.engine.recurrence |= L.REC_EARLY_GRADS
batch = trainer.Trainer.attach_live({k: v.cuda() for k, v in synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1).items()})
if args.all_positions:
    batch = {k: v for k, v in batch.items() if not k.startswith("live_")}
real = lib.uic_topdown_xe_train_step
acc = {"n": 0, "t": 0.0}


def timed(*a):
    t0 = time.perf_counter()
    r = real(*a)
    acc["t"] += time.perf_counter() - t0
    acc["n"] += 1
    return r


lib.uic_topdown_xe_train_step = timed
for phase in ("warm", "run"):
    acc["n"], acc["t"] = 0, 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5 if phase == "warm" else args.steps):
        loss, grads = trainer.xe_step(model, batch)
        loss.item()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
print("steps %d: wall %.3f ms/step, host time inside uic_topdown_xe_train_step %.3f ms/call (%d calls)" %
      (args.steps, wall / args.steps * 1e3, acc["t"] / max(acc["n"], 1) * 1e3, acc["n"]))

# where the step's time goes on the GPU: timing marks recorded by the fused step itself (no profiler attached)
import ctypes as C
lib.uic_topdown_xe_train_step = real
L.check(lib.uic_topdown_step_marks(1, None))
tot = [0.0] * L.STEP_MARKS
for _ in range(args.steps):
    loss, grads = trainer.xe_step(model, batch)
    loss.item()
    ms = (C.c_float * L.STEP_MARKS)()
    L.check(lib.uic_topdown_step_marks(1, ms))
    tot = [a + b for a, b in zip(tot, ms)]
L.check(lib.uic_topdown_step_marks(0, None))
names = ["start", "prologue done", "recurrence done", "side: logit layer done", "BPTT starts", "BPTT done", "side: recurrent wgrads done",
         "main tail done", "side tail done", "joined", "side: logit gradients final"]
for n, v in zip(names, tot):
    print("   %-30s %8.3f ms" % (n, v / args.steps))

# when each gradient group of uic_topdown_grad_ready_wait is final, and how many bytes it holds (f32 gradient arena)
T = trainer.Trainer
nbytes = {k: v.numel() * 4 for k, v in model.param_dict().items()}
first = [k for k in nbytes if k.startswith(T.FIRST_GRADS)]
g1 = [k for k in nbytes if k.startswith(T.LSTM_W_GRADS_EARLY if args.early else T.LSTM_W_GRADS)]
late = [k for k in nbytes if k.startswith(T.LATE_GRADS)]
g2 = [k for k in nbytes if k not in first and k not in g1 and k not in late]
end = tot[9] / args.steps
print("gradient groups of uic_topdown_grad_ready_wait (%s order): bytes, ready at, before the end of the step" % ("UIC_REC_EARLY_GRADS" if args.early else "default"))
for label, ks, mark in (("group 0  logit.*", first, 10), ("group 1  LSTM weights%s" % (" + embedding + att_lstm.weight_ih" if args.early else ""), g1, 6),
                        ("group 2  rest of the early group", g2, 8), ("tail     att_embed / ctx2att / attention", late, 7)):
    t = tot[mark] / args.steps
    print("   %-52s %6.2f MB   %7.3f ms   %6.3f ms" % (label, sum(nbytes[k] for k in ks) / 1e6, t, end - t))
lastwin = sum(sum(nbytes[k] for k in ks) for ks, mark in ((first, 10), (g1, 6), (g2, 8), (late, 7)) if end - tot[mark] / args.steps < 0.1)
print("   bytes that become final in the step's last 0.1 ms: %.2f MB of %.2f MB" % (lastwin / 1e6, sum(nbytes.values()) / 1e6))
