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
batch = {k: v.cuda() for k, v in synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1).items()}
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
tot = [0.0] * 10
for _ in range(args.steps):
    loss, grads = trainer.xe_step(model, batch)
    loss.item()
    ms = (C.c_float * 10)()
    L.check(lib.uic_topdown_step_marks(1, ms))
    tot = [a + b for a, b in zip(tot, ms)]
L.check(lib.uic_topdown_step_marks(0, None))
names = ["start", "prologue done", "recurrence done", "side: logit layer done", "BPTT starts", "BPTT done", "side: recurrent wgrads done",
         "main tail done", "side tail done", "joined"]
for n, v in zip(names, tot):
    print("   %-30s %8.3f ms" % (n, v / args.steps))
