#!/usr/bin/env python3
"""A/B of launch orders of the fused training step in ONE process on ONE device (cdna guide rule 24): the settings --
uic_topdown_dims.recurrence values, incl. the measurement knobs of csrc/uic_common.h (UIC_KNOB_*) -- alternate in blocks of
steps; per setting the wall time per step (median and min over the rounds) and the step's own timing marks.
    python tools/ab_knobs.py 0 0x100 0x200 [--steps 20] [--rounds 6]"""
import argparse
import os
import sys
import time
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=6)
args = ap.parse_args()

import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd import models, trainer
from unpaired_image_captioning_amd.synthetic import synthetic_batch

lib = L.load()
c = CFG
model = models.setup(make_opt("bf16", 1234)).cuda()
model.train()
batch = trainer.Trainer.attach_live({k: v.cuda() for k, v in synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1).items()})
settings = [int(s, 0) for s in args.settings]
names = ["start", "prologue", "recurrence", "logit layer", "BPTT starts", "BPTT done", "rec wgrads", "main tail", "side tail", "joined", "logit grads"]


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loss, grads = trainer.xe_step(model, batch)
        loss.item()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


wall = {s: [] for s in settings}
marks = {s: [0.0] * L.STEP_MARKS for s in settings}
for s in settings:
    model.engine.recurrence = s
    run(5)
for r in range(args.rounds):
    for s in settings:
        model.engine.recurrence = s
        run(3)
        wall[s].append(run(args.steps))
        # the step's own marks (a few steps; recording them costs ~10 us of events)
        L.check(lib.uic_topdown_step_marks(1, None))
        for _ in range(4):
            loss, grads = trainer.xe_step(model, batch)
            loss.item()
            ms = (C.c_float * L.STEP_MARKS)()
            L.check(lib.uic_topdown_step_marks(1, ms))
            marks[s] = [a + b / (4 * args.rounds) for a, b in zip(marks[s], ms)]
        L.check(lib.uic_topdown_step_marks(0, None))
print("%-8s %9s %9s   " % ("setting", "median ms", "min ms") + " ".join("%11s" % n for n in names[1:]))
for s in settings:
    w = sorted(wall[s])
    print("%-8s %9.3f %9.3f   " % (hex(s), w[len(w) // 2], w[0]) + " ".join("%11.3f" % v for v in marks[s][1:]))
