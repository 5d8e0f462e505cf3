#!/usr/bin/env python3
"""The live-position list (uic_topdown_batch.live_rows) against the step over all positions, in ONE process on one box:
  1. two trainers from the same weights train on the benchmark batch side by side; their losses are printed step by step
     (they differ by summation order only, so they drift apart slowly);
  2. blocks of steps alternate between a batch with the list and the same batch without it: median / min ms per step.
    python tools/live_ab.py [--steps 40] [--rounds 8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--trajectory", type=int, default=60)
args = ap.parse_args()

import torch
import bench
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer

c = bench.CFG
batch = {k: v.cuda() for k, v in synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234).items()}
live = Trainer.attach_live(dict(batch))
T = batch["labels"].shape[1] - 1
den = float(batch["masks"][:, 1:T + 1].sum().item())
print("positions %d, live %d (%.1f%%)" % (batch["labels"].shape[0] * T, int(live["live_count"].sum()),
                                          100.0 * live["live_count"].sum() / (batch["labels"].shape[0] * T)))

if args.trajectory:
    trs = []
    for _ in range(2):
        torch.manual_seed(1234)
        tr = Trainer(bench.make_opt("bf16", 1234))
        tr.build_optimizer()
        trs.append(tr)
    t_run = trs[0].i2t_model._steps_to_run(batch["labels"])
    for i in range(args.trajectory):
        la = trs[0].train_device_batch(batch, t_run, den).item()
        lb = trs[1].train_device_batch(live, t_run, den).item()
        if i < 10 or i % 10 == 9:
            print("step %3d  all %.6f  live %.6f  diff %+.2e" % (i, la, lb, lb - la))
    del trs

tr = Trainer(bench.make_opt("bf16", 1234))
tr.build_optimizer()
t_run = tr.i2t_model._steps_to_run(batch["labels"])
res = {"all": [], "live": []}
for r in range(args.rounds + 1):
    for name, b in (("all", batch), ("live", live)):
        for _ in range(5):
            tr.train_device_batch(b, t_run, den)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.train_device_batch(b, t_run, den)
        torch.cuda.synchronize()
        if r:
            res[name].append((time.perf_counter() - t0) / args.steps * 1e3)
for name, v in res.items():
    v = sorted(v)
    print("%-6s median %.4f ms  min %.4f ms  max %.4f ms" % (name, v[len(v) // 2], v[0], v[-1]))
