#!/usr/bin/env python3
"""Trainer.train fed from HOST arrays (numpy batch of DataLoader.get_batch, features once per image), with and without the
per-step counts of unmasked positions (opt.live_positions), alternating blocks in ONE process: what the list costs on the host
against what it saves on the device in the loop that synchronises every step (P/trainer.py:172).
    python tools/host_fed_ab.py [--steps 30] [--rounds 6]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30); ap.add_argument("--rounds", type=int, default=6)
a = ap.parse_args()
import torch
import bench
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer

c = bench.CFG
torch.manual_seed(1234)
tr = Trainer(bench.make_opt("bf16", 1234)); tr.build_optimizer()
tr.opt.seq_per_img = c["S"]
batch = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
data = {k: v.cpu().numpy() for k, v in batch.items()}
data2 = dict(data)
res = {}
for r in range(a.rounds + 1):
    for live in (0, 1):
        tr.opt.live_positions = live
        for mode in ("plain", "next_data"):
            cur, nxt = data, data2
            def step():
                global cur, nxt
                if mode == "plain":
                    tr.train(cur)
                else:
                    tr.train(cur, next_data=nxt)
                    cur, nxt = nxt, cur
            for _ in range(3):
                step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            if r:
                res.setdefault((mode, live), []).append((time.perf_counter() - t0) / a.steps * 1e3)
for (mode, live), v in sorted(res.items()):
    v = sorted(v)
    print("%-10s live_positions=%d  median %.3f ms  min %.3f  max %.3f" % (mode, live, v[len(v) // 2], v[0], v[-1]))
