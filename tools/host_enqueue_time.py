#!/usr/bin/env python3
"""Is the bench step GPU-bound or launch-bound?  Host time to ENQUEUE a step (no sync) vs wall time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer

c = bench.CFG
torch.manual_seed(1234)
tr = Trainer(bench.make_opt("bf16", 1234)); tr.build_optimizer()
batch = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
T = c["L"] + 1
t_run = tr.i2t_model._steps_to_run(batch["labels"])
den = float(batch["masks"][:, 1:T + 1].sum().item())
for _ in range(5):
    tr.train_device_batch(batch, t_run, den)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    tr.train_device_batch(batch, t_run, den)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, wall %.3f ms/step (tail after last enqueue %.3f ms)" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t1) * 1e3))
# single isolated step: enqueue time with an empty queue
for _ in range(3):
    torch.cuda.synchronize(); a = time.perf_counter(); tr.train_device_batch(batch, t_run, den); b = time.perf_counter(); torch.cuda.synchronize(); cc = time.perf_counter()
    print("  isolated step: enqueue %.3f ms, until done %.3f ms" % ((b - a) * 1e3, (cc - a) * 1e3))
