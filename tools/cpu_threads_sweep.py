#!/usr/bin/env python3
"""How many host threads should bench.py's cpu_baseline use?  Times one oracle training step per thread count."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads())
for n in [int(x) for x in (sys.argv[1:] or ["8", "16", "32", "64"])]:
    torch.set_num_threads(n)
    t0 = time.perf_counter()
    r = bench.cpu_baseline()
    print(n, r["value"], "captions/s  (%.1f s)" % (time.perf_counter() - t0), flush=True)
