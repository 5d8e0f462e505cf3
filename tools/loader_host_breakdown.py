#!/usr/bin/env python3
"""Where the host time of DataLoader.get_batch goes inside a training loop (tools/loader_bench.py's data set)."""
import os, sys, time, random, shutil, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from unpaired_image_captioning_amd.misc.dataloader import dataloader as M
from dataset_files import loader_opt, write_dataset

n, R, D, N = 128, 36, 2048, 512
rng = np.random.default_rng(0)
tmp = tempfile.mkdtemp(prefix="uic_lhb_", dir="/tmp")
try:
    att = [np.abs(rng.standard_normal((R, D))).astype(np.float32) for _ in range(N)]
    box = [np.hstack([xy, xy + 10]).astype(np.float32) for xy in (rng.uniform(0, 200, (R, 2)) for _ in range(N))]
    labels = rng.integers(1, 9487, (N * 5, 16)).astype(np.uint32)
    ends = np.arange(5, N * 5 + 1, 5)
    lp = write_dataset(tmp, att, box, [x.mean(0) for x in att], [(480, 640)] * N, list(range(N)), labels, ends - 4, ends, 9487, label_format="npz")
    loader = M.DataLoader(loader_opt(tmp, lp, n, 5, D, D + 5, 1, 1, 1))
    T = {}
    def timed(name, fn):
        def w(*a, **k):
            t0 = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
        return w
    loader._staged = timed("staged(wait for read-ahead)", loader._staged)
    loader._ship = timed("ship(H2D + kernel enqueue)", loader._ship)
    loader._read_ahead = timed("read_ahead submit", loader._read_ahead)
    stage0 = loader._stage
    loader._stage = timed("stage (read-ahead thread)", stage0)
    random.seed(0)
    for _ in range(3): loader.get_batch("train")
    T.clear(); K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        loader.get_batch("train"); time.sleep(0.004)          # a training step's worth of time for the read-ahead
    total = time.perf_counter() - t0 - K * 0.004
    print("get_batch total %.2f ms" % (total / K * 1e3))
    for k, v in T.items(): print("  %-32s %.2f ms" % (k, v / K * 1e3))
finally:
    shutil.rmtree(tmp)
