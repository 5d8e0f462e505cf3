#!/usr/bin/env python3
"""How the file reader's thread team scales (host only, no GPU): uic_loader_read over N deflated per-image att files
(np.savez_compressed, the reference's make_bu_data.py:55) with 1 .. T threads, each setting FOUR times in a row, in an order that
shows placement effects -- a team that the kernel has to spread over the CPUs first runs its first calls no faster than one
thread (round 6: the pool's workers are pinned, csrc/loader_io.hip; UIC_LOADER_NO_PIN=1 shows the unpinned behaviour).

    python3 tools/loader_threads.py [--files 64] [--max-threads 8]"""
import argparse
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unpaired_image_captioning_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=64)
ap.add_argument("--max-threads", type=int, default=min(128, os.cpu_count() or 8))
args = ap.parse_args()
lib = L.load()
n = args.files
with tempfile.TemporaryDirectory() as d:
    rng = np.random.default_rng(0)
    for i in range(n):
        np.savez_compressed(os.path.join(d, "%d" % i), feat=np.maximum(rng.standard_normal((36, 2048)).astype(np.float32), 0) * 3)
    paths = [os.path.join(d, "%d.npz" % i).encode() for i in range(n)]
    print("%d files of %d bytes (%d raw); UIC_LOADER_NO_PIN=%s" % (n, os.path.getsize(paths[0]), 36 * 2048 * 4, os.environ.get("UIC_LOADER_NO_PIN")))
    arr = (C.c_char_p * n)(*paths)
    info = np.empty((n, 6), dtype=np.int64)
    L.check(lib.uic_loader_scan(arr, n, b"feat", info.ctypes.data, 1))
    buf = np.empty((n, 36 * 2048), dtype=np.float32)
    dst = (C.c_void_p * n)(*[buf.ctypes.data + i * 36 * 2048 * 4 for i in range(n)])
    counts = [1]
    while counts[-1] * 2 <= args.max_threads:
        counts.append(counts[-1] * 2)
    for nt in counts + counts[1:3]:
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            L.check(lib.uic_loader_read(arr, n, info.ctypes.data, dst, nt))
            ts.append((time.perf_counter() - t0) * 1e3)
        print("threads %3d: %s ms   (%.0f MB/s per thread at the best)" % (nt, "  ".join("%6.2f" % t for t in ts), n * 36 * 2048 * 4 / min(ts) / nt / 1e3))
