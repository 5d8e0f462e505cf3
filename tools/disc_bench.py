#!/usr/bin/env python3
"""Sentence-discriminator step (BASELINE configs[3] shapes: 64 real + 64 generated caption rows per GPU, seq_len 16, vocabulary
9487, E 512, 4 widths x 128 filters): forward + BCE + backward + Adam through the module surface, bf16 and f32; the CPU
oracle of the same spec beside it.  Parity of this row is UNPINNED (no reference discriminator exists).
    python tools/disc_bench.py [--rows 128] [--steps 50]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=128)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--no-cpu", action="store_true")
args = ap.parse_args()

import torch
from unpaired_image_captioning_amd.misc.optimizer import FlatArena
from unpaired_image_captioning_amd.models import SentenceDiscriminator

V, L = 9487, 16
g = torch.Generator().manual_seed(0)
tok = torch.randint(1, V + 1, (args.rows, L), generator=g).cuda()
lab = (torch.rand(args.rows, generator=g) > 0.5).float().cuda()
for dtype in ("bf16", "f32"):
    opt = argparse.Namespace(vocab_size=V, seq_length=L, input_encoding_size=512, disc_num_filters=128, disc_filter_sizes=(1, 2, 3, 4),
                             disc_dropout=0.25, compute_dtype=dtype, seed=1)
    torch.manual_seed(0)
    D = SentenceDiscriminator(opt).cuda()
    D.train()
    arena = FlatArena(D)

    def step(i):
        arena.zero_grad()
        loss = D.bce(D(tok), lab)
        loss.backward()
        arena.adam(1e-4, (0.9, 0.999), 1e-8, i)
        return loss
    for i in range(1, 6):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6, 6 + args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print("disc %s: %.3f ms/step, %.0f sentences/s (%d rows), loss %.4f" % (dtype, dt * 1e3, args.rows / dt, args.rows, loss.item()))
if not args.no_cpu:
    from oracle import discriminator as O
    nt = min(16, os.cpu_count() or 1)
    torch.set_num_threads(nt)
    W = O.init_weights(V + 1, 512, 128, (1, 2, 3, 4), seed=0)
    tc, lc = tok.cpu(), lab.cpu()
    O.loss_and_grads(W, tc, lc, (1, 2, 3, 4))
    t0 = time.perf_counter()
    for _ in range(5):
        O.loss_and_grads(W, tc, lc, (1, 2, 3, 4))
    dt = (time.perf_counter() - t0) / 5
    print("disc CPU oracle (%d threads, forward + backward, no optimizer): %.1f ms/step, %.0f sentences/s" % (nt, dt * 1e3, args.rows / dt))
