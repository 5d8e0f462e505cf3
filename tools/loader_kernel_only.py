#!/usr/bin/env python3
"""Launch ONLY the batch-assembly kernel at the bench shape (128 images x 36 regions x 2048 + 5 box features), for PMC
passes (tools/loader_pmc.sh).  Every launch gets its own input / output set out of a pool larger than the 256 MiB
Infinity Cache, so the counters see HBM-side traffic."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unpaired_image_captioning_amd import _lib as L

n, R, D, ld = 128, 36, 2048, 2176
lib = L.load()
POOL = 6                                   # 6 x (37.7 + 40.1) MB = 467 MB
start = torch.arange(0, (n + 1) * R, R, dtype=torch.int32, device="cuda")
slot = torch.arange(n, dtype=torch.int32, device="cuda")
hw = torch.tensor([[480., 640., 480. * 640.]] * n, device="cuda")
sets = []
for _ in range(POOL):
    xy = torch.rand(n * R, 2, device="cuda") * 200
    sets.append((torch.rand(n * R, D, device="cuda"), torch.cat([xy, xy + 10 + torch.rand(n * R, 2, device="cuda") * 200], 1).contiguous(),
                 torch.empty(n, R, ld, device="cuda"), torch.empty(n, R, device="cuda")))
torch.cuda.synchronize()
for i in range(48):
    feat, box, out, m = sets[i % POOL]
    L.check(lib.uic_att_batch_assemble(L.ptr(feat), L.ptr(box), L.ptr(start), L.ptr(hw), L.ptr(slot), n, D, 1, 1, R, ld, L.ptr(out),
                                       L.ptr(m), L.stream()))
torch.cuda.synchronize()
print("done", float(sets[0][2].abs().mean()))
