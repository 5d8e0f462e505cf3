#!/usr/bin/env python3
"""Launch ONLY the attention-step forward kernel at the bench shapes (N=640, R=36, A=H=512), for PMC passes:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT/fetch -- python3 /root/repo/tools/attn_kernel_only.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d OUT/write -- python3 /root/repo/tools/attn_kernel_only.py
    python3 tools/pmc_traffic.py OUT attn_fwd      # -> per-launch HBM bytes (FETCH_SIZE doubled per the gfx950 note)

Each launch gets its own p_att/att pair out of a pool larger than the 256 MiB Infinity Cache, so the counters see
HBM-side streaming reads as in a training step (where a step's other kernels evict them), not on-die re-reads.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unpaired_image_captioning_amd import _lib as L

N, R, A, H = 640, 36, 512, 512
dtype_id = L.BF16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else L.F32
td = L.TORCH_DTYPE[dtype_id]
lib = L.load()
POOL = 8                                   # 8 x (23.6 + 23.6) MB bf16 = 377 MB > 256 MiB
g = torch.Generator(device="cuda").manual_seed(1)
att_h = torch.randn(N, A, device="cuda", generator=g)
p_att = [torch.randn(N, R, A, device="cuda", generator=g).to(td) for _ in range(POOL)]
att = [torch.randn(N, R, H, device="cuda", generator=g).abs().to(td) for _ in range(POOL)]
w = torch.randn(A, device="cuda", generator=g) * 0.05
b = torch.zeros(1, device="cuda")
alpha = torch.empty(N, R, device="cuda")
ctx = torch.empty(N, H, device="cuda", dtype=td)
torch.cuda.synchronize()
for i in range(64):
    k = i % POOL
    L.check(lib.uic_attention_fwd(dtype_id, N, R, A, H, L.ptr(att_h), L.ptr(p_att[k]), L.ptr(att[k]), L.ptr(w), L.ptr(b),
                                  None, L.ptr(alpha), L.ptr(ctx), L.stream()))
torch.cuda.synchronize()
print("done", float(ctx.float().abs().mean()))
