"""Soak of uic_linear_f32a (csrc/gemm_pp.hip, f32 A operand): thousands of launches beside an HBM-bound and / or an MFMA-bound
neighbour on another stream, output and bf16 image compared bit for bit with cast + GEMM every time.  This is what found the
store-data hazard of the first form (the image's global_store as inline asm: the compiler did not know the registers were VMEM
store data and reused them too early -- 32-64 wrong elements in 1 % of the launches, only with a busy neighbour).
    gpurun -- python tools/gemm_f32a_soak.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unpaired_image_captioning_amd import _lib as L
lib = L.load()
torch.manual_seed(1)
bad = 0
tot = 0
for (M, N, K) in ((23040, 512, 2048), (4608, 512, 2048), (1000, 1028, 640), (2880, 512, 384)):
    A = torch.randn(M, K, device="cuda") * 3
    Ab = A.bfloat16()
    B = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    L.check(lib.uic_linear(1, M, N, K, L.ptr(Ab), K, L.ptr(B), K, L.ptr(ref), N, L.ptr(bias), 1 | 0x400, L.stream()))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    hog_a = torch.randn(64 << 20, device="cuda")
    hog_b = torch.empty_like(hog_a)
    X = torch.randn(4096, 4096, device="cuda").bfloat16()
    iters = 400 if M > 10000 else 800
    for it in range(iters):
        mode = it % 4
        with torch.cuda.stream(side):
            if mode == 1: hog_b.copy_(hog_a)            # HBM-bound neighbour
            elif mode == 2: torch.matmul(X, X)          # MFMA-bound neighbour
            elif mode == 3: hog_b.copy_(hog_a); torch.matmul(X, X)
        C = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        img = torch.full((M, K), 7.0, device="cuda", dtype=torch.bfloat16)
        L.check(lib.uic_linear_f32a(M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(C), N, L.ptr(bias), 1, L.ptr(img), K, L.stream()))
        torch.cuda.synchronize()
        ok = bool(torch.equal(C, ref)) and bool(torch.equal(img, Ab))
        tot += 1
        if not ok:
            bad += 1
            if bad < 5: print("MISMATCH", (M, N, K), it, mode, int((C != ref).sum()), int((img != Ab).sum()))
    print((M, N, K), "done", iters, "iterations; mismatches so far", bad, flush=True)
print("f32-A GEMM soak: %d launches beside HBM-bound / MFMA-bound neighbours, %d mismatches" % (tot, bad))
