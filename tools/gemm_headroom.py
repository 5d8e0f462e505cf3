#!/usr/bin/env python3
"""How far the hot path's GEMM shapes are from what the vendor library reaches on the same GPU: torch.matmul (hipBLASLt /
rocBLAS) vs libuic_hip's kernels, bf16 operands, one HIP event pair per launch, trimmed mean.  Diagnostic only."""
import os
import sys
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unpaired_image_captioning_amd import _lib as L

lib = L.load()


def timeit(fn, iters=24):
    for _ in range(4):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(d[iters // 8: iters - iters // 8]) / (iters - 2 * (iters // 8)) * 1e3   # us


g = torch.Generator(device="cuda").manual_seed(3)
print("%-44s %9s %9s %9s %9s %9s %9s %9s %9s" % ("NT shape  C[M,N] = A[M,K] B[N,K]^T", "lib us", "lib TF/s", "uic us", "uic TF/s", "128x128", "pp 256", "pp 192", "pp 128"))
for name, M, N, K in [("logit fwd chunk", 2560, 9488, 512), ("att_embed fwd", 23040, 512, 2048), ("ctx2att fwd", 23040, 512, 512),
                      ("Gx batched input GEMM", 10880, 2048, 1024), ("d xt", 10880, 512, 2048), ("logit dX chunk", 2560, 512, 9536),
                      ("BPTT d x2 (one step)", 640, 1536, 2048), ("BPTT d x1 (one step)", 640, 1024, 2048), ("BPTT h2att (one step)", 640, 512, 512)]:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    ldc = (N + 63) // 64 * 64
    Cu = torch.empty(M, ldc, device="cuda", dtype=torch.float32)
    bias = torch.zeros(N, device="cuda")
    t_lib = timeit(lambda: torch.matmul(A, B.t()))
    t_uic = timeit(lambda: L.check(lib.uic_linear(1, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(Cu), ldc, L.ptr(bias), 4, L.stream())))
    fl = 2.0 * M * N * K
    rows = torch.arange(0, M, max(1, M // 512), device="cuda")
    ref = A[rows].float() @ B.float().t()
    err = float((Cu[rows][:, :N] - ref).abs().max() / ref.abs().max())
    # the two large-GEMM kernels forced per call (UIC_GEMM_FORCE_128 / _256): which one the dispatcher should pick for the shape
    t128 = timeit(lambda: L.check(lib.uic_linear(1, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(Cu), ldc, L.ptr(bias), 4 | 0x100, L.stream())))
    tpp = [float("nan")] * 3
    if K % 128 == 0 and N % 4 == 0:
        for i, force in enumerate((0x200, 0x400, 0x800)):
            tpp[i] = timeit(lambda: L.check(lib.uic_linear(1, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(Cu), ldc, L.ptr(bias), 4 | force, L.stream())))
            err = max(err, float((Cu[rows][:, :N] - ref).abs().max() / ref.abs().max()))
    print("%-44s %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f   max rel err %.1e" % ("%s %dx%dx%d" % (name, M, N, K), t_lib, fl / t_lib / 1e6, t_uic, fl / t_uic / 1e6, t128, tpp[0], tpp[1], tpp[2], err))
# att_embed on the loader's f32 features: cast pass + bf16 GEMM against the GEMM that rounds its f32 A operand itself
# (uic_linear_f32a, with the bf16 image stored for the weight gradient)
print("%-44s %10s %10s %10s %10s" % ("f32 input (att_embed 23040x512x2048)", "cast us", "bf16 GEMM", "sum", "f32-A GEMM"))
for name, M, N, K in [("att_embed fwd", 23040, 512, 2048), ("att_embed fwd, 128 images", 4608, 512, 2048)]:
    Af = torch.randn(M, K, device="cuda", generator=g)
    Ab = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    t_cast = timeit(lambda: L.check(lib.uic_cast_from_f32(1, L.ptr(Af), L.ptr(Ab), M * K, L.stream())))
    # (forced onto the ping-pong kernel: the f32-A form shares its accumulation order, so the outputs must be identical)
    t_gemm = timeit(lambda: L.check(lib.uic_linear(1, M, N, K, L.ptr(Ab), K, L.ptr(B), K, L.ptr(Cb), N, L.ptr(bias), 1 | 0x400, L.stream())))
    ref = Cb.clone()
    t_fold = timeit(lambda: L.check(lib.uic_linear_f32a(M, N, K, L.ptr(Af), K, L.ptr(B), K, L.ptr(Cb), N, L.ptr(bias), 1, L.ptr(Ab), K, L.stream())))
    same = bool(torch.equal(ref, Cb))
    t_noimg = timeit(lambda: L.check(lib.uic_linear_f32a(M, N, K, L.ptr(Af), K, L.ptr(B), K, L.ptr(Cb), N, L.ptr(bias), 1, None, K, L.stream())))
    print("%-44s %10.1f %10.1f %10.1f %10.1f   identical %s   (without the bf16 image: %.1f us)" % ("%s %dx%dx%d" % (name, M, N, K), t_cast, t_gemm, t_cast + t_gemm, t_fold, same, t_noimg))
print("%-44s %10s %10s" % ("NT split-K partials (uic_linear_partials)", "uic us", "uic TF/s"))
for name, M, N, K, sk in [("BPTT d x2, 4 slices", 640, 1536, 2048, 4), ("BPTT d x2, 2 slices", 640, 1536, 2048, 2), ("BPTT d x2, 8 slices", 640, 1536, 2048, 8),
                          ("BPTT d x1, 4 slices", 640, 1024, 2048, 4), ("BPTT d x1, 8 slices", 640, 1024, 2048, 8),
                          ("logit dX chunk, 3 slices", 2560, 512, 9536, 3), ("logit dX chunk, 5 slices", 2560, 512, 9536, 5)]:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    slab = torch.empty(sk, M, N, device="cuda", dtype=torch.float32)
    t_uic = timeit(lambda: L.check(lib.uic_linear_partials(1, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(slab), sk, L.stream())))
    rows = torch.arange(0, M, max(1, M // 256), device="cuda")
    ref = A[rows].float() @ B.float().t()
    err = float((slab.sum(0)[rows] - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * K
    print("%-44s %10.1f %10.1f   max rel err %.1e" % ("%s %dx%dx%d" % (name, M, N, K), t_uic, fl / t_uic / 1e6, err))
print("%-44s %10s %10s %10s %10s" % ("TN shape  C[M,N] = A[K,M]^T B[K,N]", "lib us", "lib TF/s", "uic us", "uic TF/s"))
wsb = 64 << 20
wsp = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for name, M, N, K in [("logit dW", 9488, 512, 10880), ("LSTM dW chunk", 2048, 1536, 2560), ("LSTM dW all steps", 2048, 1536, 10880),
                      ("att_embed dW (folded)", 512, 2048, 4608), ("ctx2att dW", 512, 512, 23040)]:
    A = torch.randn(K, M, device="cuda", generator=g).bfloat16()
    B = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    t_lib = timeit(lambda: torch.matmul(A.t(), B))
    dW = torch.empty(M, N, device="cuda", dtype=torch.float32)
    t_uic = timeit(lambda: L.check(lib.uic_linear_wgrad(1, M, N, K, L.ptr(A), M, L.ptr(B), N, L.ptr(dW), N, L.ptr(wsp), wsb, 0, L.stream())))
    ref = torch.matmul(A.t().float(), B.float())
    err = float((dW - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * K
    print("%-44s %10.1f %10.1f %10.1f %10.1f   max rel err %.1e" % ("%s %dx%dx%d" % (name, M, N, K), t_lib, fl / t_lib / 1e6, t_uic, fl / t_uic / 1e6, err))
