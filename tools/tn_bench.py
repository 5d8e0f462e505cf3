#!/usr/bin/env python3
"""The step's weight-gradient (TN) GEMM shapes on the two transposing-read kernels: the 128 x 128 tile (csrc/gemm_tn.hip) and the
256 x 256 ping-pong tile (csrc/gemm_tn_pp.hip), per split-K, next to torch.matmul (hipBLASLt).  One HIP event pair per launch,
variants interleaved in rounds (one process, one device), trimmed mean; every variant is checked against an f32 matmul of the
same bf16 values first.  Diagnostic only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unpaired_image_captioning_amd import _lib as L

lib = L.load()
F128, F256 = 0x100, 0x200


def SK(n):
    return (n & 0xff) << 16


def time_variants(fns, rounds=12):
    for f in fns.values():
        for _ in range(2):
            f()
    ev = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record()
            ev[k].append((a, b))
    torch.cuda.synchronize()
    out = {}
    for k, l in ev.items():
        d = sorted(a.elapsed_time(b) for a, b in l)
        cut = len(d) // 6
        d = d[cut: len(d) - cut]
        out[k] = sum(d) / len(d) * 1e3
    return out


g = torch.Generator(device="cuda").manual_seed(5)
wsb = 256 << 20
wsp = torch.empty(wsb, dtype=torch.uint8, device="cuda")
shapes = [("lang / att LSTM dW, 4-step chunk", 2048, 1664, 2560), ("h2att dW, 4-step chunk", 512, 640, 2560), ("LSTM dW, 1-step chunk", 2048, 1664, 640),
          ("logit dW", 9488, 512, 10880), ("att_embed dW", 512, 2176, 23040), ("ctx2att dW", 512, 640, 23040),
          ("fc_embed dW", 512, 2048, 640), ("att_lstm fc' columns", 2048, 512, 640), ("NMT generator dW", 50004, 512, 1984), ("NMT generator dW, rows padded to 2048", 50004, 512, 2048), ("NMT LSTM dW, rows padded", 2048, 1024, 2048)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
for name, M, N, K in shapes:
    lda = (M + 7) // 8 * 8
    A = torch.randn(K, lda, device="cuda", generator=g).bfloat16()
    B = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    ref = torch.matmul(A[:, :M].t().float(), B.float())
    dW = torch.empty(M, N, device="cuda", dtype=torch.float32)
    nt = K // 64
    variants = {"auto": 0, "128 auto": F128}
    for sk in (1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 18, 20, 24, 30, 36):
        t256 = ((M + 255) // 256) * ((N + 255) // 256)
        if K % (128 * sk) == 0 and nt // sk >= 2 and t256 * sk <= 512 and sk * M * N * 4 <= wsb:
            variants["256 sk%d" % sk] = F256 | SK(sk)
    fns = {}
    for k, how in variants.items():
        def f(how=how):
            L.check(lib.uic_linear_wgrad(1, M, N, K, L.ptr(A), lda, L.ptr(B), N, L.ptr(dW), N, L.ptr(wsp), wsb, how, L.stream()))
        dW.zero_()
        f()
        err = float((dW - ref).abs().max() / ref.abs().max())
        assert err < 2e-3, (name, k, err)
        # accumulate form: twice the product
        if k in ("auto", "256 sk1"):
            def fa(how=how):
                L.check(lib.uic_linear_wgrad(1, M, N, K, L.ptr(A), lda, L.ptr(B), N, L.ptr(dW), N, L.ptr(wsp), wsb, how | 1, L.stream()))
            fa()
            err = float((dW - 2 * ref).abs().max() / ref.abs().max())
            assert err < 4e-3, (name, k, "accumulate", err)
        fns[k] = f
    fns["torch"] = lambda: torch.matmul(A[:, :M].t(), B)
    t = time_variants(fns)
    fl = 2.0 * M * N * K
    print("%-34s %6d x %5d x %6d  " % (name, M, N, K) + "  ".join("%s %.1f us (%.0f TF/s)" % (k, v, fl / v / 1e6) for k, v in t.items()), flush=True)
