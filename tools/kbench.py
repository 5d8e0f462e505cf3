"""Micro-benchmarks of single libuic_hip.so kernels at the BASELINE config-2 shapes (HIP events on the
launch stream), next to torch.matmul (vendor BLAS) and a device copy as same-hardware reference points."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unpaired_image_captioning_amd import _lib as L

lib = L.load()


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters      # us


def bench_linear(M, N, K, dt=1, flags=4, label=""):
    td = L.TORCH_DTYPE[dt]
    A = torch.randn(M, K, device="cuda").to(td)
    B = (torch.randn(N, K, device="cuda") / K ** 0.5).to(td)
    Cm = torch.empty(M, N, device="cuda", dtype=torch.float32 if (flags & 4 or dt == 0) else td)
    us = timeit(lambda: L.check(lib.uic_linear(dt, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(Cm), N, None, flags, L.stream())))
    us_ref = timeit(lambda: torch.matmul(A, B.t()))
    fl = 2.0 * M * N * K
    print("linear %-22s M=%6d N=%5d K=%6d  uic %8.1f us %7.1f TF/s | torch.matmul %8.1f us %7.1f TF/s" % (
        label, M, N, K, us, fl / us / 1e6, us_ref, fl / us_ref / 1e6), flush=True)


def bench_wgrad_tn(M, N, K, label=""):
    """dW = dY^T X: transposing-read kernel vs (two transposes + NT GEMM) vs torch."""
    dY = torch.randn(K, M, device="cuda").bfloat16()
    X = torch.randn(K, N, device="cuda").bfloat16()
    dW = torch.empty(M, N, device="cuda")
    ws = torch.empty(8 * M * N * 4, dtype=torch.uint8, device="cuda")
    us = timeit(lambda: L.check(lib.uic_linear_wgrad(1, M, N, K, L.ptr(dY), M, L.ptr(X), N, L.ptr(dW), N, L.ptr(ws), ws.numel(), 0, L.stream())))
    dYt = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    Xt = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)

    def old():
        L.check(lib.uic_transpose(1, L.ptr(dY), K, M, M, L.ptr(dYt), K, L.stream()))
        L.check(lib.uic_transpose(1, L.ptr(X), K, N, N, L.ptr(Xt), K, L.stream()))
        L.check(lib.uic_linear(1, M, N, K, L.ptr(dYt), K, L.ptr(Xt), K, L.ptr(dW), N, None, 4, L.stream()))
    us_old = timeit(old)
    us_ref = timeit(lambda: torch.matmul(dY.t(), X))
    fl = 2.0 * M * N * K
    print("wgrad %-18s M=%5d N=%5d K=%6d  tn %8.1f us %6.1f TF/s | transposes+NT %8.1f us | torch %8.1f us %6.1f TF/s" % (
        label, M, N, K, us, fl / us / 1e6, us_old, us_ref, fl / us_ref / 1e6), flush=True)


def bench_lstm(M, H, Ks, dt=1):
    td = L.TORCH_DTYPE[dt]
    xs = [torch.randn(M, k, device="cuda").to(td) for k in Ks]
    wih = (torch.randn(4 * H, sum(Ks), device="cuda") * 0.03).to(td)
    h = torch.randn(M, H, device="cuda").to(td)
    whh = (torch.randn(4 * H, H, device="cuda") * 0.03).to(td)
    b1, b2 = torch.zeros(4 * H, device="cuda"), torch.zeros(4 * H, device="cuda")
    c = torch.randn(M, H, device="cuda")
    c_out, h_out = torch.empty(M, H, device="cuda"), torch.empty(M, H, device="cuda", dtype=td)
    gates = torch.empty(M, 4 * H, device="cuda", dtype=td)
    es = wih.element_size()
    n = len(Ks)
    xp = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    kp = (C.c_int32 * n)(*Ks)
    offs = [sum(Ks[:i]) for i in range(n)]
    wp = (C.c_void_p * n)(*[wih.data_ptr() + o * es for o in offs])
    lp = (C.c_int32 * n)(*[sum(Ks)] * n)
    us = timeit(lambda: L.check(lib.uic_lstm_cell_fwd(dt, M, H, n, xp, kp, wp, lp, L.ptr(h), L.ptr(whh), L.ptr(b1), L.ptr(b2),
                                                      L.ptr(c), L.ptr(c_out), L.ptr(h_out), L.ptr(gates), L.stream())))
    fl = 2.0 * M * 4 * H * (sum(Ks) + H)
    print("lstm_cell M=%d H=%d K=%d+%d: %8.1f us %7.1f TF/s" % (M, H, sum(Ks), H, us, fl / us / 1e6), flush=True)


def bench_attention(N=640, R=36, A=512, H=512, dt=1, T=17):
    td = L.TORCH_DTYPE[dt]
    att_h = torch.randn(N, A, device="cuda")
    p_att = torch.randn(N, R, A, device="cuda").to(td)
    att = torch.randn(N, R, H, device="cuda").abs().to(td)
    w = torch.randn(A, device="cuda") * 0.05
    b = torch.zeros(1, device="cuda")
    alpha = torch.empty(N, R, device="cuda")
    ctx = torch.empty(N, H, device="cuda", dtype=td)
    es = p_att.element_size()
    nbytes = N * (R * A + R * H + 2 * H + R) * es
    us = timeit(lambda: L.check(lib.uic_attention_fwd(dt, N, R, A, H, L.ptr(att_h), L.ptr(p_att), L.ptr(att), L.ptr(w), L.ptr(b),
                                                      None, L.ptr(alpha), L.ptr(ctx), L.stream())))
    print("attention_fwd      %8.1f us  %7.1f GB/s algorithmic" % (us, nbytes / us / 1e3), flush=True)
    dctx = torch.randn(N, H, device="cuda")
    de = torch.empty(N, R, device="cuda")
    dah = torch.empty(N, A, device="cuda", dtype=td)
    us = timeit(lambda: L.check(lib.uic_attention_bwd_step(dt, N, R, A, H, L.ptr(att_h), L.ptr(p_att), L.ptr(att), L.ptr(w),
                                                           L.ptr(alpha), L.ptr(dctx), L.ptr(de), L.ptr(dah), L.stream())))
    print("attention_bwd_step %8.1f us  %7.1f GB/s algorithmic" % (us, nbytes / us / 1e3), flush=True)
    atth_all = torch.randn(T, N, A, device="cuda")
    al_all = torch.softmax(torch.randn(T, N, R, device="cuda"), 2)
    de_all = torch.randn(T, N, R, device="cuda") * 0.01
    dctx_all = torch.randn(T, N, H, device="cuda")
    d_att = torch.empty(N, R, H, device="cuda")
    d_p = torch.empty(N, R, A, device="cuda", dtype=td)
    part = torch.empty(N, A + 1, device="cuda")
    us = timeit(lambda: L.check(lib.uic_attention_bwd_accum(dt, N, R, A, H, T, L.ptr(atth_all), L.ptr(al_all), L.ptr(de_all),
                                                            L.ptr(dctx_all), L.ptr(p_att), L.ptr(w), L.ptr(d_att), L.ptr(d_p),
                                                            L.ptr(part), L.stream())), iters=10)
    print("attention_bwd_accum %8.1f us" % us, flush=True)
    src = torch.empty(nbytes // 4, device="cuda")
    dst = torch.empty_like(src)
    us = timeit(lambda: dst.copy_(src))
    print("device copy of the same %d MB: %8.1f us  %7.1f GB/s (read) " % (nbytes >> 20, us, nbytes / us / 1e3), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "attn"):
        bench_attention()
    if which in ("all", "lstm"):
        bench_lstm(640, 512, [512])            # att_lstm recurrent part: K = 512 + 512
        bench_lstm(640, 512, [512, 512])       # lang_lstm: K = 1024 + 512
    if which in ("all", "tn"):
        bench_wgrad_tn(2048, 1536, 2560, "lstm chunk")
        bench_wgrad_tn(2048, 1536, 640, "lstm last chunk")
        bench_wgrad_tn(512, 512, 2560, "h2att chunk")
        bench_wgrad_tn(9488, 512, 10880, "logit")
        bench_wgrad_tn(512, 512, 23040, "ctx2att")
        bench_wgrad_tn(512, 2048, 23040, "att_embed")
    if which in ("all", "gemm"):
        bench_linear(640, 512, 512, label="h2att fwd")
        bench_linear(640, 1536, 2048, label="dX2")
        bench_linear(640, 1024, 2048, label="dX1")
        bench_linear(2048, 512, 10880, label="dW lstm block")
        bench_linear(2048, 1024, 10880, label="dW lstm 2 blocks")
        bench_linear(512, 512, 10880, label="dW h2att")
        bench_linear(512, 2048, 23040, label="dW att_embed")
        bench_linear(512, 512, 23040, label="dW ctx2att")
        bench_linear(23040, 512, 2048, flags=1, label="att_embed fwd")
        bench_linear(23040, 512, 512, flags=0, label="ctx2att fwd")
        bench_linear(10880, 2048, 512, label="Gx")
        bench_linear(10880, 9488, 512, label="logits")
        bench_linear(10880, 512, 9488, label="dH")
        bench_linear(9488, 512, 10880, label="dW logit")
