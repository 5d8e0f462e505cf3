"""GPU parity of the single operators of libuic_hip.so against the CPU oracle / plain torch fp32
(called through the C ABI).  Tolerances: f32 path 1e-4 relative to the tensor's scale (MFMA f32 is
an exact-f32 fma chain, only the summation order differs); bf16 path 2e-2 (operands rounded to bf16)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import topdown as O  # noqa: E402


def _lib():
    from unpaired_image_captioning_amd import _lib
    return _lib


TOL = {0: 1e-4, 1: 2e-2}
TD = {0: torch.float32, 1: torch.bfloat16}


_KEEP = []


def dev(t, dt=None):
    """Device copy that stays alive until the test module is torn down (raw pointers are passed to C)."""
    if t is None:
        return None
    t = t.cuda()
    t = t.to(TD[dt]) if dt is not None else t
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t


def rel_err(got, ref):
    got = got.detach().float().cpu().double()
    ref = ref.detach().float().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


def rounded(t, dt):
    return t.to(TD[dt]).float()


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(70, 51, 72), (640, 512, 512), (129, 200, 40), (1, 8, 8), (2304, 96, 2048)])
@pytest.mark.parametrize("flags", [0, 1, 4, 6])
def test_linear(dt, shape, flags):
    L = _lib()
    lib = L.load()
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K + flags)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    out_f32 = bool(flags & 4) or dt == 0
    Ad, Bd = dev(A, dt), dev(B, dt)
    Cd = dev(C0) if out_f32 else dev(C0, dt)
    L.check(lib.uic_linear(dt, M, N, K, L.ptr(Ad), K, L.ptr(Bd), K, L.ptr(Cd), N, L.ptr(dev(bias)), flags, L.stream()))
    torch.cuda.synchronize()
    ref = rounded(A, dt) @ rounded(B, dt).t() + bias
    if flags & 1:
        ref = torch.relu(ref)
    if flags & 2:
        ref = ref + (C0 if out_f32 else rounded(C0, dt))
    assert rel_err(Cd, ref) < TOL[dt]


FORCE_128, FORCE_256, FORCE_192, FORCE_PP128 = 0x100, 0x200, 0x400, 0x800


@pytest.mark.parametrize("shape", [(256, 256, 128), (300, 260, 256), (2304, 512, 2048), (10880, 512, 512), (1, 4, 384), (777, 1028, 640)])
@pytest.mark.parametrize("flags", [0, 1, 4, 6, 2])
@pytest.mark.parametrize("FORCE_PP", [FORCE_256, FORCE_192, FORCE_PP128])
def test_linear_256_tile_kernel_equals_the_128_tile_kernel_and_torch(shape, flags, FORCE_PP):
    """The ping-pong GEMM (csrc/gemm_pp.hip; 256 / 192 / 128-row x 256-column tiles), forced per call, against plain torch on the
    bf16-rounded operands AND against the 128 x 128 LDS-DMA kernel on the same operands: one tile exactly, ragged row / column
    tiles (clamped source rows, guarded stores), 2 / 4 / 6 / 8 / 10 / 32 K tiles (prologue + last-iteration forms of its
    pipeline), f32 and bf16 outputs, bias, ReLU, accumulate.  Repeated launches must agree bit for bit (a staged buffer read one
    phase early would not)."""
    L = _lib()
    lib = L.load()
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K + flags)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    out_f32 = bool(flags & 4)
    Ad, Bd, bd = dev(A, 1), dev(B, 1), dev(bias)
    outs = []
    for force in (FORCE_PP, FORCE_PP, FORCE_128):
        Cd = C0.clone().cuda() if out_f32 else C0.clone().cuda().bfloat16()
        L.check(lib.uic_linear(1, M, N, K, L.ptr(Ad), K, L.ptr(Bd), K, L.ptr(Cd), N, L.ptr(bd), flags | force, L.stream()))
        torch.cuda.synchronize()
        outs.append(Cd)
    ref = rounded(A, 1).double() @ rounded(B, 1).double().t() + bias.double()
    if flags & 1:
        ref = torch.relu(ref)
    if flags & 2:
        ref = ref + (C0 if out_f32 else rounded(C0, 1)).double()
    assert torch.equal(outs[0], outs[1])
    tol = 2e-5 if out_f32 else 1e-2
    assert rel_err(outs[0], ref.float()) < tol
    assert rel_err(outs[0], outs[2].float()) < tol


@pytest.mark.parametrize("shape", [(192, 256, 128), (300, 260, 256), (4608, 512, 2048), (2880, 512, 384), (1000, 1028, 640), (130, 8, 1024)])
@pytest.mark.parametrize("flags", [0, 1, 5])
@pytest.mark.parametrize("copy", [0, 1])
def test_linear_with_f32_input_equals_cast_then_linear(shape, flags, copy):
    """uic_linear_f32a (the ping-pong GEMM with an f32 A operand rounded on its way to LDS -- att_embed on the loader's f32 region
    features): bit for bit what the cast pass followed by the bf16 ping-pong GEMM gives, and the stored bf16 image of A equals
    torch's round-to-nearest-even cast.  Ragged row / column tiles (clamped source rows), 2 / 4 / 6 / 10 / 16 / 32 K tiles (the
    register-staged A units' prologue, steady-state and last-iteration waits), with and without the image (the copy stores are
    counted in the same vmcnt sequence), repeated launches equal."""
    L = _lib()
    lib = L.load()
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 13 + N * 5 + K + flags)
    A = torch.randn(M, K, generator=g) * 3
    B = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    out_f32 = bool(flags & 4)
    Af, Ab, Bd, bd = A.cuda(), A.cuda().bfloat16(), dev(B, 1), dev(bias)
    outs = []
    for rep in range(2):
        Cd = torch.full((M, N), 7.0, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
        img = torch.full((M, K), 7.0, device="cuda", dtype=torch.bfloat16)
        L.check(lib.uic_linear_f32a(M, N, K, L.ptr(Af), K, L.ptr(Bd), K, L.ptr(Cd), N, L.ptr(bd), flags, L.ptr(img) if copy else None, K,
                                    L.stream()))
        torch.cuda.synchronize()
        outs.append((Cd, img))
    Cr = torch.full((M, N), 7.0, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
    L.check(lib.uic_linear(1, M, N, K, L.ptr(Ab), K, L.ptr(Bd), K, L.ptr(Cr), N, L.ptr(bd), flags | FORCE_192, L.stream()))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][0], Cr)
    if copy:
        assert torch.equal(outs[0][1], Ab) and torch.equal(outs[1][1], Ab)
    else:
        assert (outs[0][1] == 7.0).all()


def test_linear_with_f32_input_beside_busy_neighbours():
    """The same equality, 300 launches with an HBM-bound copy and an MFMA-bound GEMM running on another stream (timing inside
    the kernel changes: its counted waits and the image stores must not care).  The first form of the image store -- inline asm
    the compiler could not see as a VMEM store -- failed exactly here, 1 launch in 100 (tools/gemm_f32a_soak.py)."""
    L = _lib()
    lib = L.load()
    M, N, K = 4608, 512, 2048
    g = torch.Generator().manual_seed(9)
    A = (torch.randn(M, K, generator=g) * 3).cuda()
    Ab = A.bfloat16()
    Bd = dev(torch.randn(N, K, generator=g) / K ** 0.5, 1)
    bd = dev(torch.randn(N, generator=g))
    ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    L.check(lib.uic_linear(1, M, N, K, L.ptr(Ab), K, L.ptr(Bd), K, L.ptr(ref), N, L.ptr(bd), 1 | FORCE_192, L.stream()))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    hog_a = torch.randn(32 << 20, device="cuda")
    hog_b = torch.empty_like(hog_a)
    X = torch.randn(4096, 4096, device="cuda").bfloat16()
    bad = 0
    for it in range(300):
        with torch.cuda.stream(side):
            if it % 2 == 0:
                hog_b.copy_(hog_a)
            if it % 3 != 0:
                torch.matmul(X, X)
        C = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        img = torch.full((M, K), 7.0, device="cuda", dtype=torch.bfloat16)
        L.check(lib.uic_linear_f32a(M, N, K, L.ptr(A), K, L.ptr(Bd), K, L.ptr(C), N, L.ptr(bd), 1, L.ptr(img), K, L.stream()))
        torch.cuda.synchronize()
        bad += int(not (torch.equal(C, ref) and torch.equal(img, Ab)))
    assert bad == 0, bad


def test_linear_with_f32_input_rejects_what_it_cannot_run():
    L = _lib()
    lib = L.load()
    a = torch.zeros(256, 256, device="cuda")
    w = torch.zeros(256, 256, device="cuda", dtype=torch.bfloat16)
    o = torch.zeros(256, 256, device="cuda")
    assert lib.uic_linear_f32a(256, 256, 192, L.ptr(a), 192, L.ptr(w), 192, L.ptr(o), 256, None, 4, None, 192, L.stream()) < 0   # odd K tile count
    assert lib.uic_linear_f32a(256, 256, 128, L.ptr(a), 130, L.ptr(w), 128, L.ptr(o), 256, None, 4, None, 128, L.stream()) < 0   # lda % 4
    assert lib.uic_linear_f32a(256, 256, 128, L.ptr(a), 128, L.ptr(w), 128, L.ptr(o), 256, None, 4, L.ptr(w), 100, L.stream()) < 0   # image narrower than K


def test_linear_256_tile_kernel_rejects_what_it_cannot_run():
    L = _lib()
    lib = L.load()
    x = torch.zeros(256, 256, device="cuda", dtype=torch.bfloat16)
    o = torch.zeros(256, 256, device="cuda")
    assert lib.uic_linear(1, 256, 256, 64, L.ptr(x), 64, L.ptr(x), 64, L.ptr(o), 256, None, 4 | FORCE_256, L.stream()) < 0      # one K tile
    assert lib.uic_linear(1, 256, 256, 192, L.ptr(x), 192, L.ptr(x), 192, L.ptr(o), 256, None, 4 | FORCE_256, L.stream()) < 0   # odd tile count
    assert lib.uic_linear(0, 256, 256, 128, L.ptr(o), 128, L.ptr(o), 128, L.ptr(o), 256, None, 4 | FORCE_256, L.stream()) < 0   # f32 operands


@pytest.mark.parametrize("dt", [0, 1])
def test_linear_strided_operands(dt):
    """Column blocks of wider matrices: lda/ldb/ldc larger than the logical widths (weight_ih column blocks)."""
    L = _lib()
    lib = L.load()
    M, N, K, ld = 37, 64, 24, 88
    g = torch.Generator().manual_seed(5)
    Abig = torch.randn(M, ld, generator=g)
    Bbig = torch.randn(N, ld, generator=g)
    Cbig = torch.zeros(M, 100)
    Ad, Bd, Cd = dev(Abig, dt), dev(Bbig, dt), dev(Cbig)
    es = Ad.element_size()
    L.check(lib.uic_linear(dt, M, N, K, Ad.data_ptr() + 16 * es, ld, Bd.data_ptr() + 40 * es, ld,
                           Cd.data_ptr() + 4 * 4, 100, None, 4, L.stream()))
    torch.cuda.synchronize()
    ref = rounded(Abig, dt)[:, 16:16 + K] @ rounded(Bbig, dt)[:, 40:40 + K].t()
    assert rel_err(Cd[:, 4:4 + N], ref) < TOL[dt]
    assert Cd[:, :4].abs().max().item() == 0 and Cd[:, 4 + N:].abs().max().item() == 0


def test_linear_rejects_bad_arguments():
    L = _lib()
    lib = L.load()
    x = torch.zeros(64, 64, device="cuda")
    rc = lib.uic_linear(0, 8, 8, 6, L.ptr(x), 6, L.ptr(x), 6, L.ptr(x), 8, None, 0, L.stream())   # K % 4 != 0
    assert rc < 0 and b"multiples" in lib.uic_last_error_string()
    rc = lib.uic_linear(7, 8, 8, 8, L.ptr(x), 8, L.ptr(x), 8, L.ptr(x), 8, None, 0, L.stream())
    assert rc < 0


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("MH", [(70, 40), (640, 512), (6, 32), (33, 96)])
def test_lstm_cell_fwd_bwd(dt, MH):
    L = _lib()
    lib = L.load()
    M, H = MH
    K1, K2 = 24, H
    g = torch.Generator().manual_seed(M + H)
    x1, x2, h = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g), torch.randn(M, H, generator=g)
    c = torch.randn(M, H, generator=g)
    w_ih = torch.randn(4 * H, K1 + K2, generator=g) / (K1 + K2) ** 0.5
    w_hh = torch.randn(4 * H, H, generator=g) / H ** 0.5
    b_ih, b_hh = torch.randn(4 * H, generator=g), torch.randn(4 * H, generator=g)
    x1d, x2d, hd, wihd, whhd = dev(x1, dt), dev(x2, dt), dev(h, dt), dev(w_ih, dt), dev(w_hh, dt)
    es = x1d.element_size()
    c_out = torch.empty(M, H, device="cuda")
    h_out = torch.empty(M, H, device="cuda", dtype=TD[dt])
    gates = torch.empty(M, 4 * H, device="cuda", dtype=TD[dt])
    xs = (C.c_void_p * 2)(x1d.data_ptr(), x2d.data_ptr())
    ks = (C.c_int32 * 2)(K1, K2)
    ws = (C.c_void_p * 2)(wihd.data_ptr(), wihd.data_ptr() + K1 * es)
    lds = (C.c_int32 * 2)(K1 + K2, K1 + K2)
    L.check(lib.uic_lstm_cell_fwd(dt, M, H, 2, xs, ks, ws, lds, L.ptr(hd), L.ptr(whhd), L.ptr(dev(b_ih)), L.ptr(dev(b_hh)),
                                  L.ptr(dev(c)), L.ptr(c_out), L.ptr(h_out), L.ptr(gates), L.stream()))
    torch.cuda.synchronize()
    xr = torch.cat([rounded(x1, dt), rounded(x2, dt)], 1)
    hr, cr = O.lstm_cell(xr, rounded(h, dt), c, rounded(w_ih, dt), rounded(w_hh, dt), b_ih, b_hh)
    assert rel_err(h_out, hr) < TOL[dt]
    assert rel_err(c_out, cr) < TOL[dt]
    # pointwise backward vs autograd on the same activated gates
    dh, dc_in = torch.randn(M, H, generator=g), torch.randn(M, H, generator=g)
    gact = gates.float().cpu()
    pre = torch.zeros(M, 4 * H, requires_grad=True)
    i, f, gg, o = gact.chunk(4, 1)
    # express activated gates as functions of a dummy pre-activation so autograd yields d(pre)
    pre_vals = torch.cat([torch.logit(i.clamp(1e-6, 1 - 1e-6)), torch.logit(f.clamp(1e-6, 1 - 1e-6)),
                          torch.atanh(gg.clamp(-1 + 1e-6, 1 - 1e-6)), torch.logit(o.clamp(1e-6, 1 - 1e-6))], 1)
    pre = pre_vals.clone().requires_grad_(True)
    c_prev = c.clone().requires_grad_(True)
    pi, pf, pg, po = pre.chunk(4, 1)
    c2 = torch.sigmoid(pf) * c_prev + torch.sigmoid(pi) * torch.tanh(pg)
    h2 = torch.sigmoid(po) * torch.tanh(c2)
    (h2 * dh).sum().backward(retain_graph=True, inputs=[pre, c_prev])
    gpre, gc = pre.grad.clone(), c_prev.grad.clone()
    pre.grad = None
    c_prev.grad = None
    (c2 * dc_in).sum().backward(inputs=[pre, c_prev])
    gpre += pre.grad
    gc += c_prev.grad
    dcd = dev(dc_in).clone()
    dgates = torch.empty(M, 4 * H, device="cuda", dtype=TD[dt])
    c2d = c2.detach().cuda()
    L.check(lib.uic_lstm_cell_bwd(dt, M, H, L.ptr(dev(dh)), L.ptr(dcd), L.ptr(gates), L.ptr(dev(c)), L.ptr(c2d), L.ptr(dgates), L.stream()))
    torch.cuda.synchronize()
    assert rel_err(dgates, gpre) < max(TOL[dt], 2e-3)
    assert rel_err(dcd, gc) < max(TOL[dt], 2e-3)


def _attn_inputs(N, R, A, H, seed, masked):
    g = torch.Generator().manual_seed(seed)
    att_h = torch.randn(N, A, generator=g)
    p_att = torch.randn(N, R, A, generator=g)
    att = torch.randn(N, R, H, generator=g).abs()
    w = torch.randn(A, generator=g) / A ** 0.5
    b = torch.randn(1, generator=g)
    mask = None
    if masked:
        cnt = torch.randint(1, R + 1, (N,), generator=g)
        cnt[0] = R
        mask = (torch.arange(R)[None, :] < cnt[:, None]).float()
    return att_h, p_att, att, w, b, mask


def _attn_ref(att_h, p_att, att, w, b, mask):
    dot = torch.tanh(p_att + att_h.unsqueeze(1)) @ w + b
    weight = torch.softmax(dot, 1)
    if mask is not None:
        weight = weight * mask
        weight = weight / weight.sum(1, keepdim=True)
    return torch.bmm(weight.unsqueeze(1), att).squeeze(1), weight


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("dims", [(6, 5, 32, 32), (7, 7, 48, 40), (40, 36, 512, 512), (3, 196, 64, 128)])
@pytest.mark.parametrize("masked", [False, True])
def test_attention_fwd_bwd(dt, dims, masked):
    L = _lib()
    lib = L.load()
    N, R, A, H = dims
    att_h, p_att, att, w, b, mask = _attn_inputs(N, R, A, H, N * R + A, masked)
    pr, ar = rounded(p_att, dt), rounded(att, dt)
    att_h_r = att_h.clone().requires_grad_(True)
    pr_g, ar_g, w_g = pr.clone().requires_grad_(True), ar.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ctx_ref, alpha_ref = _attn_ref(att_h_r, pr_g, ar_g, w_g, b, mask)
    pd, ad = dev(p_att, dt), dev(att, dt)
    alpha = torch.empty(N, R, device="cuda")
    ctx = torch.empty(N, H, device="cuda", dtype=TD[dt])
    L.check(lib.uic_attention_fwd(dt, N, R, A, H, L.ptr(dev(att_h)), L.ptr(pd), L.ptr(ad), L.ptr(dev(w)), L.ptr(dev(b)),
                                  L.ptr(dev(mask)) if masked else None, L.ptr(alpha), L.ptr(ctx), L.stream()))
    torch.cuda.synchronize()
    tol = 1e-4 if dt == 0 else 2e-2
    assert rel_err(alpha, alpha_ref) < tol
    assert rel_err(ctx, ctx_ref) < tol
    # backward of one step
    g = torch.Generator().manual_seed(3)
    dctx = torch.randn(N, H, generator=g)
    (ctx_ref * dctx).sum().backward()
    de = torch.empty(N, R, device="cuda")
    d_att_h = torch.empty(N, A, device="cuda", dtype=TD[dt])
    alpha_in = alpha_ref.detach().cuda().contiguous()
    L.check(lib.uic_attention_bwd_step(dt, N, R, A, H, L.ptr(dev(att_h)), L.ptr(pd), L.ptr(ad), L.ptr(dev(w)), L.ptr(alpha_in),
                                       L.ptr(dev(dctx)), L.ptr(de), L.ptr(d_att_h), L.stream()))
    torch.cuda.synchronize()
    assert rel_err(d_att_h, att_h_r.grad) < max(tol, 1e-3)
    # deferred accumulation with T = 1 reproduces d att, d p_att, d w_alpha of that single step
    d_att = torch.empty(N, R, H, device="cuda")
    d_p_att = torch.empty(N, R, A, device="cuda", dtype=TD[dt])
    part = torch.empty(N, A + 1, device="cuda")
    L.check(lib.uic_attention_bwd_accum(dt, N, R, A, H, 1, L.ptr(dev(att_h)), L.ptr(alpha_in), L.ptr(de), L.ptr(dev(dctx)),
                                        L.ptr(pd), L.ptr(dev(w)), L.ptr(d_att), L.ptr(d_p_att), L.ptr(part), L.stream()))
    torch.cuda.synchronize()
    assert rel_err(d_att, ar_g.grad) < max(tol, 1e-3)
    assert rel_err(d_p_att, pr_g.grad) < max(tol, 1e-3)
    assert rel_err(part[:, :A].sum(0), w_g.grad) < max(tol, 1e-3)
    assert part[:, A].abs().max().item() < 1e-3       # d b_alpha = sum(de) = 0 (softmax shift invariance)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("large", [0, 1, 2])
def test_attention_bwd_accum_multi_step(dt, large):
    """large = 1 / 2: pre-activations beyond the range in which the bf16 kernel may factor e^{2(p + h)} into
    e^{2p} e^{2h} (|2x| > 40), incl. operands of opposite sign whose SUM is small -- the kernel must fall back to the
    direct form for those rows (1: a few p_att entries, 2: a whole caption row's att_h)."""
    L = _lib()
    lib = L.load()
    N, R, A, H, T = 5, 9, 64, 48, 4
    g = torch.Generator().manual_seed(77)
    p_att = torch.randn(N, R, A, generator=g)
    w = torch.randn(A, generator=g) / A ** 0.5
    att_h = torch.randn(T, N, A, generator=g)
    if large == 1:
        p_att[1, 2, 5] = 32.0; att_h[:, 1, 5] = -31.5          # sum 0.5: tanh far from saturation
        p_att[3, 0, 9] = -64.0; p_att[4, 8, 63] = 100.0
    elif large == 2:
        att_h[2, 2] = 48.0 * torch.sign(torch.randn(A, generator=g))
        p_att[2] -= att_h[2, 2] * 0.99
    alpha = torch.softmax(torch.randn(T, N, R, generator=g), 2)
    de = torch.randn(T, N, R, generator=g) * 0.1
    dctx = torch.randn(T, N, H, generator=g)
    pr = rounded(p_att, dt)
    d_att_ref = torch.einsum("tnr,tnh->nrh", alpha, dctx)
    th = torch.tanh(pr.unsqueeze(0) + att_h.unsqueeze(2))                       # [T,N,R,A]
    d_p_ref = (de.unsqueeze(3) * (1 - th * th)).sum(0) * w
    d_w_ref = (de.unsqueeze(3) * th).sum((0, 1, 2))
    d_att = torch.empty(N, R, H, device="cuda")
    d_p = torch.empty(N, R, A, device="cuda", dtype=TD[dt])
    part = torch.empty(N, A + 1, device="cuda")
    L.check(lib.uic_attention_bwd_accum(dt, N, R, A, H, T, L.ptr(dev(att_h)), L.ptr(dev(alpha)), L.ptr(dev(de)), L.ptr(dev(dctx)),
                                        L.ptr(dev(p_att, dt)), L.ptr(dev(w)), L.ptr(d_att), L.ptr(d_p), L.ptr(part), L.stream()))
    torch.cuda.synchronize()
    tol = 1e-4 if dt == 0 else 2e-2
    assert rel_err(d_att, d_att_ref) < tol
    assert rel_err(d_p, d_p_ref) < tol
    assert rel_err(part[:, :A].sum(0), d_w_ref) < tol
    assert rel_err(part[:, A].sum().view(1), de.sum().view(1)) < 1e-3


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(3, 40, 128, 64, 5), (2, 36, 512, 512, 17), (4, 37, 72, 40, 6), (1, 1, 8, 8, 1), (2, 75, 256, 128, 7)])
def test_attention_bwd_accum_shapes(dt, shape):
    """The accumulation passes at shapes that exercise every loop bound of the round-6 kernels: more than 36 regions (a second
    block of nine accumulators per wave), odd and even step counts (the shared-reciprocal pairs + the odd step), column counts that
    are not a multiple of the 64-lane pass, the reference's own 36 x 512 x 512 x 17."""
    L = _lib()
    lib = L.load()
    N, R, A, H, T = shape
    g = torch.Generator().manual_seed(N * 1000 + R)
    p_att = torch.randn(N, R, A, generator=g) * 1.5
    w = torch.randn(A, generator=g) / A ** 0.5
    att_h = torch.randn(T, N, A, generator=g) * 1.5
    alpha = torch.softmax(torch.randn(T, N, R, generator=g), 2)
    de = torch.randn(T, N, R, generator=g) * 0.1
    dctx = torch.randn(T, N, H, generator=g)
    pr = rounded(p_att, dt)
    d_att_ref = torch.einsum("tnr,tnh->nrh", alpha, dctx)
    th = torch.tanh(pr.unsqueeze(0).double() + att_h.unsqueeze(2).double())    # [T,N,R,A]
    d_p_ref = ((de.unsqueeze(3).double() * (1 - th * th)).sum(0) * w.double()).float()
    d_w_ref = (de.unsqueeze(3).double() * th).sum((0, 1, 2)).float()
    d_att = torch.full((N, R, H), 7.0, device="cuda")
    d_p = torch.full((N, R, A), 7.0, device="cuda", dtype=TD[dt])
    part = torch.full((N, A + 1), 7.0, device="cuda")
    L.check(lib.uic_attention_bwd_accum(dt, N, R, A, H, T, L.ptr(dev(att_h)), L.ptr(dev(alpha)), L.ptr(dev(de)), L.ptr(dev(dctx)),
                                        L.ptr(dev(p_att, dt)), L.ptr(dev(w)), L.ptr(d_att), L.ptr(d_p), L.ptr(part), L.stream()))
    torch.cuda.synchronize()
    tol = 1e-4 if dt == 0 else 1e-2
    assert rel_err(d_att, d_att_ref) < 1e-5
    assert rel_err(d_p, d_p_ref) < tol
    assert rel_err(part[:, :A].sum(0), d_w_ref) < (1e-4 if dt == 0 else 2e-3)
    assert rel_err(part[:, A].sum().view(1), de.sum().view(1)) < 1e-3
    # a second launch reproduces the first bit for bit
    d_p2 = torch.empty_like(d_p)
    part2 = torch.empty_like(part)
    d_att2 = torch.empty_like(d_att)
    L.check(lib.uic_attention_bwd_accum(dt, N, R, A, H, T, L.ptr(dev(att_h)), L.ptr(dev(alpha)), L.ptr(dev(de)), L.ptr(dev(dctx)),
                                        L.ptr(dev(p_att, dt)), L.ptr(dev(w)), L.ptr(d_att2), L.ptr(d_p2), L.ptr(part2), L.stream()))
    torch.cuda.synchronize()
    assert torch.equal(d_p, d_p2) and torch.equal(part, part2) and torch.equal(d_att, d_att2)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(42, 78, 80), (130, 64, 64), (7, 5, 8)])
def test_transpose_and_casts(dt, shape):
    L = _lib()
    lib = L.load()
    rows, cols, lds = shape
    ldd = (rows + 7) // 8 * 8
    g = torch.Generator().manual_seed(rows)
    src = torch.randn(rows, lds, generator=g)
    sd = torch.empty(rows, lds, device="cuda", dtype=TD[dt])
    L.check(lib.uic_cast_from_f32(dt, L.ptr(dev(src)), L.ptr(sd), rows * lds, L.stream()))
    dst = torch.full((cols, ldd), 7.0, device="cuda", dtype=TD[dt])
    L.check(lib.uic_transpose(dt, L.ptr(sd), rows, cols, lds, L.ptr(dst), ldd, L.stream()))
    back = torch.empty(cols, ldd, device="cuda")
    L.check(lib.uic_cast_to_f32(dt, L.ptr(dst), L.ptr(back), cols * ldd, L.stream()))
    torch.cuda.synchronize()
    ref = rounded(src, dt)[:, :cols].t()
    assert torch.equal(back[:, :rows].cpu(), ref)            # bit-exact data movement
    assert back[:, rows:].abs().max().item() == 0 if ldd > rows else True


def test_adam_step_matches_oracle():
    L = _lib()
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    n = 100003
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) for _ in range(3)]
    P = {"w": p0.clone()}
    m, v = {"w": torch.zeros(n)}, {"w": torch.zeros(n)}
    pd, md, vd = dev(p0).clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step, gr in enumerate(grads, 1):
        O.adam_step(P, {"w": gr}, m, v, step, 5e-4)
        L.check(lib.uic_adam_step(L.ptr(pd), L.ptr(dev(gr)), L.ptr(md), L.ptr(vd), n, 5e-4, 0.9, 0.999, 1e-8, step, 1.0, L.stream()))
    torch.cuda.synchronize()
    assert (pd.cpu() - P["w"]).abs().max().item() < 1e-6


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("M,N,K,S", [(640, 1536, 2048, 4), (640, 1024, 2048, 4), (640, 512, 2048, 8), (600, 1000, 1024, 2), (128, 128, 512, 1)])
def test_linear_partials_slices_add_up_to_the_product_and_repeat_bit_for_bit(dtype, M, N, K, S):
    """uic_linear_partials (the BPTT loop's split-K d x GEMMs, csrc/gemm.hip: LDS-DMA staging ordered by hand-placed vmcnt waits
    and raw barriers): the slices add up to the product, and 30 launches beside an HBM-bound neighbour on another stream all give
    the first launch's bits (a wait that is one round short shows up as a result that depends on timing).  Round 6 also tried a
    ring of four staging buffers for these launches: same bits, same 13.4 us -- their time is MFMA issue (3.9 us per wave),
    the slab stores and launch ramps, not exposed load latency -- and it was removed again."""
    L = _lib()
    lib = L.load()
    dt = L.dtype_id(dtype)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(td)
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.5).to(td)
    ref = A.double() @ B.double().t()

    def run():
        slab = torch.full((S, M, N), float("nan"), device="cuda")
        L.check(lib.uic_linear_partials(dt, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(slab), S, L.stream()))
        return slab
    first = run()
    torch.cuda.synchronize()
    err = (first.double().sum(0) - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device="cuda")
    for _ in range(30):
        with torch.cuda.stream(side):
            big.add_(1.0)
        assert torch.equal(run(), first)
    torch.cuda.synchronize()


@pytest.mark.parametrize("w16", [False, True], ids=["f32", "bf16-copy"])
@pytest.mark.parametrize("clip,guard", [(False, None), (True, None), (False, 0), (False, 7)])
def test_adam_step_ranges_is_adam_step_on_the_ranges_and_nowhere_else(w16, clip, guard):
    """uic_adam_step_ranges (the sharded exchange's optimizer launch): on every listed range bit for bit what uic_adam_step[_clip]
    leaves there, nothing written outside the ranges or past the arenas (canaries around all four arrays and the operand copy),
    the bf16 copy of exactly the updated elements, and no update at all behind a non-zero guard word."""
    L = _lib()
    lib = L.load()
    g = torch.Generator().manual_seed(21)
    n, pad = 70016, 256
    ranges = [(0, 64), (128, 128), (4096, 4096 + 9984), (20032, 20032 + 1), (50048, 70016)]      # (one empty, one single element, the tail)
    big = torch.full((5, n + 2 * pad), 7.25)
    p0, gr, m0, v0 = torch.randn(n, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.01
    for row, t in enumerate((p0, gr, m0, v0)):
        big[row, pad:pad + n] = t
    dev_big = big.cuda()
    P, G, M, V = (dev_big[r, pad:pad + n] for r in range(4))
    w_out = torch.full((n + 2 * pad,), 3.0, device="cuda").bfloat16()
    W = w_out[pad:pad + n]
    sq = torch.tensor([float((gr * gr).sum())], device="cuda") if clip else None
    gw = torch.tensor([0, guard if guard is not None else 0], dtype=torch.int32, device="cuda") if guard is not None else None
    lo = (C.c_uint64 * len(ranges))(*[a for a, _ in ranges])
    hi = (C.c_uint64 * len(ranges))(*[b for _, b in ranges])
    L.check(lib.uic_adam_step_ranges(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), len(ranges), lo, hi, 5e-3, 0.9, 0.999, 1e-8, 3, 0.5,
                                     0.25 if clip else 0.0, L.ptr(sq), gw[1:2].data_ptr() if gw is not None else None,
                                     W.data_ptr() if w16 else None, 1 if w16 else 0, L.stream()))
    # the reference: the whole-arena kernel on copies
    rp, rm, rv = dev(p0).clone(), dev(m0).clone(), dev(v0).clone()
    if clip:
        L.check(lib.uic_adam_step_clip(L.ptr(rp), L.ptr(dev(gr)), L.ptr(rm), L.ptr(rv), n, 5e-3, 0.9, 0.999, 1e-8, 3, 0.5, 0.25, L.ptr(sq), L.stream()))
    else:
        L.check(lib.uic_adam_step(L.ptr(rp), L.ptr(dev(gr)), L.ptr(rm), L.ptr(rv), n, 5e-3, 0.9, 0.999, 1e-8, 3, 0.5, L.stream()))
    torch.cuda.synchronize()
    inside = torch.zeros(n, dtype=torch.bool)
    for a, b in ranges:
        inside[a:b] = True
    skipped = guard is not None and guard != 0
    out = dev_big.cpu()
    for row, (new, old) in enumerate(((rp.cpu(), p0), (None, gr), (rm.cpu(), m0), (rv.cpu(), v0))):
        got = out[row, pad:pad + n]
        want = old.clone()
        if new is not None and not skipped:
            want[inside] = new[inside]
        assert torch.equal(got, want), (row, (got - want).abs().max().item(), (got != want).nonzero()[:4].tolist())
        assert (out[row, :pad] == 7.25).all() and (out[row, pad + n:] == 7.25).all(), row        # canaries
    assert (out[4] == 7.25).all()
    wo = w_out.float().cpu()
    assert (wo[:pad] == 3.0).all() and (wo[pad + n:] == 3.0).all()
    wmid = wo[pad:pad + n]
    if w16 and not skipped:
        assert torch.equal(wmid[inside], rp.cpu()[inside].bfloat16().float()) and (wmid[~inside] == 3.0).all()
    else:
        assert (wmid == 3.0).all()


@pytest.mark.parametrize("M,N,K,sk", [(2048, 1664, 2560, 1), (2048, 1664, 2560, 2), (512, 640, 2560, 5), (9488, 512, 10880, 5), (520, 384, 1280, 1)])
def test_linear_wgrad_256_tile_repeats_bit_for_bit_beside_busy_neighbours(M, N, K, sk):
    """csrc/gemm_tn_pp.hip keeps three LDS-DMA units in flight across raw barriers behind counted vmcnt / lgkmcnt waits: a wait that
    is one short, or a region restaged a phase early, shows up as a result that depends on timing.  The step's shapes (LSTM chunk
    direct and split, h2att chunk, logit dW) and a ragged one, 60 launches each beside an HBM-bound and an MFMA-bound neighbour on
    another stream: every launch must give the first launch's bits, and those must be the product."""
    from unpaired_image_captioning_amd import _lib as L
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    lda = (M + 7) // 8 * 8
    A = torch.randn(K, lda, device="cuda", generator=g).bfloat16()
    B = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    ref = A[:, :M].float().t() @ B.float()
    ws = torch.empty(max(sk, 1) * M * N * 4, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    hog_a = torch.randn(32 << 20, device="cuda")
    hog_b = torch.empty_like(hog_a)
    X = torch.randn(4096, 4096, device="cuda").bfloat16()
    first = None
    for it in range(60):
        mode = it % 4
        with torch.cuda.stream(side):
            if mode in (1, 3):
                hog_b.copy_(hog_a)
            if mode in (2, 3):
                torch.matmul(X, X)
        dW = torch.full((M, N), float("nan"), device="cuda")
        L.check(lib.uic_linear_wgrad(L.BF16, M, N, K, L.ptr(A), lda, L.ptr(B), N, L.ptr(dW), N, L.ptr(ws), ws.numel(), TN_256 | TN_SK(sk),
                                     L.stream()), "linear_wgrad")
        torch.cuda.synchronize()
        if first is None:
            first = dW
            assert (dW - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
        else:
            assert torch.equal(dW, first), (it, mode, int((dW != first).sum()))


def test_lm_criterion_matches_oracle():
    from unpaired_image_captioning_amd.misc.criterion import LanguageModelCriterion
    g = torch.Generator().manual_seed(4)
    N, T, V1 = 6, 7, 51
    logp = torch.log_softmax(torch.randn(N, T, V1, generator=g), 2)
    target = torch.randint(0, V1, (N, T + 2), generator=g)
    mask = (torch.rand(N, T + 2, generator=g) > 0.3).float()
    lr = logp.clone().requires_grad_(True)
    ref = O.lm_criterion(lr, target, mask)
    ref.backward()
    ld = logp.cuda().requires_grad_(True)
    out = LanguageModelCriterion()(ld, target.cuda(), mask.cuda())
    out.backward()
    assert abs(out.item() - ref.item()) < 1e-5
    assert rel_err(ld.grad, lr.grad) < 1e-5


def test_library_exports_every_declared_symbol():
    L = _lib()
    lib = L.load()
    for name in L.EXPORTS:
        assert hasattr(lib, name), name
    assert lib.uic_version() >= 100


TN_128, TN_256 = 0x100, 0x200          # uic_linear_wgrad's kernel bits (include/uic_hip.h)


def TN_SK(n):
    return (n & 0xff) << 16


@pytest.mark.parametrize("M,N,K,ldy,ldx", [(128, 128, 64, 128, 128), (256, 384, 640, 256, 384), (2048, 512, 2560, 2048, 512),
                                            (1192, 256, 1280, 1216, 320), (512, 2048, 4608, 512, 2048), (2048, 1664, 2560, 2048, 1664),
                                            (520, 640, 3840, 520, 768)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_linear_wgrad_tn_matches_matmul(M, N, K, ldy, ldx, accumulate):
    """dW = dY^T X straight from row-major [K, M] / [K, N] bf16 operands (transposing LDS reads, no transposed copies):
    exact products, f32 accumulation -> compare with a float64 matmul of the same bf16 values; ragged M (1192, 520), leading
    dimensions larger than the width, one K round (64) up to 72, split-K chosen from the workspace size.  Both kernels: the
    128 x 128 tile (csrc/gemm_tn.hip) and the 256 x 256 ping-pong tile (csrc/gemm_tn_pp.hip: whole tiles, half-empty column tiles
    (N = 384, 640, 1664), a ragged last row tile, direct stores and every split-K that leaves whole pairs of K tiles)."""
    from unpaired_image_captioning_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(K, ldy, generator=g).bfloat16()
    X = torch.randn(K, ldx, generator=g).bfloat16()
    ref = dY[:, :M].double().t() @ X[:, :N].double()
    dW0 = torch.randn(M, N, generator=g)
    ws = torch.empty(8 * M * N * 4, dtype=torch.uint8, device="cuda")
    dYd, Xd = dY.cuda(), X.cuda()
    want = ref + (dW0.double() if accumulate else 0)
    tol = 2e-5 * max(1.0, want.abs().max().item()) * (K ** 0.5)
    hows = [0, TN_128]
    if K % 128 == 0:
        hows += [TN_256 | TN_SK(sk) for sk in (1, 2, 3, 5, 6) if K % (128 * sk) == 0]
    outs = {}
    for how in hows:
        dW = dW0.clone().cuda()
        L.check(lib.uic_linear_wgrad(L.BF16, M, N, K, L.ptr(dYd), ldy, L.ptr(Xd), ldx, L.ptr(dW), N, L.ptr(ws), ws.numel(), accumulate | how,
                                     L.stream()), "linear_wgrad")
        err = (dW.cpu().double() - want).abs().max().item()
        assert err <= tol, (hex(how), err)
        outs[how] = dW
        # deterministic: the same call again gives the same bits (fixed slice order, no atomics)
        dW2 = dW0.clone().cuda()
        L.check(lib.uic_linear_wgrad(L.BF16, M, N, K, L.ptr(dYd), ldy, L.ptr(Xd), ldx, L.ptr(dW2), N, L.ptr(ws), ws.numel(), accumulate | how,
                                     L.stream()), "linear_wgrad")
        assert torch.equal(dW, dW2), hex(how)
    dW = outs[0]
    # a workspace with room for a single slice only (no split-K) gives the same result up to summation order
    dW2 = dW0.clone().cuda()
    ws1 = torch.empty(M * N * 4, dtype=torch.uint8, device="cuda")
    L.check(lib.uic_linear_wgrad(L.BF16, M, N, K, L.ptr(dYd), ldy, L.ptr(Xd), ldx, L.ptr(dW2), N, L.ptr(ws1), ws1.numel(), accumulate,
                                 L.stream()), "linear_wgrad")
    assert (dW2 - dW).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
    # a forced kernel that cannot take the problem is an argument error (odd number of K tiles / slices that split a tile pair)
    if K % 128 != 0:
        assert lib.uic_linear_wgrad(L.BF16, M, N, K, L.ptr(dYd), ldy, L.ptr(Xd), ldx, L.ptr(dW), N, L.ptr(ws), ws.numel(), TN_256, L.stream()) != 0
    else:
        assert lib.uic_linear_wgrad(L.BF16, M, N, K, L.ptr(dYd), ldy, L.ptr(Xd), ldx, L.ptr(dW), N, L.ptr(ws), ws.numel(), TN_256 | TN_SK(7), L.stream()) != 0
    # ineligible shapes are argument errors, not silent fallbacks
    assert lib.uic_linear_wgrad(L.BF16, M, N, K - 8, L.ptr(dYd), ldy, L.ptr(Xd), ldx, L.ptr(dW), N, L.ptr(ws), ws.numel(), 0, L.stream()) != 0
