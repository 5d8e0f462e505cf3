// CNN sentence discriminator of the unpaired / adversarial configuration (BASELINE configs[3]; north_star "the
// sentence-discriminator forward/backward").  THE REFERENCE TREE HOLDS NO DISCRIMINATOR CODE (SURVEY finding 2): the
// architecture below is this package's own statement of the usual text-CNN critic (Kim 2014 / SeqGAN) and its parity is
// UNPINNED -- oracle/discriminator.py restates the same spec on the CPU, nothing of the reference backs it.
//
//   x[n, t]   = relu(Emb[tok[n, t]])                                    t < L            (embedding table [V1, E])
//   y_w[n, t] = relu(b_w + sum_{j < w} W_w[:, j, :] x[n, t + j])        x = 0 beyond L   (nw widths w_i, F filters each)
//   p[n]      = concat_w max_t y_w[n, t]                                                 [Ft = nw * F]
//   g, h      = sigmoid(W_g p + b_g), relu(W_h p + b_h);   z = g h + (1 - g) p           (highway; W_gh = [W_g; W_h])
//   logit[n]  = w_o . dropout(z) + b_o;    D(sentence) = sigmoid(logit)
//
// Mapping: rows are time-major (m = t N + n, the captioner's layout), so a shift by j time steps is a row offset of
// j N and each convolution is ONE multi-segment NT GEMM over the embedded rows (K segments j = 0 .. w-1, A_j = X + j N E,
// B_j = W_w[:, j, :]) with bias + ReLU in the epilogue -- no im2col buffer; its weight gradient is one multi-segment TN GEMM,
// its input gradient a multi-segment NT GEMM over row-shifted dY.  Max-over-time, the highway gate math, the output dot
// product and the BCE loss are small HBM-bound kernels below.
#include "uic_common.h"
#include "uic_host.h"
#include "../../include/uic_hip.h"
#include <string.h>

namespace {

constexpr int NT_ = 256;
constexpr int PADT = UIC_DISC_MAX_WIDTH - 1;     // zero time slots behind X / in front of dY

struct DLayout {
  void* x;                         // [(L + PADT) N, E]           relu(Emb[tok]); the PADT extra time slots are zero
  void* y[UIC_DISC_MAX_WIDTHS];    // [L N, F]                    relu(conv_w)
  float* p; void* p_op; int* arg;  // [N, Ft]
  float* gh;                       // [N, 2 Ft] highway pre-activations (gate | transform)
  float* logits;                   // [N]
  // backward
  void* dgh; float* dp; float* tz;                  // [N, 2Ft] operand dtype, [N, Ft], [N, Ft] (dlogit * dropped z: d w_o rows)
  void* dy[UIC_DISC_MAX_WIDTHS];   // [(PADT + L) N, F]           PADT zero time slots in FRONT, then the rows
  float* dx;                       // [L N, E]
  // operand-dtype weight copies / transposes
  void* c_conv[UIC_DISC_MAX_WIDTHS];                // [F, w E]
  void* convT[UIC_DISC_MAX_WIDTHS];                 // w x [E, F]
  void* c_hw; void* hwT;                            // [2Ft, Ft], [Ft, 2Ft]
  void* tA; void* tB; float* colscratch; size_t colscratch_floats; float* slab; size_t slab_bytes;
  float* scal;
  int* embed_scratch;   // uic_embed_bwd_sorted_launch
  size_t total;
};

int check_dims(const uic_disc_dims* d) {
  UIC_REQUIRE(d, "disc: null dims");
  UIC_REQUIRE(d->dtype == UIC_F32 || d->dtype == UIC_BF16, "disc: bad dtype %d", d->dtype);
  UIC_REQUIRE(d->N > 0 && d->L > 0 && d->V1 > 0, "disc: N=%d L=%d V1=%d", d->N, d->L, d->V1);
  UIC_REQUIRE(d->E > 0 && d->E % 8 == 0 && d->F > 0 && d->F % 8 == 0, "disc: E=%d and F=%d must be multiples of 8", d->E, d->F);
  UIC_REQUIRE(d->nw >= 1 && d->nw <= UIC_DISC_MAX_WIDTHS, "disc: %d filter widths (1..%d)", d->nw, UIC_DISC_MAX_WIDTHS);
  for (int i = 0; i < d->nw; ++i)
    UIC_REQUIRE(d->widths[i] >= 1 && d->widths[i] <= UIC_DISC_MAX_WIDTH && d->widths[i] <= d->L, "disc: filter width %d outside [1, min(%d, L)]", d->widths[i], UIC_DISC_MAX_WIDTH);
  return UIC_OK;
}

DLayout make_layout(const uic_disc_dims& d, void* ws) {
  DLayout L;
  memset(&L, 0, sizeof(L));
  Bump b{(char*)ws, 0};
  const size_t Sz = uic_dtype_size(d.dtype);
  const size_t N = d.N, T = d.L, E = d.E, F = d.F, Ft = (size_t)d.nw * d.F, M = T * N;
  L.x = b.take((T + PADT) * N * E * Sz);
  for (int i = 0; i < d.nw; ++i) L.y[i] = b.take(M * F * Sz);
  L.p = (float*)b.take(N * Ft * 4);
  L.p_op = b.take(N * Ft * Sz);
  L.arg = (int*)b.take(N * Ft * 4);
  L.gh = (float*)b.take(N * 2 * Ft * 4);
  L.logits = (float*)b.take(N * 4);
  L.dgh = b.take(N * 2 * Ft * Sz);
  L.dp = (float*)b.take(N * Ft * 4);
  L.tz = (float*)b.take(N * Ft * 4);
  for (int i = 0; i < d.nw; ++i) L.dy[i] = b.take((PADT + T) * N * F * Sz);
  L.dx = (float*)b.take(M * E * 4);
  for (int i = 0; i < d.nw; ++i) {
    L.c_conv[i] = b.take(F * d.widths[i] * E * Sz);
    L.convT[i] = b.take((size_t)d.widths[i] * E * F * Sz);
  }
  L.c_hw = b.take(2 * Ft * Ft * Sz);
  L.hwT = b.take(Ft * 2 * Ft * Sz);
  const size_t Mp = rup8(M), Np = rup8(N);
  size_t ta = F * Mp > 2 * Ft * Np ? F * Mp : 2 * Ft * Np;
  size_t tb = (size_t)UIC_DISC_MAX_WIDTH * E * Mp > Ft * Np ? (size_t)UIC_DISC_MAX_WIDTH * E * Mp : Ft * Np;
  L.tA = b.take(ta * Sz);
  L.tB = b.take(tb * Sz);
  size_t maxcols = 2 * Ft > F ? 2 * Ft : F;
  if (E > maxcols) maxcols = E;
  L.colscratch_floats = 128 * maxcols;
  L.colscratch = (float*)b.take(L.colscratch_floats * 4);
  size_t sl = 8 * F * (size_t)UIC_DISC_MAX_WIDTH * E * 4;
  if (8 * 2 * Ft * Ft * 4 > sl) sl = 8 * 2 * Ft * Ft * 4;
  L.slab_bytes = sl;
  L.slab = (float*)b.take(sl);
  L.scal = (float*)b.take(256);
  L.embed_scratch = (int*)b.take(uic_embed_bwd_sorted_scratch_ints(d.N, d.L, d.V1, d.E) * 4);
  L.total = (b.off + 255) & ~(size_t)255;
  return L;
}

// ---------------------------------------------------------------- kernels
// p[n, c] = max_t y[(t N + n), f] (y >= 0 after the ReLU; first maximum wins), c = col0 + f
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ y, int N, int TS, int F, int Ft, int col0, float* __restrict__ p,
                                   T* __restrict__ p_op, int* __restrict__ arg) {
  const size_t total = (size_t)N * F;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / F), f = (int)(i - (size_t)n * F);
    float best = uic_to_f(y[(size_t)n * F + f]);
    int bt = 0;
    for (int t = 1; t < TS; ++t) {
      const float v = uic_to_f(y[((size_t)t * N + n) * F + f]);
      if (v > best) { best = v; bt = t; }
    }
    const size_t o = (size_t)n * Ft + col0 + f;
    p[o] = best; p_op[o] = uic_from_f<T>(best); arg[o] = bt;
  }
}
// dy[(t N + n), f] = (t == arg && p > 0) ? dp[n, c] : 0   -- every row of the width's block is written
template <typename T>
__global__ void maxpool_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p, const int* __restrict__ arg, int N, int TS,
                                   int F, int Ft, int col0, T* __restrict__ dy) {
  const size_t total = (size_t)TS * N * F;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const size_t row = i / F;
    const int t = (int)(row / N), n = (int)(row - (size_t)t * N);
    const size_t o = (size_t)n * Ft + col0 + f;
    dy[i] = uic_from_f<T>((arg[o] == t && p[o] > 0.f) ? dp[o] : 0.f);
  }
}
// highway + output layer of row n (one workgroup per row): logit = w_o . dropout(g h + (1 - g) p) + b_o
__global__ void head_fwd_kernel(const float* __restrict__ gh, const float* __restrict__ p, const float* __restrict__ w_o,
                                const float* __restrict__ b_o, int Ft, float drop_p, unsigned seed, float* __restrict__ logits) {
  __shared__ float s_part[NT_ / 64];
  const int n = blockIdx.x;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float acc = 0.f;
  for (int a = threadIdx.x; a < Ft; a += blockDim.x) {
    const float g = uic_sigmoid(gh[(size_t)n * 2 * Ft + a]);
    const float h = fmaxf(gh[(size_t)n * 2 * Ft + Ft + a], 0.f);
    const float pv = p[(size_t)n * Ft + a];
    float z = g * h + (1.f - g) * pv;
    if (drop_p > 0.f) z *= uic_drop_scale(seed, UIC_SITE_DISC, (unsigned)(n * Ft + a), drop_p, inv_keep);
    acc += z * w_o[a];
  }
  acc = uic_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = b_o[0];
    for (int i = 0; i < (int)blockDim.x / 64; ++i) r += s_part[i];
    logits[n] = r;
  }
}
// backward of the same: dgh (operand dtype), dp (direct path of the highway), tz[n, a] = dlogit[n] * dropped z (rows of d w_o)
template <typename T>
__global__ void head_bwd_kernel(const float* __restrict__ gh, const float* __restrict__ p, const float* __restrict__ w_o,
                                const float* __restrict__ dlogits, int Ft, float drop_p, unsigned seed, T* __restrict__ dgh,
                                float* __restrict__ dp, float* __restrict__ tz) {
  const int n = blockIdx.x;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const float dl = dlogits[n];
  for (int a = threadIdx.x; a < Ft; a += blockDim.x) {
    const float gpre = gh[(size_t)n * 2 * Ft + a], hpre = gh[(size_t)n * 2 * Ft + Ft + a];
    const float g = uic_sigmoid(gpre), h = fmaxf(hpre, 0.f), pv = p[(size_t)n * Ft + a];
    const float z = g * h + (1.f - g) * pv;
    const float ds = drop_p > 0.f ? uic_drop_scale(seed, UIC_SITE_DISC, (unsigned)(n * Ft + a), drop_p, inv_keep) : 1.f;
    const float dz = dl * w_o[a] * ds;
    tz[(size_t)n * Ft + a] = dl * z * ds;
    dgh[(size_t)n * 2 * Ft + a] = uic_from_f<T>(dz * (h - pv) * g * (1.f - g));
    dgh[(size_t)n * 2 * Ft + Ft + a] = uic_from_f<T>(hpre > 0.f ? dz * g : 0.f);
    dp[(size_t)n * Ft + a] = dz * (1.f - g);
  }
}
// binary cross entropy with logits, mean over the N sentences; dlogits = (sigmoid(l) - y) / N
__global__ void bce_kernel(const float* __restrict__ logits, const float* __restrict__ labels, int N, float* __restrict__ loss,
                           float* __restrict__ dlogits) {
  __shared__ float s_part[NT_ / 64];
  float acc = 0.f;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const float l = logits[n], y = labels[n];
    acc += fmaxf(l, 0.f) - l * y + log1pf(expf(-fabsf(l)));
    if (dlogits) dlogits[n] = (uic_sigmoid(l) - y) / (float)N;
  }
  acc = uic_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = 0.f;
    for (int i = 0; i < (int)blockDim.x / 64; ++i) r += s_part[i];
    loss[0] = r / (float)N;
  }
}
__global__ void sigmoid_kernel(const float* __restrict__ x, int n, float* __restrict__ y) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = uic_sigmoid(x[i]);
}

inline int grid1(size_t n) {
  size_t g = (n + NT_ - 1) / NT_;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
#define DISC_DISPATCH(dt, BF, F32) do { if ((dt) == UIC_BF16) { BF; } else { F32; } } while (0)

// operand-dtype copies + the transposes the two dX GEMMs need (the model is small: rebuilt by every call) -- ONE cast launch and
// ONE transpose launch (bf16; up to 5 + 11 jobs), not one launch per tensor and convolution tap: at this model size a training
// step is made of launches
int refresh(const uic_disc_dims& d, const uic_disc_weights* w, const DLayout& L, bool with_backward, hipStream_t s) {
  const int dt = d.dtype, E = d.E, F = d.F, Ft = d.nw * d.F;
  for (int i = 0; i < d.nw; ++i) UIC_REQUIRE(w->conv_w[i] && w->conv_b[i], "disc: null conv weights of width %d", d.widths[i]);
  int taps = 0;
  for (int i = 0; i < d.nw; ++i) taps += d.widths[i];
  const bool multi = dt == UIC_BF16 && d.nw + 1 <= UIC_CAST_MULTI && taps + 1 <= UIC_TRANSPOSE_MULTI;
  if (multi) {
    const float* src[UIC_CAST_MULTI];
    void* dst[UIC_CAST_MULTI];
    size_t n[UIC_CAST_MULTI];
    for (int i = 0; i < d.nw; ++i) { src[i] = w->conv_w[i]; dst[i] = L.c_conv[i]; n[i] = (size_t)F * d.widths[i] * E; }
    src[d.nw] = w->hw_w; dst[d.nw] = L.c_hw; n[d.nw] = (size_t)2 * Ft * Ft;
    UIC_TRY(uic_cast_f32_multi_launch(dt, d.nw + 1, src, dst, n, s));
    if (with_backward) {
      UicTransposeJob jobs[UIC_TRANSPOSE_MULTI];
      int nj = 0;
      for (int i = 0; i < d.nw; ++i)
        for (int j = 0; j < d.widths[i]; ++j)   // W_w[:, j, :] ([F, E], row stride w E) -> [E, F]
          jobs[nj++] = UicTransposeJob{off(L.c_conv[i], (size_t)j * E, dt), offw(L.convT[i], (size_t)j * E * F, dt), F, E, d.widths[i] * E, F};
      jobs[nj++] = UicTransposeJob{L.c_hw, L.hwT, 2 * Ft, Ft, Ft, 2 * Ft};
      UIC_TRY(uic_transpose_multi_launch(dt, nj, jobs, s));
    }
    return UIC_OK;
  }
  for (int i = 0; i < d.nw; ++i) {
    const int wd = d.widths[i];
    UIC_TRY(uic_cast_f32_launch(dt, w->conv_w[i], L.c_conv[i], (size_t)F * wd * E, s));
    if (with_backward)
      for (int j = 0; j < wd; ++j)   // W_w[:, j, :] ([F, E], row stride w E) -> [E, F]
        UIC_TRY(uic_transpose_launch(dt, off(L.c_conv[i], (size_t)j * E, dt), F, E, wd * E, offw(L.convT[i], (size_t)j * E * F, dt), F, s));
  }
  UIC_TRY(uic_cast_f32_launch(dt, w->hw_w, L.c_hw, (size_t)2 * Ft * Ft, s));
  if (with_backward) UIC_TRY(uic_transpose_launch(dt, L.c_hw, 2 * Ft, Ft, Ft, L.hwT, 2 * Ft, s));
  return UIC_OK;
}

int forward(const uic_disc_dims& d, const uic_disc_weights* w, const int64_t* tokens, int ld_tokens, int training, unsigned seed,
            const DLayout& L, bool with_backward, hipStream_t s) {
  const int dt = d.dtype, N = d.N, T = d.L, E = d.E, F = d.F, Ft = d.nw * d.F, M = T * N;
  UIC_TRY(refresh(d, w, L, with_backward, s));
  UIC_TRY(uic_embed_fwd_launch(dt, w->embed_w, d.V1, E, tokens, ld_tokens, N, T, 0.f, 0u, 0u, 0, 1, L.x, s));
  UIC_TRY(uic_fill_launch(offw(L.x, (size_t)M * E, dt), 0, (size_t)PADT * N * E * uic_dtype_size(dt), s));
  for (int i = 0; i < d.nw; ++i) {
    const int wd = d.widths[i];
    UicGemmParams g = gemm_base(dt, M, F);
    for (int j = 0; j < wd; ++j) add_seg(g, off(L.x, (size_t)j * N * E, dt), E, off(L.c_conv[i], (size_t)j * E, dt), wd * E, E);
    g.C = L.y[i]; g.ldc = F; g.bias = w->conv_b[i]; g.flags = UIC_GEMM_RELU;
    UIC_TRY(uic_gemm_launch(g, s));
    DISC_DISPATCH(dt,
      hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid1((size_t)N * F)), dim3(NT_), 0, s, (const bf16_t*)L.y[i], N, T, F, Ft, i * F, L.p, (bf16_t*)L.p_op, L.arg),
      hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid1((size_t)N * F)), dim3(NT_), 0, s, (const float*)L.y[i], N, T, F, Ft, i * F, L.p, (float*)L.p_op, L.arg));
    UIC_LAUNCH_CHECK("disc maxpool_fwd");
  }
  {
    UicGemmParams g = gemm_base(dt, N, 2 * Ft);
    add_seg(g, L.p_op, Ft, L.c_hw, Ft, Ft);
    g.C = L.gh; g.ldc = 2 * Ft; g.bias = w->hw_b; g.flags = UIC_GEMM_OUT_F32;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  hipLaunchKernelGGL(head_fwd_kernel, dim3(N), dim3(NT_), 0, s, L.gh, L.p, w->out_w, w->out_b, Ft, training ? d.drop_p : 0.f, seed, L.logits);
  UIC_LAUNCH_CHECK("disc head_fwd");
  return UIC_OK;
}

}  // namespace

extern "C" {

size_t uic_disc_workspace_bytes(const uic_disc_dims* d) {
  if (check_dims(d)) return 0;
  return make_layout(*d, nullptr).total;
}

int uic_disc_forward(const uic_disc_dims* d, const uic_disc_weights* w, const int64_t* tokens, int32_t ld_tokens, int32_t training,
                     uint32_t seed, void* workspace, float* logits_out, float* prob_out, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && tokens && workspace && w->embed_w && w->hw_w && w->hw_b && w->out_w && w->out_b, "disc_forward: null pointer");
  UIC_REQUIRE(ld_tokens >= d->L, "disc_forward: ld_tokens=%d < L=%d", ld_tokens, d->L);
  hipStream_t s = (hipStream_t)stream;
  const DLayout L = make_layout(*d, workspace);
  UIC_TRY(forward(*d, w, tokens, ld_tokens, training, seed, L, training != 0, s));
  if (logits_out) UIC_TRY(uic_copy_launch(logits_out, L.logits, (size_t)d->N * 4, s));
  if (prob_out) {
    hipLaunchKernelGGL(sigmoid_kernel, dim3(grid1(d->N)), dim3(NT_), 0, s, L.logits, d->N, prob_out);
    UIC_LAUNCH_CHECK("disc sigmoid");
  }
  return UIC_OK;
}

int uic_disc_bce(const float* logits, const float* labels, int32_t N, float* loss_out, float* dlogits_out, void* stream) {
  UIC_REQUIRE(logits && labels && loss_out && N > 0, "disc_bce: null pointer or N=%d", N);
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(NT_), 0, (hipStream_t)stream, logits, labels, N, loss_out, dlogits_out);
  UIC_LAUNCH_CHECK("disc bce");
  return UIC_OK;
}

int uic_disc_backward(const uic_disc_dims* d, const uic_disc_weights* w, const int64_t* tokens, int32_t ld_tokens, int32_t training,
                      uint32_t seed, void* workspace, const float* dlogits, const uic_disc_weights* G, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && tokens && workspace && dlogits && G, "disc_backward: null pointer");
  UIC_REQUIRE(G->embed_w && G->hw_w && G->hw_b && G->out_w && G->out_b, "disc_backward: null gradient tensor");
  hipStream_t s = (hipStream_t)stream;
  const DLayout L = make_layout(*d, workspace);
  const int dt = d->dtype, N = d->N, T = d->L, E = d->E, F = d->F, Ft = d->nw * d->F, M = T * N;
  const size_t Sz = uic_dtype_size(dt);
  // output layer + highway gate math
  DISC_DISPATCH(dt,
    hipLaunchKernelGGL(head_bwd_kernel<bf16_t>, dim3(N), dim3(NT_), 0, s, L.gh, L.p, w->out_w, dlogits, Ft, training ? d->drop_p : 0.f, seed, (bf16_t*)L.dgh, L.dp, L.tz),
    hipLaunchKernelGGL(head_bwd_kernel<float>, dim3(N), dim3(NT_), 0, s, L.gh, L.p, w->out_w, dlogits, Ft, training ? d->drop_p : 0.f, seed, (float*)L.dgh, L.dp, L.tz));
  UIC_LAUNCH_CHECK("disc head_bwd");
  UIC_TRY(uic_colsum_launch(UIC_F32, L.tz, N, Ft, Ft, G->out_w, L.colscratch, L.colscratch_floats, s));
  UIC_TRY(uic_colsum_launch(UIC_F32, dlogits, N, 1, 1, G->out_b, L.colscratch, L.colscratch_floats, s));
  {  // d W_gh = dGH^T p,  d b_gh,  d p += dGH W_gh
    const UicGemmTnSeg seg{L.p_op, Ft, Ft};
    const WDest d1{G->hw_w, Ft, 0, Ft};
    UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.dgh, 2 * Ft, 2 * Ft, &seg, 1, N, &d1, 1, s, false, L.tA, L.tB));
    UIC_TRY(uic_colsum_launch(dt, L.dgh, N, 2 * Ft, 2 * Ft, G->hw_b, L.colscratch, L.colscratch_floats, s));
    UicGemmParams g = gemm_base(dt, N, Ft);
    add_seg(g, L.dgh, 2 * Ft, L.hwT, 2 * Ft, 2 * Ft);
    g.C = L.dp; g.ldc = Ft; g.flags = UIC_GEMM_OUT_F32 | UIC_GEMM_ACCUM;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  // max-over-time routing, conv weight gradients, d x
  int nseg_total = 0;
  UicGemmParams gx = gemm_base(dt, M, E);
  bool first = true;
  auto flush = [&]() -> int {
    if (gx.nseg == 0) return UIC_OK;
    gx.C = L.dx; gx.ldc = E; gx.flags = UIC_GEMM_OUT_F32 | (first ? 0 : UIC_GEMM_ACCUM);
    first = false;
    const int rc = uic_gemm_launch(gx, s);
    gx = gemm_base(dt, M, E);
    return rc;
  };
  for (int i = 0; i < d->nw; ++i) {
    const int wd = d->widths[i];
    UIC_REQUIRE(G->conv_w[i] && G->conv_b[i], "disc_backward: null conv gradient of width %d", wd);
    void* dy_rows = offw(L.dy[i], (size_t)PADT * N * F, dt);
    UIC_TRY(uic_fill_launch(L.dy[i], 0, (size_t)PADT * N * F * Sz, s));
    DISC_DISPATCH(dt,
      hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid1((size_t)M * F)), dim3(NT_), 0, s, L.dp, L.p, L.arg, N, T, F, Ft, i * F, (bf16_t*)dy_rows),
      hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid1((size_t)M * F)), dim3(NT_), 0, s, L.dp, L.p, L.arg, N, T, F, Ft, i * F, (float*)dy_rows));
    UIC_LAUNCH_CHECK("disc maxpool_bwd");
    {  // d W_w[:, j, :] = dY_w^T X[. + j N]   (one TN GEMM, w column segments)
      UicGemmTnSeg segs[UIC_DISC_MAX_WIDTH];
      WDest dd[UIC_DISC_MAX_WIDTH];
      for (int j = 0; j < wd; ++j) {
        segs[j] = UicGemmTnSeg{off(L.x, (size_t)j * N * E, dt), E, E};
        dd[j] = WDest{G->conv_w[i] + (size_t)j * E, wd * E, j * E, E};
      }
      UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, dy_rows, F, F, segs, wd, M, dd, wd, s, false, L.tA, L.tB));
      UIC_TRY(uic_colsum_launch(dt, dy_rows, M, F, F, G->conv_b[i], L.colscratch, L.colscratch_floats, s));
    }
    for (int j = 0; j < wd; ++j) {   // d x[m] += dY_w[m - j N] W_w[:, j, :]
      if (gx.nseg == UIC_GEMM_MAX_SEG) UIC_TRY(flush());
      add_seg(gx, off(L.dy[i], (size_t)(PADT - j) * N * F, dt), F, off(L.convT[i], (size_t)j * E * F, dt), F, F);
      ++nseg_total;
    }
  }
  UIC_TRY(flush());
  UIC_TRY(uic_embed_bwd_sorted_launch(dt, L.dx, L.x, tokens, ld_tokens, N, T, d->V1, E, 0.f, -1, G->embed_w, L.embed_scratch, s));
  return UIC_OK;
}

}  // extern "C"
