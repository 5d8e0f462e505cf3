// extern "C" surface of libuic_hip.so for the single operators + error reporting.
#include "uic_common.h"
#include "uic_host.h"
#include "../../include/uic_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

static thread_local char g_err[512] = "";

void uic_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int uic_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return UIC_OK;
  uic_set_error("%s: %s", what, hipGetErrorString(e));
  return (int)e > 0 ? (int)e : 1;
}

namespace {
// LanguageModelCriterion on materialised log-probs (P/misc/criterion.py:143-150)
__global__ void lm_crit_rows_kernel(int N, int T, int V1, const float* logp, const int64_t* target, int ldt,
                                    const float* mask, int ldm, float* row_loss, float* row_mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * T) return;
  const int n = i / T, t = i - n * T;
  long y = target[(size_t)n * ldt + t];
  if (y < 0 || y >= V1) y = 0;
  const float m = mask[(size_t)n * ldm + t];
  row_loss[i] = -logp[((size_t)n * T + t) * V1 + y] * m;
  row_mask[i] = m;
}
__global__ void lm_crit_final_kernel(const float* row_loss, const float* row_mask, int n, float* out) {
  __shared__ float s_a[256], s_b[256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { a += row_loss[i]; b += row_mask[i]; }
  s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s_a[threadIdx.x] += s_a[threadIdx.x + o]; s_b[threadIdx.x] += s_b[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = s_a[0] / s_b[0]; out[1] = s_b[0]; }
}
__global__ void lm_crit_bwd_kernel(int N, int T, int V1, const int64_t* target, int ldt, const float* mask, int ldm,
                                   const float* loss_den, float grad_out, float* dlogp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * T) return;
  const int n = i / T, t = i - n * T;
  long y = target[(size_t)n * ldt + t];
  if (y < 0 || y >= V1) y = 0;
  dlogp[((size_t)n * T + t) * V1 + y] = -mask[(size_t)n * ldm + t] / loss_den[1] * grad_out;
}
// RewardCriterion (P/misc/criterion.py:117-124), one block
__global__ void reward_crit_kernel(int N, int L, const float* logp, const int64_t* seq, const float* reward, float* loss_out, float* dlogp) {
  // One workgroup (the sums are small and their order is part of the result).  Eight elements per thread in flight: with one
  // element per trip the 40 trips of a 640 x 16 step were 40 memory latencies in a row (40 us between the reward and the backward
  // pass of the self-critical step).  The per-thread sums still run over i = tid, tid + 256, ... in that order.
  __shared__ float s_a[256], s_b[256];
  const int total = N * L;
  float a = 0.f, b = 0.f;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 8) {
    float lp[8], rw[8], m[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      const bool in = i < total;
      const int n = in ? i / L : 0, t = in ? i - n * L : 0;
      lp[u] = in ? logp[i] : 0.f;
      rw[u] = in ? reward[i] : 0.f;
      m[u] = !in ? 0.f : (t == 0 || seq[(size_t)n * L + t - 1] > 0) ? 1.f : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * 256 < total) { a += -lp[u] * rw[u] * m[u]; b += m[u]; }
  }
  s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s_a[threadIdx.x] += s_a[threadIdx.x + o]; s_b[threadIdx.x] += s_b[threadIdx.x + o]; }
    __syncthreads();
  }
  const float den = s_b[0];
  if (threadIdx.x == 0) loss_out[0] = s_a[0] / den;
  if (dlogp)
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 8) {
      float rw[8], m[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const bool in = i < total;
        const int n = in ? i / L : 0, t = in ? i - n * L : 0;
        rw[u] = in ? reward[i] : 0.f;
        m[u] = !in ? 0.f : (t == 0 || seq[(size_t)n * L + t - 1] > 0) ? 1.f : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * 256 < total) dlogp[i0 + u * 256] = -rw[u] * m[u] / den;
    }
}
}  // namespace

extern "C" {

const char* uic_last_error_string(void) { return g_err; }
int uic_version(void) { return 100; }

int uic_linear(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* B, int32_t ldb,
               void* C, int32_t ldc, const float* bias, int32_t flags, void* stream) {
  UicGemmParams g;
  memset(&g, 0, sizeof(g));
  g.dtype = dtype; g.M = M; g.N = N; g.nseg = 1;
  g.seg[0].A = A; g.seg[0].B = B; g.seg[0].K = K; g.seg[0].lda = lda; g.seg[0].ldb = ldb;
  g.C = C; g.ldc = ldc; g.bias = bias; g.flags = flags;
  return uic_gemm_launch(g, (hipStream_t)stream);
}

int uic_linear_f32a(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const void* B, int32_t ldb,
                    void* C, int32_t ldc, const float* bias, int32_t flags, void* a_bf16, int32_t ld_a_bf16, void* stream) {
  UIC_REQUIRE(A && B && C, "linear_f32a: null pointer");
  UicGemmParams g;
  memset(&g, 0, sizeof(g));
  g.dtype = UIC_BF16; g.M = M; g.N = N; g.nseg = 1;
  g.seg[0].A = A; g.seg[0].B = B; g.seg[0].K = K; g.seg[0].lda = lda; g.seg[0].ldb = ldb;
  g.C = C; g.ldc = ldc; g.bias = bias; g.flags = flags;
  g.a_f32 = 1; g.a_copy = a_bf16; g.ld_a_copy = ld_a_bf16;
  UIC_REQUIRE(uic_gemm_pp_eligible(g), "linear_f32a: needs K %% 128 == 0, N %% 4 == 0, lda %% 4 == 0, 16-byte aligned operands, "
              "ld_a_bf16 %% 8 == 0 and >= K, A below 4 GB (M=%d N=%d K=%d lda=%d)", M, N, K, lda);
  return uic_gemm_launch(g, (hipStream_t)stream);
}

int uic_linear_partials(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* B, int32_t ldb,
                        float* slab, int32_t splitk, void* stream) {
  UIC_REQUIRE(slab && splitk >= 1 && splitk <= 16, "linear_partials: slab / splitk=%d", splitk);
  UIC_REQUIRE(uic_gemm_glds_eligible(dtype, K), "linear_partials: K=%d must be a multiple of one 128-byte round", K);
  UicGemmParams g;
  memset(&g, 0, sizeof(g));
  g.dtype = dtype; g.M = M; g.N = N; g.nseg = 1;
  g.seg[0].A = A; g.seg[0].B = B; g.seg[0].K = K; g.seg[0].lda = lda; g.seg[0].ldb = ldb;
  g.slab = slab; g.splitk = splitk;
  return uic_gemm_launch(g, (hipStream_t)stream);
}

int uic_linear_wgrad(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* dY, int32_t ldy, const void* X, int32_t ldx,
                     float* dW, int32_t ldw, void* workspace, size_t workspace_bytes, int32_t accumulate, void* stream) {
  UIC_REQUIRE(dY && X && dW && workspace, "linear_wgrad: null pointer");
  UIC_REQUIRE(dtype == UIC_BF16, "linear_wgrad: the transposing-read kernel is bf16 only (dtype=%d)", dtype);
  const WDest d1{dW, ldw, 0, N};
  const UicGemmTnSeg seg{X, ldx, N};
  bool done = false;
  UIC_TRY(wgrad_tn((float*)workspace, workspace_bytes, dtype, dY, ldy, M, &seg, 1, K, &d1, 1, (hipStream_t)stream, (accumulate & 1) != 0, &done,
                   accumulate & (UIC_TN_FORCE_128 | UIC_TN_FORCE_256 | UIC_TN_SPLITK(0xff))));
  UIC_REQUIRE(done, "linear_wgrad: shape M=%d N=%d K=%d not eligible (needs M >= 128, M %% 8 == 0, N %% 128 == 0, K %% 64 == 0, "
                    "16-byte aligned rows) or workspace of %zu bytes too small", M, N, K, workspace_bytes);
  return UIC_OK;
}

int uic_lstm_cell_fwd(int32_t dtype, int32_t M, int32_t H, int32_t nx, const void* const* x, const int32_t* Kx,
                      const void* const* Wx, const int32_t* ldw, const void* h, const void* W_hh,
                      const float* b_ih, const float* b_hh, const float* c_prev, float* c_out, void* h_out,
                      void* gates_out, void* stream) {
  UIC_REQUIRE(nx >= 0 && nx <= 3, "lstm_cell_fwd: nx=%d outside [0,3]", nx);
  UicGemmParams g;
  memset(&g, 0, sizeof(g));
  g.dtype = dtype; g.M = M; g.N = 4 * H; g.lstm = 1; g.H = H;
  for (int i = 0; i < nx; ++i) {
    UicGemmSeg& s = g.seg[g.nseg++];
    s.A = x[i]; s.B = Wx[i]; s.K = Kx[i]; s.lda = Kx[i]; s.ldb = ldw[i];
  }
  if (h) {
    UicGemmSeg& s = g.seg[g.nseg++];
    s.A = h; s.B = W_hh; s.K = H; s.lda = H; s.ldb = H;
  }
  g.bias = b_ih; g.bias2 = b_hh; g.c_prev = c_prev; g.c_out = c_out; g.h_out = h_out; g.ldh = H; g.gates_out = gates_out;
  return uic_gemm_launch(g, (hipStream_t)stream);
}

int uic_lstm_cell_bwd(int32_t dtype, int32_t M, int32_t H, const float* dh, float* dc, const void* gates,
                      const float* c_prev, const float* c, void* dgates, void* stream) {
  UicLstmBwdParams p;
  memset(&p, 0, sizeof(p));
  p.dtype = dtype; p.M = M; p.H = H; p.dh0 = dh; p.lddh0 = H; p.dc = dc; p.gates = gates; p.c_prev = c_prev; p.c = c;
  p.dgates = dgates;
  return uic_lstm_bwd_launch(p, (hipStream_t)stream);
}

int uic_attention_fwd(int32_t dtype, int32_t N, int32_t R, int32_t A, int32_t H, const float* att_h, const void* p_att,
                      const void* att, const float* w_alpha, const float* b_alpha, const float* mask, float* alpha,
                      void* ctx, void* stream) {
  UicAttnParams a;
  memset(&a, 0, sizeof(a));
  a.dtype = dtype; a.N = N; a.R = R; a.A = A; a.H = H; a.att_h = att_h; a.p_att = p_att; a.att = att;
  a.w_alpha = w_alpha; a.b_alpha = b_alpha; a.mask = mask; a.ldmask = R; a.alpha = alpha; a.ctx = ctx; a.ldctx = H;
  return uic_attention_fwd_launch(a, (hipStream_t)stream);
}
int uic_attention_bwd_step(int32_t dtype, int32_t N, int32_t R, int32_t A, int32_t H, const float* att_h,
                           const void* p_att, const void* att, const float* w_alpha, const float* alpha,
                           const float* dctx, float* de, void* d_att_h, void* stream) {
  UicAttnParams a;
  memset(&a, 0, sizeof(a));
  a.dtype = dtype; a.N = N; a.R = R; a.A = A; a.H = H; a.att_h = att_h; a.p_att = p_att; a.att = att;
  a.w_alpha = w_alpha; a.alpha = (float*)alpha; a.dctx = dctx; a.lddctx = H; a.de = de; a.d_att_h = d_att_h;
  return uic_attention_bwd_step_launch(a, (hipStream_t)stream);
}
int uic_attention_bwd_accum(int32_t dtype, int32_t N, int32_t R, int32_t A, int32_t H, int32_t T,
                            const float* att_h_all, const float* alpha_all, const float* de_all, const float* dctx_all,
                            const void* p_att, const float* w_alpha, float* d_att, void* d_p_att, float* d_walpha_part,
                            void* stream) {
  UicAttnAccumParams a;
  memset(&a, 0, sizeof(a));
  a.dtype = dtype; a.N = N; a.R = R; a.A = A; a.H = H; a.T = T;
  a.att_h_all = att_h_all; a.alpha_all = alpha_all; a.de_all = de_all;
  a.dctx_all = dctx_all; a.lddctx = H; a.dctx_step_stride = (size_t)N * H;
  a.p_att = p_att; a.w_alpha = w_alpha; a.d_att = d_att; a.d_p_att = d_p_att; a.d_walpha_part = d_walpha_part;
  return uic_attention_bwd_accum_launch(a, (hipStream_t)stream);
}

int uic_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                  float eps, int32_t step, float grad_scale, void* stream) {
  UIC_REQUIRE(p && g && m && v, "adam_step: null pointer");
  UIC_REQUIRE(step >= 1, "adam_step: step=%d must be >= 1", step);
  UicAdamParams a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  a.grad_scale = grad_scale;
  a.max_norm = 0.f; a.sqnorm = nullptr; a.guard = nullptr;
  return uic_adam_launch(a, (hipStream_t)stream);
}

int uic_adam_step_guarded(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                          float eps, int32_t step, float grad_scale, const int32_t* skip_if_nonzero, void* stream) {
  UIC_REQUIRE(p && g && m && v, "adam_step_guarded: null pointer");
  UIC_REQUIRE(step >= 1, "adam_step_guarded: step=%d must be >= 1", step);
  UicAdamParams a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  a.grad_scale = grad_scale;
  a.max_norm = 0.f; a.sqnorm = nullptr; a.guard = skip_if_nonzero;
  return uic_adam_launch(a, (hipStream_t)stream);
}

int uic_grad_sqnorm(const float* g, size_t n, float* scratch, float* out, void* stream) {
  return uic_sqnorm_launch(g, n, scratch, out, (hipStream_t)stream);
}

int uic_adam_step_clip(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                       float eps, int32_t step, float grad_scale, float max_norm, const float* sqnorm, void* stream) {
  UIC_REQUIRE(p && g && m && v && sqnorm, "adam_step_clip: null pointer");
  UIC_REQUIRE(step >= 1 && max_norm > 0.f, "adam_step_clip: step=%d must be >= 1 and max_norm=%f > 0", step, (double)max_norm);
  UicAdamParams a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  a.grad_scale = grad_scale;
  a.max_norm = max_norm; a.sqnorm = sqnorm; a.guard = nullptr;
  return uic_adam_launch(a, (hipStream_t)stream);
}

int uic_adam_step_clip_guarded(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                               float eps, int32_t step, float grad_scale, float max_norm, const float* sqnorm,
                               const int32_t* skip_if_nonzero, void* stream) {
  UIC_REQUIRE(p && g && m && v && sqnorm, "adam_step_clip_guarded: null pointer");
  UIC_REQUIRE(step >= 1 && max_norm > 0.f, "adam_step_clip_guarded: step=%d must be >= 1 and max_norm=%f > 0", step, (double)max_norm);
  UicAdamParams a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  a.grad_scale = grad_scale;
  a.max_norm = max_norm; a.sqnorm = sqnorm; a.guard = skip_if_nonzero;
  return uic_adam_launch(a, (hipStream_t)stream);
}

int uic_adam_step_ranges(float* p, const float* g, float* m, float* v, int32_t n_ranges, const uint64_t* lo, const uint64_t* hi,
                         float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale, float max_norm,
                         const float* sqnorm, const int32_t* skip_if_nonzero, void* w_out, int32_t w_dtype, void* stream) {
  UIC_REQUIRE(p && g && m && v && (n_ranges == 0 || (lo && hi)), "adam_step_ranges: null pointer");
  UIC_REQUIRE(step >= 1, "adam_step_ranges: step=%d must be >= 1", step);
  UIC_REQUIRE(n_ranges >= 0 && n_ranges <= UIC_ADAM_RANGES, "adam_step_ranges: %d ranges (max %d)", n_ranges, UIC_ADAM_RANGES);
  UIC_REQUIRE(!w_out || w_dtype == UIC_F32 || w_dtype == UIC_BF16, "adam_step_ranges: bad w_dtype %d", w_dtype);
  UIC_REQUIRE((max_norm > 0.f) == (sqnorm != nullptr), "adam_step_ranges: max_norm and sqnorm go together");
  UicAdamParams a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = 0; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  a.grad_scale = grad_scale;
  a.max_norm = max_norm; a.sqnorm = sqnorm; a.guard = skip_if_nonzero;
  UicAdamRanges r;
  memset(&r, 0, sizeof(r));
  for (int i = 0; i < n_ranges; ++i) {
    UIC_REQUIRE(hi[i] >= lo[i], "adam_step_ranges: range %d is [%llu, %llu)", i, (unsigned long long)lo[i], (unsigned long long)hi[i]);
    if (hi[i] == lo[i]) continue;
    r.lo[r.count] = (size_t)lo[i]; r.start[r.count] = r.total; r.total += (size_t)(hi[i] - lo[i]); ++r.count;
  }
  return uic_adam_ranges_launch(a, r, w_out, w_dtype, (hipStream_t)stream);
}

int uic_lm_criterion(int32_t N, int32_t T, int32_t V1, const float* logp, const int64_t* target, int32_t ld_target,
                     const float* mask, int32_t ld_mask, float* loss_out, float* scratch, float* dlogp, float grad_out,
                     void* stream) {
  UIC_REQUIRE(logp && target && mask && loss_out && scratch, "lm_criterion: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int n = N * T;
  if (n == 0) return UIC_OK;
  float* row_loss = scratch;
  float* row_mask = scratch + n;
  float* fin = scratch + 2 * (size_t)n;   // {loss, den}
  hipLaunchKernelGGL(lm_crit_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, N, T, V1, logp, target, ld_target, mask, ld_mask, row_loss, row_mask);
  UIC_LAUNCH_CHECK("lm_crit_rows");
  hipLaunchKernelGGL(lm_crit_final_kernel, dim3(1), dim3(256), 0, s, row_loss, row_mask, n, fin);
  UIC_LAUNCH_CHECK("lm_crit_final");
  UIC_TRY(uic_copy_launch(loss_out, fin, 4, s));
  if (dlogp) {
    UIC_TRY(uic_fill_launch(dlogp, 0, (size_t)n * V1 * 4, s));
    hipLaunchKernelGGL(lm_crit_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, s, N, T, V1, target, ld_target, mask, ld_mask, fin, grad_out, dlogp);
    UIC_LAUNCH_CHECK("lm_crit_bwd");
  }
  return UIC_OK;
}

int uic_reward_criterion(int32_t N, int32_t L, const float* logp, const int64_t* seq, const float* reward, float* loss_out,
                         float* dlogp, void* stream) {
  UIC_REQUIRE(logp && seq && reward && loss_out, "reward_criterion: null pointer");
  UIC_REQUIRE(N > 0 && L > 0, "reward_criterion: empty input");
  hipLaunchKernelGGL(reward_crit_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, N, L, logp, seq, reward, loss_out, dlogp);
  UIC_LAUNCH_CHECK("reward_crit_kernel");
  return UIC_OK;
}

int uic_cast_from_f32(int32_t dtype, const float* src, void* dst, size_t n, void* stream) {
  return uic_cast_f32_launch(dtype, src, dst, n, (hipStream_t)stream);
}
int uic_cast_to_f32(int32_t dtype, const void* src, float* dst, size_t n, void* stream) {
  return uic_to_f32_launch(dtype, src, dst, n, (hipStream_t)stream);
}
int uic_transpose(int32_t dtype, const void* src, int32_t rows, int32_t cols, int32_t ld_src, void* dst, int32_t ld_dst, void* stream) {
  return uic_transpose_launch(dtype, src, rows, cols, ld_src, dst, ld_dst, (hipStream_t)stream);
}
int uic_dropout_mask(float* out, size_t n, float p, uint32_t seed, uint32_t site, size_t base, void* stream) {
  return uic_dropout_mask_launch(out, n, p, seed, site, base, (hipStream_t)stream);
}

}  // extern "C"
