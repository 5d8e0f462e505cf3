// BatchNorm1d inside att_embed (P/models/AttModel.py:78-84, --use_bn 1|2; opts.py:52 makes 1 the default), applied by
// pack_wrapper (:44-53) to the PACKED live regions only: row (n, r) takes part iff r < row_len[n].
//
// use_bn >= 1: BatchNorm1d(att_feat_size) on the raw region features, in front of the Linear.  The affine part is
//   folded into the Linear (W' = W diag(gamma), b' = b + W beta), so the only extra HBM pass over the 188 MB feature
//   tensor is the statistics pass; the normalisation rides on the operand cast the GEMM needs anyway.  Backward needs
//   no extra GEMM: with dW' = d_pre^T xhat and db = colsum(d_pre),
//       dW = dW' diag(gamma) + db beta^T,   dgamma_c = sum_h W[h,c] dW'[h,c],   dbeta_c = sum_h W[h,c] db[h].
// use_bn == 2: a second BatchNorm1d(rnn_size) after the Dropout; padded regions stay exactly zero.
// All reductions are two-stage in a fixed order (deterministic); batch statistics are merged with Chan's formula.
#include "uic_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ bool live_row(int row, int R, const int* row_len) {
  if (!row_len) return true;
  const int n = row / R;
  return row - n * R < row_len[n];
}
__device__ __forceinline__ void load4(const float* p, float* v) {
  const float4 q = *(const float4*)p;
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float* v) {
  const uint2 q = *(const uint2*)p;
  v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
  v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store4(bf16_t* p, const float* v) {
  *(uint2*)p = make_uint2(uic_pack_bf16x2(v[0], v[1]), uic_pack_bf16x2(v[2], v[3]));
}

// part[(chunk*3 + {0,1,2}) * C + c] = {count, mean, M2} of column c over the live rows of the chunk
template <typename TI>
__global__ __launch_bounds__(NT) void bn_stats_part_kernel(const TI* __restrict__ x, int NR, int R, int C, const int* row_len,
                                                           int rows_per_chunk, float* __restrict__ part) {
  const int c = (blockIdx.x * NT + threadIdx.x) * 4;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(NR, r0 + rows_per_chunk);
  float K[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  for (int row = r0; row < r1; ++row) {
    if (!live_row(row, R, row_len)) continue;
    float v[4];
    load4(x + (size_t)row * C + c, v);
    if (cnt == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) K[j] = v[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dlt = v[j] - K[j];
      s1[j] += dlt;
      s2[j] += dlt * dlt;
    }
    ++cnt;
  }
  float* p = part + (size_t)blockIdx.y * 3 * C + c;
  const float inv = cnt ? 1.f / (float)cnt : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[j] = (float)cnt;
    p[C + j] = K[j] + s1[j] * inv;
    p[2 * C + j] = s2[j] - s1[j] * s1[j] * inv;
  }
}

// stat[c] = batch mean, stat[C + c] = 1/sqrt(biased var + eps); running stats as nn.BatchNorm1d (momentum, unbiased var).
// 64 columns per workgroup; its 4 waves each Chan-merge a quarter of the chunks, then the 4 partials are merged in a fixed
// order (deterministic).
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& M2, float nb, float mb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb, dlt = mb - mean;
  mean += dlt * (nb / nt);
  M2 += m2b + dlt * dlt * (n * nb / nt);
  n = nt;
}
// `rep`: every row stands for `rep` identical rows (features given once per image, seq_per_img caption rows each): mean and
// biased variance are unchanged, the unbiased variance of the running statistics becomes rep*M2 / (rep*n - 1).
__global__ __launch_bounds__(NT) void bn_stats_final_kernel(const float* __restrict__ part, int nchunks, int C, float momentum, float eps,
                                                            float* __restrict__ stat, float* run_mean, float* run_var, float rep) {
  __shared__ float s_n[4][64], s_m[4][64], s_q[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float n = 0.f, mean = 0.f, M2 = 0.f;
  if (c < C)
    for (int b = wave; b < nchunks; b += 4)
      chan_merge(n, mean, M2, part[(size_t)b * 3 * C + c], part[((size_t)b * 3 + 1) * C + c], part[((size_t)b * 3 + 2) * C + c]);
  s_n[wave][lane] = n; s_m[wave][lane] = mean; s_q[wave][lane] = M2;
  __syncthreads();
  if (wave != 0 || c >= C) return;
  for (int wv = 1; wv < 4; ++wv) chan_merge(n, mean, M2, s_n[wv][lane], s_m[wv][lane], s_q[wv][lane]);
  const float var = n > 0.f ? M2 / n : 0.f;
  stat[c] = mean;
  stat[C + c] = rsqrtf(var + eps);
  if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
  if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (n * rep > 1.f ? M2 * rep / (n * rep - 1.f) : var);
}

__global__ void bn_stats_running_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int C, float eps,
                                        float* __restrict__ stat) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  stat[c] = rm[c];
  stat[C + c] = rsqrtf(rv[c] + eps);
}

// out = gamma * (x - mean) * rstd + beta (gamma / beta optional); padded rows -> 0 when zero_padded
template <typename TI, typename TO>
__global__ void bn_apply_kernel(const TI* __restrict__ x, size_t NR, int R, int C, const int* row_len,
                                const float* __restrict__ stat, const float* gamma, const float* beta, int zero_padded,
                                TO* __restrict__ out) {
  const int c4 = C / 4;
  const size_t total = NR * c4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / c4;
    const int c = (int)(i - row * c4) * 4;
    float v[4], o[4];
    if (zero_padded && !live_row((int)row, R, row_len)) {
      o[0] = o[1] = o[2] = o[3] = 0.f;
    } else {
      load4(x + row * C + c, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[j] - stat[c + j]) * stat[C + c + j];
        o[j] = (gamma ? gamma[c + j] : 1.f) * xh + (beta ? beta[c + j] : 0.f);
      }
    }
    store4(out + row * C + c, o);
  }
}

// part[(chunk*3 + {0,1,2}) * C + c] = {count, sum d, sum d * yhat} over the live rows of the chunk
template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_part_kernel(const float* __restrict__ d, const T* __restrict__ y, int NR, int R, int C,
                                                         const int* row_len, const float* __restrict__ stat,
                                                         int rows_per_chunk, float* __restrict__ part) {
  const int c = (blockIdx.x * NT + threadIdx.x) * 4;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(NR, r0 + rows_per_chunk);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  for (int row = r0; row < r1; ++row) {
    if (!live_row(row, R, row_len)) continue;
    float v[4], g[4];
    load4(y + (size_t)row * C + c, v);
    load4(d + (size_t)row * C + c, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1[j] += g[j];
      s2[j] += g[j] * (v[j] - stat[c + j]) * stat[C + c + j];
    }
    ++cnt;
  }
  float* p = part + (size_t)blockIdx.y * 3 * C + c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[j] = (float)cnt;
    p[C + j] = s1[j];
    p[2 * C + j] = s2[j];
  }
}
// red[c] = n, red[C + c] = dbeta, red[2C + c] = dgamma   (64 columns per workgroup, 4 waves split the chunks)
__global__ __launch_bounds__(NT) void bn_bwd_final_kernel(const float* __restrict__ part, int nchunks, int C, float* __restrict__ red,
                                                          float* dgamma, float* dbeta) {
  __shared__ float s_n[4][64], s_1[4][64], s_2[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float n = 0.f, s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int b = wave; b < nchunks; b += 4) {
      n += part[(size_t)b * 3 * C + c];
      s1 += part[((size_t)b * 3 + 1) * C + c];
      s2 += part[((size_t)b * 3 + 2) * C + c];
    }
  s_n[wave][lane] = n; s_1[wave][lane] = s1; s_2[wave][lane] = s2;
  __syncthreads();
  if (wave != 0 || c >= C) return;
  n = (s_n[0][lane] + s_n[1][lane]) + (s_n[2][lane] + s_n[3][lane]);
  s1 = (s_1[0][lane] + s_1[1][lane]) + (s_1[2][lane] + s_1[3][lane]);
  s2 = (s_2[0][lane] + s_2[1][lane]) + (s_2[2][lane] + s_2[3][lane]);
  red[c] = n; red[C + c] = s1; red[2 * C + c] = s2;
  if (dbeta) dbeta[c] = s1;
  if (dgamma) dgamma[c] = s2;
}
// d <- gamma rstd (d - [training] (dbeta + yhat dgamma) / n) on live rows, 0 on padded rows (in place, f32)
template <typename T>
__global__ void bn_bwd_apply_kernel(float* __restrict__ d, const T* __restrict__ y, size_t NR, int R, int C, const int* row_len,
                                    const float* __restrict__ stat, const float* __restrict__ gamma,
                                    const float* __restrict__ red, int training) {
  const int c4 = C / 4;
  const size_t total = NR * c4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / c4;
    const int c = (int)(i - row * c4) * 4;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (live_row((int)row, R, row_len)) {
      float v[4], g[4];
      load4(y + row * C + c, v);
      load4(d + row * C + c, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float rstd = stat[C + c + j];
        float t = g[j];
        if (training) {
          const float inv_n = 1.f / red[c + j];
          t -= (red[C + c + j] + (v[j] - stat[c + j]) * rstd * red[2 * C + c + j]) * inv_n;
        }
        o[j] = gamma[c + j] * rstd * t;
      }
    }
    store4(d + row * C + c, o);
  }
}

// W'[h, c] = W[h, c] gamma[c]  (operand dtype);  b'[h] = b[h] + sum_c W[h, c] beta[c]   -- one block per h
template <typename T>
__global__ __launch_bounds__(NT) void bn_fold_weight_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ b,
                                                            int D, T* __restrict__ Weff, float* __restrict__ beff) {
  __shared__ float s_buf[NT / 64];
  const int h = blockIdx.x;
  float acc = 0.f;
  for (int c = threadIdx.x; c < D; c += NT) {
    const float w = W[(size_t)h * D + c];
    Weff[(size_t)h * D + c] = uic_from_f<T>(w * gamma[c]);
    acc += w * beta[c];
  }
  acc = uic_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < NT / 64; ++i) t += s_buf[i];
    beff[h] = b[h] + t;
  }
}

// in: dW = dW' [H, D], db [H]; out: dgamma, dbeta [D] and dW <- dW' diag(gamma) + db beta^T.  64 columns per workgroup, the 4
// waves split the H rows (fixed-order combination -> deterministic).
__global__ __launch_bounds__(NT) void bn_fold_grad_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ dW,
                                                          const float* __restrict__ db, int H, int D, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  __shared__ float s_g[4][64], s_b[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float ag = 0.f, ab = 0.f;
  const float g = c < D ? gamma[c] : 0.f, bt = c < D ? beta[c] : 0.f;
  if (c < D)
    for (int h = wave; h < H; h += 4) {
      const float w = W[(size_t)h * D + c];
      const float dw = dW[(size_t)h * D + c];
      const float dbh = db[h];
      ag += w * dw;
      ab += w * dbh;
      dW[(size_t)h * D + c] = dw * g + dbh * bt;
    }
  s_g[wave][lane] = ag; s_b[wave][lane] = ab;
  __syncthreads();
  if (wave == 0 && c < D) {
    dgamma[c] = (s_g[0][lane] + s_g[1][lane]) + (s_g[2][lane] + s_g[3][lane]);
    dbeta[c] = (s_b[0][lane] + s_b[1][lane]) + (s_b[2][lane] + s_b[3][lane]);
  }
}

inline int chunks_for(int NR, int* rows_per_chunk) {
  int rpc = 64;
  int n = (NR + rpc - 1) / rpc;
  while (n > 512) { rpc *= 2; n = (NR + rpc - 1) / rpc; }
  *rows_per_chunk = rpc;
  return n;
}
inline int gridn(size_t n) { size_t g = (n + NT - 1) / NT; return (int)(g > 65536 ? 65536 : (g ? g : 1)); }

}  // namespace

size_t uic_bn_scratch_floats(int NR, int C) {
  int rpc;
  return (size_t)chunks_for(NR, &rpc) * 3 * C;
}

int uic_bn_stats_launch(int in_dtype, const void* x, int NR, int R, int C, const int* row_len, float* part, float momentum,
                        float eps, float* stat, float* run_mean, float* run_var, hipStream_t s, float rep) {
  UIC_REQUIRE(x && part && stat && C % 4 == 0 && rep >= 1.f, "bn_stats: bad arguments (C=%d)", C);
  int rpc;
  const int nch = chunks_for(NR, &rpc);
  const dim3 grid((C / 4 + NT - 1) / NT, nch);
  if (in_dtype == UIC_BF16)
    hipLaunchKernelGGL(bn_stats_part_kernel<bf16_t>, grid, dim3(NT), 0, s, (const bf16_t*)x, NR, R, C, row_len, rpc, part);
  else
    hipLaunchKernelGGL(bn_stats_part_kernel<float>, grid, dim3(NT), 0, s, (const float*)x, NR, R, C, row_len, rpc, part);
  UIC_LAUNCH_CHECK("bn_stats_part");
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 63) / 64), dim3(NT), 0, s, part, nch, C, momentum, eps, stat, run_mean, run_var, rep);
  UIC_LAUNCH_CHECK("bn_stats_final");
  return UIC_OK;
}

int uic_bn_stats_running_launch(const float* run_mean, const float* run_var, int C, float eps, float* stat, hipStream_t s) {
  UIC_REQUIRE(run_mean && run_var && stat, "bn_stats_running: null pointer");
  hipLaunchKernelGGL(bn_stats_running_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, s, run_mean, run_var, C, eps, stat);
  UIC_LAUNCH_CHECK("bn_stats_running");
  return UIC_OK;
}

int uic_bn_apply_launch(int in_dtype, int out_dtype, const void* x, int NR, int R, int C, const int* row_len, const float* stat,
                        const float* gamma, const float* beta, int zero_padded, void* out, hipStream_t s) {
  UIC_REQUIRE(x && stat && out && C % 4 == 0, "bn_apply: bad arguments");
  const int g = gridn((size_t)NR * (C / 4));
#define BN_APPLY(TI, TO) hipLaunchKernelGGL((bn_apply_kernel<TI, TO>), dim3(g), dim3(NT), 0, s, (const TI*)x, (size_t)NR, R, C, row_len, stat, gamma, beta, zero_padded, (TO*)out)
  if (in_dtype == UIC_BF16) { if (out_dtype == UIC_BF16) BN_APPLY(bf16_t, bf16_t); else BN_APPLY(bf16_t, float); }
  else { if (out_dtype == UIC_BF16) BN_APPLY(float, bf16_t); else BN_APPLY(float, float); }
#undef BN_APPLY
  UIC_LAUNCH_CHECK("bn_apply");
  return UIC_OK;
}

int uic_bn_bwd_launch(int dtype, float* d, const void* y, int NR, int R, int C, const int* row_len, const float* stat,
                      const float* gamma, int training, float* part, float* red, float* dgamma, float* dbeta, hipStream_t s) {
  UIC_REQUIRE(d && y && stat && gamma && part && red && C % 4 == 0, "bn_bwd: bad arguments");
  int rpc;
  const int nch = chunks_for(NR, &rpc);
  const dim3 grid((C / 4 + NT - 1) / NT, nch);
  if (dtype == UIC_BF16)
    hipLaunchKernelGGL(bn_bwd_part_kernel<bf16_t>, grid, dim3(NT), 0, s, (const float*)d, (const bf16_t*)y, NR, R, C, row_len, stat, rpc, part);
  else
    hipLaunchKernelGGL(bn_bwd_part_kernel<float>, grid, dim3(NT), 0, s, (const float*)d, (const float*)y, NR, R, C, row_len, stat, rpc, part);
  UIC_LAUNCH_CHECK("bn_bwd_part");
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(NT), 0, s, part, nch, C, red, dgamma, dbeta);
  UIC_LAUNCH_CHECK("bn_bwd_final");
  const int g = gridn((size_t)NR * (C / 4));
  if (dtype == UIC_BF16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, d, (const bf16_t*)y, (size_t)NR, R, C, row_len, stat, gamma, red, training);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(g), dim3(NT), 0, s, d, (const float*)y, (size_t)NR, R, C, row_len, stat, gamma, red, training);
  UIC_LAUNCH_CHECK("bn_bwd_apply");
  return UIC_OK;
}

int uic_bn_fold_weight_launch(int dtype, const float* W, const float* gamma, const float* beta, const float* b, int H, int D,
                              void* Weff, float* beff, hipStream_t s) {
  UIC_REQUIRE(W && gamma && beta && b && Weff && beff, "bn_fold_weight: null pointer");
  if (dtype == UIC_BF16)
    hipLaunchKernelGGL(bn_fold_weight_kernel<bf16_t>, dim3(H), dim3(NT), 0, s, W, gamma, beta, b, D, (bf16_t*)Weff, beff);
  else
    hipLaunchKernelGGL(bn_fold_weight_kernel<float>, dim3(H), dim3(NT), 0, s, W, gamma, beta, b, D, (float*)Weff, beff);
  UIC_LAUNCH_CHECK("bn_fold_weight");
  return UIC_OK;
}

int uic_bn_fold_grad_launch(const float* W, const float* gamma, const float* beta, float* dW, const float* db, int H, int D,
                            float* dgamma, float* dbeta, hipStream_t s) {
  UIC_REQUIRE(W && gamma && beta && dW && db && dgamma && dbeta, "bn_fold_grad: null pointer");
  hipLaunchKernelGGL(bn_fold_grad_kernel, dim3((D + 63) / 64), dim3(NT), 0, s, W, gamma, beta, dW, db, H, D, dgamma, dbeta);
  UIC_LAUNCH_CHECK("bn_fold_grad");
  return UIC_OK;
}
