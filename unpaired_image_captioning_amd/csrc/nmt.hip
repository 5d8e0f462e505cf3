// Pivot NMT step (zh -> en teacher) on one MI355X: SURVEY.md section 8a rows 12-15.
//
// Replaces NMTModel.forward (P/models/NMT_Models.py:414-420) = Embeddings + packed bi-LSTM Encoder (:27-135),
// _fix_enc_hidden / init_decoder_state (:284-295), input-feed Decoder over StackedLSTM + dot GlobalAttention
// (:183-271, O/modules/StackedRNN.py:20-34, O/modules/GlobalAttention.py:112-167), the generator +
// NMTCriterion with the NMT_loss.score counters (P/misc/criterion.py:126-136,175-184), and their backward.
//
// Layout: time-major like the reference ([S,B,*], [T,B,*]).  pack_padded_sequence semantics without packing: the
// batch is length-sorted (descending), so the rows alive at source step s are the prefix [0, nb(s)) and every
// recurrent GEMM simply runs on that many rows; state buffers carry one zero slot before and after the S steps, so
// "previous state" of both directions is a plain view and padded positions stay exactly zero (unpack's padding).
// Input projections (W_ih x) are batched over all steps; only W_hh (and the input-feed part) stays in the loops.
// Like the reference, attention is NOT masked over padded source positions (GlobalAttention.mask is never set in
// training): their context vectors are zero, their scores are 0, and they take part in the softmax.
#include "uic_common.h"
#include <mutex>
#include "uic_host.h"
#include "../../include/uic_hip.h"
#include <string.h>

#define SITE_NMT_ENC(l) (1000u + (unsigned)(l))
#define SITE_NMT_DEC(l, t) (2000u + (unsigned)(l) * 256u + (unsigned)(t))
#define SITE_NMT_OUT(t) (4000u + (unsigned)(t))

namespace {

constexpr int NT = 256;

// tgt [T,B] -> target_bt [B,T-1] (= tgt[1:]), mask_bt [B,T-1] (target != PAD), tgt_in [(T-1)*B] (= tgt[:-1], flat)
__global__ void nmt_prep_kernel(const int64_t* tgt, int T, int B, int64_t* target_bt, float* mask_bt, int64_t* tgt_in) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (T - 1) * B) return;
  const int t = i / B, b = i - t * B;
  const long y = tgt[(size_t)(t + 1) * B + b];
  target_bt[(size_t)b * (T - 1) + t] = y;
  mask_bt[(size_t)b * (T - 1) + t] = y != 0 ? 1.f : 0.f;
  tgt_in[i] = tgt[i];
}

template <typename T>
__global__ void dropout_apply_kernel(const void* src_, void* dst_, size_t n, float p, unsigned seed, unsigned site) {
  const T* src = (const T*)src_;
  T* dst = (T*)dst_;
  const float inv_keep = 1.f / (1.f - p);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = uic_from_f<T>(uic_to_f(src[i]) * uic_drop_scale(seed, site, (unsigned)i, p, inv_keep));
}
// g[i] *= mask (f32 gradient through a dropout site)
__global__ void dropout_grad_kernel(float* g, size_t n, float p, unsigned seed, unsigned site) {
  const float inv_keep = 1.f / (1.f - p);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    g[i] *= uic_drop_scale(seed, site, (unsigned)i, p, inv_keep);
}

// _fix_enc_hidden (NMT_Models.py:284-287): decoder initial state of layer l, row b = [fwd final | bwd final]; the
// forward direction ends at the row's last valid step (slot len), the backward one at step 0 (slot 1).
template <typename T>
__global__ void enc_final_kernel(const void* xl_, const float* c_f, const float* c_b, const int* lens, int B, int H, int Hd,
                                 void* h0_, float* c0) {
  const T* xl = (const T*)xl_;
  T* h0 = (T*)h0_;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i - b * H;
  const int d = j >= Hd;
  const int slot = d ? 1 : lens[b];
  h0[i] = xl[((size_t)slot * B + b) * H + j];
  c0[i] = (d ? c_b : c_f)[((size_t)slot * B + b) * Hd + (j - d * Hd)];
}

// sum_j row[j] * vec[j] over one wavefront, 16-byte loads (H is a multiple of the vector width)
template <typename T>
__device__ __forceinline__ float row_dot(const T* __restrict__ row, const float* __restrict__ vec, int H, int lane) {
  constexpr int VEC = uic_vec<T>::N;
  float p = 0.f;
  for (int c = lane; c < H / VEC; c += 64) {
    float f[VEC];
    uic_unpack<T>(*(const uint4*)(row + c * VEC), f);
#pragma unroll
    for (int k = 0; k < VEC; ++k) p += f[k] * vec[c * VEC + k];
  }
  return uic_wave_sum(p);
}
// s_red[wave][:] = sum over this wave's source positions s of coef[s] * ctx[s, b, :]
template <typename T>
__device__ __forceinline__ void weighted_rows(const T* __restrict__ ctx, const float* __restrict__ coef, int S, int B, int H, int b,
                                              int lane, int wave, float* __restrict__ s_red) {
  constexpr int VEC = uic_vec<T>::N;
  for (int c = lane; c < H / VEC; c += 64) {
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int s = wave; s < S; s += 4) {
      float f[VEC];
      uic_unpack<T>(*(const uint4*)(ctx + ((size_t)s * B + b) * H + c * VEC), f);
      const float a = coef[s];
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += a * f[k];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) s_red[wave * H + c * VEC + k] = acc[k];
  }
}

// ---- dot GlobalAttention (O/modules/GlobalAttention.py:112-116,152,160-162), one workgroup per batch row
template <typename T>
// Training path: `ctxw` = context x W_in (f32, made once per batch) and `qvec` = the decoder's rnn_output, so that
// score[s] = ctxw[s, b, :] . q -- the same number as context[s, b, :] . linear_in(q) without a linear_in GEMM per step.
__global__ __launch_bounds__(NT) void gattn_fwd_kernel(const void* ctx_, const float* target, int S, int B, int H, float* attn,
                                                       void* cvec_, const int64_t* mask_src = nullptr, const float* ctxw = nullptr,
                                                       const void* qvec_ = nullptr) {
  const T* ctx = (const T*)ctx_;
  T* cvec = (T*)cvec_;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_t = sm;            // [H]
  float* s_a = s_t + H;       // [S]
  float* s_red = s_a + S;     // [4][H]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < H; j += NT) s_t[j] = ctxw ? uic_to_f(((const T*)qvec_)[(size_t)b * H + j]) : target[(size_t)b * H + j];
  __syncthreads();
  for (int s = wave; s < S; s += 4) {
    float p = ctxw ? row_dot<float>(ctxw + ((size_t)s * B + b) * H, s_t, H, lane)
                   : row_dot<T>(ctx + ((size_t)s * B + b) * H, s_t, H, lane);
    // translator only: source padding is masked (GlobalAttention.applyMask, NMT_Models.py:345,352)
    if (mask_src && mask_src[(size_t)s * B + b] == 0) p = -INFINITY;
    if (lane == 0) s_a[s] = p;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int s = 0; s < S; ++s) mx = fmaxf(mx, s_a[s]);
  float sum = 0.f;
  for (int s = 0; s < S; ++s) sum += expf(s_a[s] - mx);
  const float inv = 1.f / sum;
  __syncthreads();
  for (int s = tid; s < S; s += NT) {
    const float a = expf(s_a[s] - mx) * inv;
    s_a[s] = a;
    attn[(size_t)b * S + s] = a;
  }
  __syncthreads();
  weighted_rows<T>(ctx, s_a, S, B, H, b, lane, wave, s_red);
  __syncthreads();
  for (int j = tid; j < H; j += NT)
    cvec[(size_t)b * H + j] = uic_from_f<T>(s_red[j] + s_red[H + j] + s_red[2 * H + j] + s_red[3 * H + j]);
}

// backward of one decode step: d_a = d_c . ctx[s]; d_score = a (d_a - sum a d_a); with the hoisted linear_in the
// query's gradient is direct: d q += sum_s d_score[s] ctxw[s, b, :] (added to `dq`, which already holds the linear_out share)
template <typename T>
__global__ __launch_bounds__(NT) void gattn_bwd_step_kernel(const void* ctx_, const float* attn, const float* dcq, int lddcq,
                                                            int S, int B, int H, float* dscore, const float* ctxw, float* dq, int lddq) {
  const T* ctx = (const T*)ctx_;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_dc = sm;           // [H]
  float* s_a = s_dc + H;      // [S]
  float* s_da = s_a + S;      // [S]
  float* s_red = s_da + S;    // [4][H]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < H; j += NT) s_dc[j] = dcq[(size_t)b * lddcq + j];
  for (int s = tid; s < S; s += NT) s_a[s] = attn[(size_t)b * S + s];
  __syncthreads();
  for (int s = wave; s < S; s += 4) {
    const float p = row_dot<T>(ctx + ((size_t)s * B + b) * H, s_dc, H, lane);
    if (lane == 0) s_da[s] = p;
  }
  __syncthreads();
  float wbar = 0.f;
  for (int s = 0; s < S; ++s) wbar += s_a[s] * s_da[s];
  __syncthreads();
  for (int s = tid; s < S; s += NT) {
    const float ds = s_a[s] * (s_da[s] - wbar);
    s_da[s] = ds;
    dscore[(size_t)b * S + s] = ds;
  }
  __syncthreads();
  weighted_rows<float>(ctxw, s_da, S, B, H, b, lane, wave, s_red);
  __syncthreads();
  for (int j = tid; j < H; j += NT)
    dq[(size_t)b * lddq + j] += s_red[j] + s_red[H + j] + s_red[2 * H + j] + s_red[3 * H + j];
}

// The same two kernels for bf16 rows of H = 512 and S <= 4 * MAXR source positions, with EVERY global load of the launch issued
// at its top: a wave keeps its <= MAXR context rows (16 B per lane) and, where scores or gradients go through ctxw, those f32
// rows (2 x 16 B per lane) in registers.  The generic kernels above pay a memory latency per phase (query, score rows,
// weighted rows); at batch 64 the launch IS those latencies.  Same arithmetic in the same order: bit-identical results.
template <int MAXR>
__global__ __launch_bounds__(NT) void gattn_fwd_fast_kernel(const bf16_t* __restrict__ ctx, const float* __restrict__ target, int S, int B,
                                                            float* __restrict__ attn, bf16_t* __restrict__ cvec,
                                                            const int64_t* __restrict__ mask_src, const float* __restrict__ ctxw,
                                                            const bf16_t* __restrict__ qvec) {
  constexpr int H = 512;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_t = sm;            // [H]
  float* s_a = s_t + H;       // [S]
  float* s_red = s_a + S;     // [4][H]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint4 cr[MAXR];
  float4 wr[MAXR][2];
  long mk[MAXR];
#pragma unroll
  for (int u = 0; u < MAXR; ++u) {
    const int sp = wave + 4 * u;
    const size_t r = ((size_t)(sp < S ? sp : S - 1) * B + b) * H;
    cr[u] = *(const uint4*)(ctx + r + lane * 8);
    if (ctxw) { wr[u][0] = *(const float4*)(ctxw + r + lane * 4); wr[u][1] = *(const float4*)(ctxw + r + (lane + 64) * 4); }
    mk[u] = mask_src ? mask_src[(size_t)(sp < S ? sp : S - 1) * B + b] : 1;
  }
  for (int j = tid; j < H; j += NT) s_t[j] = ctxw ? uic_to_f(qvec[(size_t)b * H + j]) : target[(size_t)b * H + j];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < MAXR; ++u) {
    const int sp = wave + 4 * u;
    if (sp >= S) break;
    float p = 0.f;
    if (ctxw) {                                           // row_dot<float>: chunks lane, lane + 64 of four floats
      const float* v0 = s_t + lane * 4; const float* v1 = s_t + (lane + 64) * 4;
      p += wr[u][0].x * v0[0]; p += wr[u][0].y * v0[1]; p += wr[u][0].z * v0[2]; p += wr[u][0].w * v0[3];
      p += wr[u][1].x * v1[0]; p += wr[u][1].y * v1[1]; p += wr[u][1].z * v1[2]; p += wr[u][1].w * v1[3];
    } else {                                              // row_dot<bf16>: chunk lane of eight
      float f[8];
      uic_unpack<bf16_t>(cr[u], f);
#pragma unroll
      for (int k = 0; k < 8; ++k) p += f[k] * s_t[lane * 8 + k];
    }
    p = uic_wave_sum(p);
    if (mk[u] == 0) p = -INFINITY;
    if (lane == 0) s_a[sp] = p;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int sp = 0; sp < S; ++sp) mx = fmaxf(mx, s_a[sp]);
  float sum = 0.f;
  for (int sp = 0; sp < S; ++sp) sum += expf(s_a[sp] - mx);
  const float inv = 1.f / sum;
  __syncthreads();
  for (int sp = tid; sp < S; sp += NT) {
    const float a = expf(s_a[sp] - mx) * inv;
    s_a[sp] = a;
    attn[(size_t)b * S + sp] = a;
  }
  __syncthreads();
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
  for (int u = 0; u < MAXR; ++u) {
    const int sp = wave + 4 * u;
    if (sp >= S) break;
    float f[8];
    uic_unpack<bf16_t>(cr[u], f);
    const float a = s_a[sp];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += a * f[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) s_red[wave * H + lane * 8 + k] = acc[k];
  __syncthreads();
  for (int j = tid; j < H; j += NT) cvec[(size_t)b * H + j] = (bf16_t)(s_red[j] + s_red[H + j] + s_red[2 * H + j] + s_red[3 * H + j]);
}

template <int MAXR>
__global__ __launch_bounds__(NT) void gattn_bwd_step_fast_kernel(const bf16_t* __restrict__ ctx, const float* __restrict__ attn,
                                                                 const float* __restrict__ dcq, int lddcq, int S, int B,
                                                                 float* __restrict__ dscore, const float* __restrict__ ctxw,
                                                                 float* __restrict__ dq, int lddq) {
  constexpr int H = 512;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_dc = sm;           // [H]
  float* s_a = s_dc + H;      // [S]
  float* s_da = s_a + S;      // [S]
  float* s_red = s_da + S;    // [4][H]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint4 cr[MAXR];
  float4 wr[MAXR][2];
#pragma unroll
  for (int u = 0; u < MAXR; ++u) {
    const int sp = wave + 4 * u;
    const size_t r = ((size_t)(sp < S ? sp : S - 1) * B + b) * H;
    cr[u] = *(const uint4*)(ctx + r + lane * 8);
    wr[u][0] = *(const float4*)(ctxw + r + lane * 4);
    wr[u][1] = *(const float4*)(ctxw + r + (lane + 64) * 4);
  }
  const float dq0 = dq[(size_t)b * lddq + tid], dq1 = dq[(size_t)b * lddq + tid + NT];      // (H = 2 NT)
  for (int j = tid; j < H; j += NT) s_dc[j] = dcq[(size_t)b * lddcq + j];
  for (int sp = tid; sp < S; sp += NT) s_a[sp] = attn[(size_t)b * S + sp];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < MAXR; ++u) {
    const int sp = wave + 4 * u;
    if (sp >= S) break;
    float f[8];
    uic_unpack<bf16_t>(cr[u], f);
    float p = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) p += f[k] * s_dc[lane * 8 + k];
    p = uic_wave_sum(p);
    if (lane == 0) s_da[sp] = p;
  }
  __syncthreads();
  float wbar = 0.f;
  for (int sp = 0; sp < S; ++sp) wbar += s_a[sp] * s_da[sp];
  __syncthreads();
  for (int sp = tid; sp < S; sp += NT) {
    const float ds = s_a[sp] * (s_da[sp] - wbar);
    s_da[sp] = ds;
    dscore[(size_t)b * S + sp] = ds;
  }
  __syncthreads();
  // weighted_rows<float>(ctxw, s_da): chunks lane, lane + 64 of four floats
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < MAXR; ++u) {
    const int sp = wave + 4 * u;
    if (sp >= S) break;
    const float a = s_da[sp];
    a0[0] += a * wr[u][0].x; a0[1] += a * wr[u][0].y; a0[2] += a * wr[u][0].z; a0[3] += a * wr[u][0].w;
    a1[0] += a * wr[u][1].x; a1[1] += a * wr[u][1].y; a1[2] += a * wr[u][1].z; a1[3] += a * wr[u][1].w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_red[wave * H + lane * 4 + k] = a0[k]; s_red[wave * H + (lane + 64) * 4 + k] = a1[k]; }
  __syncthreads();
  dq[(size_t)b * lddq + tid] = dq0 + (s_red[tid] + s_red[H + tid] + s_red[2 * H + tid] + s_red[3 * H + tid]);
  dq[(size_t)b * lddq + tid + NT] = dq1 + (s_red[tid + NT] + s_red[H + tid + NT] + s_red[2 * H + tid + NT] + s_red[3 * H + tid + NT]);
}

// deferred over decode steps: d ctx[s,b,:] = sum_t a_t[b,s] d_c_t[b,:]  and  d ctxw[s,b,:] = sum_t d_score_t[b,s] q_t[b,:]
template <typename T>
__global__ void gattn_bwd_accum_kernel(const float* attn_all, const float* dscore_all, const float* dcq_all, int lddcq,
                                       const void* q_all_, int Td, int S, int B, int H, float* dctx, void* dctxw_) {
  const T* q_all = (const T*)q_all_;
  T* dctxw = (T*)dctxw_;
  const size_t total = (size_t)S * B * H;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int j = (int)(i % H);
    const size_t sb = i / H;
    const int b = (int)(sb % B), s = (int)(sb / B);
    float acc = 0.f, accw = 0.f;
    for (int t = 0; t < Td; ++t) {
      const size_t tb = (size_t)t * B + b;
      acc += attn_all[tb * S + s] * dcq_all[tb * lddcq + j];
      accw += dscore_all[tb * S + s] * uic_to_f(q_all[tb * H + j]);
    }
    dctx[i] = acc;
    dctxw[i] = uic_from_f<T>(accw);
  }
}

// d_pre = (d_out + d_feed) * dropout_mask * (1 - out_pre^2): backward of out = dropout(tanh(.))
template <typename T>
__global__ void tanh_drop_bwd_kernel(const float* d_out, const float* d_feed, int ld_feed, int H, const void* out_pre_, int n, float p,
                                     unsigned seed, unsigned site, void* d_pre_) {
  const T* out_pre = (const T*)out_pre_;
  T* d_pre = (T*)d_pre_;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = d_out[i] + (d_feed ? d_feed[(size_t)(i / H) * ld_feed + (i % H)] : 0.f);
  if (p > 0.f) g *= uic_drop_scale(seed, site, (unsigned)i, p, inv_keep);
  const float o = uic_to_f(out_pre[i]);
  d_pre[i] = uic_from_f<T>(g * (1.f - o * o));
}

// (NMT_loss.score's counters, criterion.py:175-179, are computed by the criterion kernel: UicXeParams.score_stats)
__global__ void fill_f32_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

inline int gridn(size_t n) { size_t g = (n + NT - 1) / NT; return (int)(g > 65536 ? 65536 : (g ? g : 1)); }

// ---- translator bookkeeping (NMTModel.translateBatch + O/Beam.py); rows are sentence-major: row = b * K + k
// dst[s, b * K + k] = src[s, b]
__global__ void nmt_replicate_src_kernel(const int64_t* __restrict__ src, int S, int B, int K, int64_t* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * B * K) return;
  const int s = i / (B * K), r = i - s * B * K;
  dst[i] = src[(size_t)s * B + r / K];
}
// Beam.__init__: every beam starts as [BOS, PAD, PAD, ...] with zero scores (O/Beam.py:28-38)
__global__ void nmt_beam_init_kernel(int B, int K, int64_t* tok, float* scores, int* done, int* flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * K) { tok[i] = (i % K) == 0 ? 2 : 0; scores[i] = 0.f; }
  if (i < B) done[i] = 0;
  if (i < 4) flags[i] = 0;
}
// `if not active: break` (translateBatch :377-378) decided ON THE DEVICE, so the host loop need not wait for every step:
// before step `step` advances, flags[1] still holds the number of active sentences after step - 1.  None left: the search is
// frozen (flags[2]) -- the remaining, speculatively enqueued steps leave every beam tensor alone -- else flags[3] = the number
// of steps that really ran (the reference's loop count) and the counter is cleared for this step.
__global__ void nmt_beam_begin_kernel(int step, int* flags) {
  if (flags[2]) return;
  if (step > 0 && flags[1] == 0) { flags[2] = 1; return; }
  flags[3] = step + 1;
  flags[1] = 0;
}
// Beam.advance for every sentence (O/Beam.py:52-89): top K of the flattened beam x word scores -- the candidates are the
// per-row top K (cand_val / cand_idx from beam_topk), enumerated beam-major so that ties resolve to the lower flat index --
// back-pointers, next tokens; a sentence is done once its TOP hypothesis ends in EOS; flags[0] = all sentences done.
__global__ __launch_bounds__(64) void nmt_beam_advance_kernel(int B, int K, int step, const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                        float* __restrict__ scores, int* __restrict__ prev_ks, int64_t* __restrict__ next_ys, int64_t* __restrict__ tok,
                                        int* __restrict__ done, int* __restrict__ flags) {
  // one wavefront per sentence: the rows x K candidates sit 4 to a lane, K rounds of a wave-wide arg-max
  constexpr int PER = (UIC_BEAM_MAX * UIC_BEAM_MAX + 63) / 64;
  const int b = blockIdx.x, lane = threadIdx.x;
  if (flags[2]) return;                                   // frozen: every sentence was done before this step (nmt_beam_begin_kernel)
  const int rows = step == 0 ? 1 : K, n = rows * K;
  float pj[PER];
  long flat[PER];
  bool used[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int id = lane + q * 64;
    used[q] = id >= n;
    pj[q] = 0.f;
    flat[q] = 0;
    if (id < n) {
      const int k = id / K;
      const size_t cr = (size_t)b * K * K + id;          // = ((b * K + k) * K + c)
      pj[q] = (step == 0 ? 0.f : scores[(size_t)b * K + k]) + cand_val[cr];
      flat[q] = (long)k * 0x40000000L + cand_idx[cr];
    }
  }
  __syncthreads();                                        // every lane has read the old scores before lane 0 overwrites them
  int first_word = -1;
  for (int j = 0; j < K; ++j) {
    float bp = 0.f;
    long bflat = 0x7fffffffffffffffL;
    int bid = -1;
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (!used[q] && (bid < 0 || pj[q] > bp || (pj[q] == bp && flat[q] < bflat))) { bp = pj[q]; bflat = flat[q]; bid = lane + q * 64; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float op = __shfl_xor(bp, o, 64);
      const long of = __shfl_xor(bflat, o, 64);
      const int oi = __shfl_xor(bid, o, 64);
      if (oi >= 0 && (bid < 0 || op > bp || (op == bp && of < bflat))) { bp = op; bflat = of; bid = oi; }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (bid == lane + q * 64) used[q] = true;
    const int k = bid / K;
    const int word = (int)(bflat - (long)k * 0x40000000L);
    if (j == 0) first_word = word;
    if (lane == 0) {
      prev_ks[((size_t)step * B + b) * K + j] = k;
      next_ys[((size_t)step * B + b) * K + j] = word;
      tok[(size_t)b * K + j] = word;
      scores[(size_t)b * K + j] = bp;
    }
  }
  if (lane == 0) {
    if (first_word == 3) done[b] = 1;
    if (!done[b]) atomicAdd(&flags[1], 1);       // flags[1]: active sentences of this step (zeroed by the host loop's memset)
  }
}
// read-out (translateBatch :384-392, Beam.getHyp :93-117): best final score (first maximum), walk the back-pointers; the
// attention of hypothesis position j is the one of the PARENT beam at step j, PAD source columns dropped (packed left)
__global__ void nmt_beam_readout_kernel(int B, int K, int S, int n_iter, int ld_out, const float* __restrict__ scores, const int* __restrict__ prev_ks,
                                        const int64_t* __restrict__ next_ys, const float* __restrict__ attn_hist, const int64_t* __restrict__ src,
                                        int64_t* __restrict__ hyp, float* __restrict__ score_out, float* __restrict__ attn_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int k = 0;
  for (int j = 1; j < K; ++j)
    if (scores[(size_t)b * K + j] > scores[(size_t)b * K + k]) k = j;
  score_out[b] = scores[(size_t)b * K + k];
  for (int j = n_iter - 1; j >= 0; --j) {
    hyp[(size_t)b * ld_out + j] = next_ys[((size_t)j * B + b) * K + k];
    const int parent = prev_ks[((size_t)j * B + b) * K + k];
    if (attn_out) {
      const float* a = attn_hist + ((size_t)j * B * K + (size_t)b * K + parent) * S;
      float* o = attn_out + ((size_t)b * ld_out + j) * S;
      int n = 0;
      for (int s2 = 0; s2 < S; ++s2)
        if (src[(size_t)s2 * B + b] != 0) o[n++] = a[s2];
      for (; n < S; ++n) o[n] = 0.f;
    }
    k = parent;
  }
}

// the attention launches: register-resident fast kernels for bf16, H = 512, S <= 64 (NT = 256 threads = 4 waves), else generic
#define NMT_GATTN_FWD(GRID, LDS, CTX, TARGET, S_, B_, H_, ATTN, CVEC, MASK, CTXW, Q)                                              \
  do {                                                                                                                           \
    if (dt == UIC_BF16 && (H_) == 512 && (S_) <= 32)                                                                             \
      hipLaunchKernelGGL(gattn_fwd_fast_kernel<8>, dim3(GRID), dim3(NT), LDS, s, (const bf16_t*)(CTX), TARGET, S_, B_, ATTN,     \
                         (bf16_t*)(CVEC), MASK, CTXW, (const bf16_t*)(Q));                                                       \
    else if (dt == UIC_BF16 && (H_) == 512 && (S_) <= 64)                                                                        \
      hipLaunchKernelGGL(gattn_fwd_fast_kernel<16>, dim3(GRID), dim3(NT), LDS, s, (const bf16_t*)(CTX), TARGET, S_, B_, ATTN,    \
                         (bf16_t*)(CVEC), MASK, CTXW, (const bf16_t*)(Q));                                                       \
    else if (dt == UIC_BF16)                                                                                                     \
      hipLaunchKernelGGL(gattn_fwd_kernel<bf16_t>, dim3(GRID), dim3(NT), LDS, s, CTX, TARGET, S_, B_, H_, ATTN, CVEC, MASK, CTXW, Q); \
    else                                                                                                                         \
      hipLaunchKernelGGL(gattn_fwd_kernel<float>, dim3(GRID), dim3(NT), LDS, s, CTX, TARGET, S_, B_, H_, ATTN, CVEC, MASK, CTXW, Q);  \
    UIC_LAUNCH_CHECK("gattn_fwd_kernel");                                                                                        \
  } while (0)
#define NMT_T(KERNEL, GRID, LDS, ...)                                                            \
  do {                                                                                           \
    if (dt == UIC_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, dim3(GRID), dim3(NT), LDS, s, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<float>, dim3(GRID), dim3(NT), LDS, s, __VA_ARGS__);           \
    UIC_LAUNCH_CHECK(#KERNEL);                                                                   \
  } while (0)

constexpr int ML = UIC_NMT_MAX_LAYERS;

struct NmtLayout {
  // operand-dtype weight views (masters in f32 mode, copies in bf16 mode) and transposes for the dX GEMMs
  const void* enc_lin_w; const void* enc_w_ih[ML][2]; const void* enc_w_hh[ML][2];
  const void* dec_w_ih[ML]; const void* dec_w_hh[ML]; const void* attn_in_w; const void* attn_out_w; const void* gen_w;
  void* c_enc_lin_w; void* c_enc_w_ih[ML][2]; void* c_enc_w_hh[ML][2]; void* c_dec_w_ih[ML]; void* c_dec_w_hh[ML];
  void* c_attn_in_w; void* c_attn_out_w; void* c_gen_w;
  void* enc_lin_wT;            // [W, W]
  void* enc_w_ihT[ML][2];      // [in, 4Hd]
  void* enc_w_hhT[ML][2];      // [Hd, 4Hd]
  void* dec_wT[ML];            // [in_l + H, 4H] = [W_ih^T ; W_hh^T]
  void* attn_in_wT;            // [H, H]
  void* attn_out_wT;           // [2H, H]
  void* gen_wT;                // [H, Vtp]
  // encoder: xl[l] = input of layer l with one zero slot before and after the S steps: [(S+2), B, in_l]
  void* xe; void* xl[ML + 1]; void* xd[ML];
  float* gx_e[ML][2]; float* c_e[ML][2]; void* gates_e[ML][2]; void* dg_e[ML][2];
  // decoder
  int64_t* target_bt; float* mask_bt; int64_t* tgt_in;
  void* emb_d; float* gx_d0;
  void* hd[ML]; float* cd[ML]; void* hdrop[ML]; void* gates_d[ML]; void* dg_d[ML]; float* dhrec_d[ML]; float* dcd[ML];
  float* ctxw; void* dctxw; float* attn_all; void* cvec_all; void* out_pre; void* out_all;   // ctxw = context x W_in [S*B, H] f32
  float* logits; void* dlogits; float* row_loss; float* scalars; int* stats;
  void* out_live; float* d_out_live; int* live_map;
  // backward
  float* d_out_all; float* dfeed; void* d_pre_all; float* d_cq_all; float* dscore_all; float* dq;
  // dx_lstm[l]: d[x_l | h_l(t-1)] of decoder layer l > 0, one buffer per layer so that nothing has to be copied
  float* dx_lstm[ML]; float* d_lay; float* dhrec_e; float* dc_e; float* dhrec_e1; float* dc_e1; float* dx_e; void* dpre_e; float* dxe; float* demb_d;
  void* tA; void* tB; float* colscratch; size_t colscratch_floats; float* slab; size_t slab_bytes;
  unsigned* rnn_sync;          // nmt_persist.hip's registration / barrier counters
  int* embed_scratch;          // uic_embed_bwd_sorted_launch (both embedding tables, one after the other)
  float* dfeed_x; float* dq_att_x;   // [Td, B, H] f32: step-indexed exchange slabs of the persistent BPTT launch
  unsigned long long* dec_bwd_dbg;   // [256][Td][16] time stamps of the persistent BPTT launch (UIC_REC_STAMPS)
  unsigned long long* dec_fwd_dbg;   // the same of the persistent forward launch
  size_t total;
};

NmtLayout nmt_layout(const uic_nmt_dims& d, const uic_nmt_weights* w, void* ws) {
  NmtLayout L;
  memset(&L, 0, sizeof(L));
  Bump b{(char*)ws, 0};
  const size_t Sz = uic_dtype_size(d.dtype);
  const size_t B = d.B, S = d.S, Td = d.T - 1, H = d.H, W = d.W, Hd = H / 2, NL = d.layers, Vt = d.Vt, Vtp = vpad(Vt);
  const bool bf = d.dtype == UIC_BF16;
  auto wcopy = [&](const float* master, size_t n, void** copy) -> const void* {
    *copy = b.take(n * Sz);
    return (bf || !w) ? *copy : (const void*)master;
  };
  L.enc_lin_w = wcopy(w ? w->enc_lin_w : nullptr, W * W, &L.c_enc_lin_w);
  for (size_t l = 0; l < NL; ++l) {
    const size_t in = l == 0 ? W : H, din = l == 0 ? W + H : H;
    for (int dd = 0; dd < 2; ++dd) {
      L.enc_w_ih[l][dd] = wcopy(w ? w->enc_w_ih[l][dd] : nullptr, 4 * Hd * in, &L.c_enc_w_ih[l][dd]);
      L.enc_w_hh[l][dd] = wcopy(w ? w->enc_w_hh[l][dd] : nullptr, 4 * Hd * Hd, &L.c_enc_w_hh[l][dd]);
      L.enc_w_ihT[l][dd] = b.take(in * 4 * Hd * Sz);
      L.enc_w_hhT[l][dd] = b.take(Hd * 4 * Hd * Sz);
    }
    L.dec_w_ih[l] = wcopy(w ? w->dec_w_ih[l] : nullptr, 4 * H * din, &L.c_dec_w_ih[l]);
    L.dec_w_hh[l] = wcopy(w ? w->dec_w_hh[l] : nullptr, 4 * H * H, &L.c_dec_w_hh[l]);
    L.dec_wT[l] = b.take((din + H) * 4 * H * Sz);
  }
  L.attn_in_w = wcopy(w ? w->attn_in_w : nullptr, H * H, &L.c_attn_in_w);
  L.attn_out_w = wcopy(w ? w->attn_out_w : nullptr, H * 2 * H, &L.c_attn_out_w);
  L.gen_w = wcopy(w ? w->gen_w : nullptr, Vt * H, &L.c_gen_w);
  L.enc_lin_wT = b.take(W * W * Sz);
  L.attn_in_wT = b.take(H * H * Sz);
  L.attn_out_wT = b.take(2 * H * H * Sz);
  L.gen_wT = b.take(H * Vtp * Sz);
  L.xe = b.take(S * B * W * Sz);
  for (size_t l = 0; l <= NL; ++l) L.xl[l] = b.take((S + 2) * B * (l == 0 ? W : H) * Sz);
  for (size_t l = 1; l < NL; ++l) L.xd[l] = b.take((S + 2) * B * H * Sz);
  for (size_t l = 0; l < NL; ++l)
    for (int dd = 0; dd < 2; ++dd) {
      L.gx_e[l][dd] = (float*)b.take(S * B * 4 * Hd * 4);
      L.c_e[l][dd] = (float*)b.take((S + 2) * B * Hd * 4);
      L.gates_e[l][dd] = b.take(S * B * 4 * Hd * Sz);
      L.dg_e[l][dd] = b.take(S * B * 4 * Hd * Sz);
    }
  L.target_bt = (int64_t*)b.take(B * Td * 8);
  L.mask_bt = (float*)b.take(B * Td * 4);
  L.tgt_in = (int64_t*)b.take(Td * B * 8);
  L.emb_d = b.take(Td * B * W * Sz);
  L.gx_d0 = (float*)b.take(Td * B * 4 * H * 4);
  for (size_t l = 0; l < NL; ++l) {
    L.hd[l] = b.take((Td + 1) * B * H * Sz);
    L.cd[l] = (float*)b.take((Td + 1) * B * H * 4);
    L.hdrop[l] = b.take(Td * B * H * Sz);
    L.gates_d[l] = b.take(Td * B * 4 * H * Sz);
    L.dg_d[l] = b.take(Td * B * 4 * H * Sz);
    L.dhrec_d[l] = (float*)b.take(B * H * 4);
    L.dcd[l] = (float*)b.take(B * H * 4);
  }
  L.ctxw = (float*)b.take(S * B * H * 4);
  L.dctxw = b.take(S * B * H * Sz);
  L.attn_all = (float*)b.take(Td * B * S * 4);
  L.cvec_all = b.take(Td * B * H * Sz);
  L.out_pre = b.take(Td * B * H * Sz);
  // (the generator's weight gradient is one TN GEMM over the T B target rows: they are padded with zero rows to a multiple of 128, the
  // K granularity of the 256 x 256 ping-pong kernel -- 190 -> 135 us at configs[2]'s 1984 rows; backward() clears the padding)
  const size_t Mdp = (Td * B + 127) & ~(size_t)127;
  L.out_all = b.take(((Td + 1) * B + (Mdp - Td * B)) * H * Sz);
  L.logits = (float*)b.take(Mdp * Vtp * 4);            // (Mdp rows: the live-position list is padded to whole 128-row tiles)
  L.dlogits = b.take(Mdp * Vtp * Sz);
  L.out_live = b.take(Mdp * H * Sz);                   // uic_nmt_dims.tgt_live_rows: the listed rows of `out_all`, their d out
  L.d_out_live = (float*)b.take(Mdp * H * 4);
  L.live_map = (int*)b.take(Mdp * 4);                  // the list made on the device (tgt_live_rows == NULL)
  L.row_loss = (float*)b.take(Td * B * 4);
  L.scalars = (float*)b.take(64);
  L.stats = (int*)b.take(64);
  L.d_out_all = (float*)b.take(Td * B * H * 4);
  L.dfeed = (float*)b.take(B * 2 * H * 4);
  L.d_pre_all = b.take(Td * B * H * Sz);
  L.d_cq_all = (float*)b.take(Td * B * 2 * H * 4);
  L.dscore_all = (float*)b.take(Td * B * S * 4);
  L.dq = (float*)b.take(B * H * 4);
  for (int l = 0; l < ML; ++l) L.dx_lstm[l] = (float*)b.take(B * 2 * H * 4);
  L.d_lay = (float*)b.take(S * B * H * 4);
  L.dhrec_e = (float*)b.take(B * Hd * 4);
  L.dc_e = (float*)b.take(B * Hd * 4);
  L.dhrec_e1 = (float*)b.take(B * Hd * 4);
  L.dc_e1 = (float*)b.take(B * Hd * 4);
  L.dx_e = (float*)b.take(S * B * (W > H ? W : H) * 4);
  L.dpre_e = b.take(S * B * W * Sz);
  L.dxe = (float*)b.take(S * B * W * 4);
  L.demb_d = (float*)b.take(Td * B * W * 4);
  const size_t rows = rup8((S > Td ? S : Td) * B) > Mdp ? rup8((S > Td ? S : Td) * B) : Mdp;   // (the generator's padded rows: the transposing fallback of wgrad_group)
  L.tA = b.take((Vt > 4 * H ? Vt : 4 * H) * rows * Sz);
  L.tB = b.take((W + 2 * H) * rows * Sz);
  const size_t maxcols = Vtp > 4 * H ? Vtp : 4 * H;
  L.colscratch_floats = 128 * maxcols;
  L.colscratch = (float*)b.take(L.colscratch_floats * 4);
  L.slab_bytes = 4 * (4 * H) * (W + 2 * H) * 4;
  L.slab = (float*)b.take(L.slab_bytes);
  L.rnn_sync = (unsigned*)b.take((2 * NL + 2) * uic_rnn_persist_sync_bytes());   // one block per persistent launch of a step (sync_block)
  L.dfeed_x = (float*)b.take(Td * B * H * 4);
  L.dq_att_x = (float*)b.take(Td * B * H * 4);
  L.dec_bwd_dbg = (unsigned long long*)b.take((size_t)256 * Td * 16 * 8);
  L.dec_fwd_dbg = (unsigned long long*)b.take((size_t)256 * Td * 16 * 8);
  {
    const size_t es = uic_embed_bwd_sorted_scratch_ints((int)(S * B), 1, d.Vs, (int)W), ed = uic_embed_bwd_sorted_scratch_ints((int)(Td * B), 1, d.Vt, (int)W);
    L.embed_scratch = (int*)b.take((es > ed ? es : ed) * 4);
  }
  L.total = (b.off + 255) & ~(size_t)255;
  return L;
}

int nmt_check(const uic_nmt_dims* d) {
  UIC_REQUIRE(d != nullptr, "null dims");
  UIC_REQUIRE(d->dtype == UIC_F32 || d->dtype == UIC_BF16, "bad dtype %d", d->dtype);
  UIC_REQUIRE(d->B > 0 && d->S > 0 && d->S <= 4096 && d->T >= 2 && d->Vs > 1 && d->Vt > 1, "bad sizes B=%d S=%d T=%d", d->B, d->S, d->T);
  UIC_REQUIRE(d->layers >= 1 && d->layers <= ML, "layers=%d outside [1,%d]", d->layers, ML);
  UIC_REQUIRE(d->H % 16 == 0 && d->W % 8 == 0, "rnn_size=%d must be a multiple of 16, word_vec_size=%d of 8", d->H, d->W);
  UIC_REQUIRE(d->T - 1 < 256, "target length %d too long for the dropout site encoding", d->T);
  UIC_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "dropout=%f outside [0,1)", (double)d->drop_p);
  return UIC_OK;
}

// Second HIP stream for the backward-direction half of every encoder layer (the two directions of a bidirectional
// layer are independent chains of S latency-bound launches each).
struct NmtSide {
  hipStream_t stream = nullptr;
  hipEvent_t ev_go = nullptr, ev_done = nullptr;
  hipEvent_t ev_r0 = nullptr, ev_gen = nullptr;   // refresh: the caller's stream has reached it; the generator's copies are made
  // uic_nmt_grad_ready_wait: the generator's gradients (side stream) / the decoder-side gradients (caller's stream) of the last
  // uic_nmt_backward are final
  hipEvent_t ev_grad_gen = nullptr, ev_grad_dec = nullptr;
  bool grads_recorded = false;
  bool ready = false;
};
NmtSide g_nmt_side[16];
std::mutex g_nmt_side_mutex;
int nmt_side(NmtSide** out) {
  int dev = 0;
  UIC_TRY(uic_check_hip(hipGetDevice(&dev), "hipGetDevice"));
  UIC_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
  NmtSide& ss = g_nmt_side[dev];
  std::lock_guard<std::mutex> lock(g_nmt_side_mutex);
  if (!ss.ready) {
    UIC_TRY(uic_check_hip(hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking), "hipStreamCreateWithFlags"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_go, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_done, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_r0, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_gen, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_grad_gen, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_grad_dec, hipEventDisableTiming), "hipEventCreate"));
    ss.ready = true;
  }
  *out = &ss;
  return UIC_OK;
}

struct Nmt {
  uic_nmt_dims d;
  const uic_nmt_weights* w;
  const uic_nmt_weights* G;
  NmtLayout L;
  int dt, B, S, Td, H, W, Hd, NL, Vs, Vt, Vtp;
  size_t Sz, BH, BHd;
  float drop_p;
  unsigned seed;
  const int64_t* src;
  const int64_t* tgt;
  int nb[4096];
  // uic_nmt_dims.tgt_live_rows: generator, criterion and their gradients over the non-PAD target positions only
  bool live = false;
  int live_n = 0, live_pad = 0;
  const int32_t* live_rows = nullptr;

  int init(const uic_nmt_dims* d_, const uic_nmt_weights* w_, const int64_t* src_, const int32_t* lengths_host,
           const int64_t* tgt_, int training, unsigned seed_, void* ws, const uic_nmt_weights* G_) {
    d = *d_; w = w_; G = G_; src = src_; tgt = tgt_;
    L = nmt_layout(d, w, ws);
    dt = d.dtype; B = d.B; S = d.S; Td = d.T - 1; H = d.H; W = d.W; Hd = H / 2; NL = d.layers; Vs = d.Vs; Vt = d.Vt;
    Vtp = (int)vpad(Vt);
    Sz = uic_dtype_size(dt); BH = (size_t)B * H; BHd = (size_t)B * Hd;
    drop_p = training ? d.drop_p : 0.f;
    seed = seed_;
    live = d.tgt_live_count > 0 && d.tgt_live_count <= Td * B && ((size_t)H * Sz) % 16 == 0;
    live_n = live ? d.tgt_live_count : 0;
    live_pad = (live_n + 127) & ~127;
    live_rows = !live ? nullptr : d.tgt_live_rows ? d.tgt_live_rows : L.live_map;
    for (int b = 0; b < B; ++b) {
      UIC_REQUIRE(lengths_host[b] >= 1 && lengths_host[b] <= S, "lengths[%d]=%d outside [1,%d]", b, lengths_host[b], S);
      UIC_REQUIRE(b == 0 || lengths_host[b] <= lengths_host[b - 1], "lengths must be sorted in decreasing order (pack_padded_sequence)");
    }
    for (int st = 0; st < S; ++st) {
      int n = 0;
      while (n < B && lengths_host[n] > st) ++n;
      nb[st] = n;
    }
    return UIC_OK;
  }

  // the generator's 25.6 M-element operand copy and its transpose (28 + 28 us) are not needed before the decoder is through: they
  // run on the side stream, and whoever reads them first waits for ev_gen (wait_gen)
  bool gen_pending = false;
  int wait_gen(hipStream_t s) {
    if (!gen_pending) return UIC_OK;
    gen_pending = false;
    NmtSide* ss = nullptr;
    UIC_TRY(nmt_side(&ss));
    return uic_check_hip(hipStreamWaitEvent(s, ss->ev_gen, 0), "hipStreamWaitEvent(generator copies)");
  }
  int refresh(hipStream_t s) {
    // operand-dtype copies and transposes of the weights, a handful of multi-tensor launches instead of one per tensor
    // (the step is a chain of small launches: 34 of them were this)
    {
      NmtSide* ss = nullptr;
      UIC_TRY(nmt_side(&ss));
      UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_r0, s), "hipEventRecord"));            // the master weights are final
      UIC_TRY(uic_check_hip(hipStreamWaitEvent(ss->stream, ss->ev_r0, 0), "hipStreamWaitEvent"));
      if (dt == UIC_BF16) UIC_TRY(uic_cast_f32_launch(dt, w->gen_w, L.c_gen_w, (size_t)Vt * H, ss->stream));   // (25.6 M elements: a launch of its own)
      UIC_TRY(uic_transpose_launch(dt, L.gen_w, Vt, H, H, L.gen_wT, Vtp, ss->stream));
      UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_gen, ss->stream), "hipEventRecord"));
      gen_pending = true;
    }
    if (dt == UIC_BF16) {
      const float* src[UIC_CAST_MULTI]; void* dst[UIC_CAST_MULTI]; size_t n[UIC_CAST_MULTI];
      int k = 0;
      auto flush = [&]() -> int { if (k) UIC_TRY(uic_cast_f32_multi_launch(dt, k, src, dst, n, s)); k = 0; return UIC_OK; };
      auto add = [&](const float* a, const void* b, size_t cnt) -> int {
        src[k] = a; dst[k] = (void*)b; n[k] = cnt;
        if (++k == UIC_CAST_MULTI) return flush();
        return UIC_OK;
      };
      UIC_TRY(add(w->enc_lin_w, L.c_enc_lin_w, (size_t)W * W));
      for (int l = 0; l < NL; ++l) {
        const int in = l == 0 ? W : H, din = l == 0 ? W + H : H;
        for (int dd = 0; dd < 2; ++dd) {
          UIC_TRY(add(w->enc_w_ih[l][dd], L.c_enc_w_ih[l][dd], (size_t)4 * Hd * in));
          UIC_TRY(add(w->enc_w_hh[l][dd], L.c_enc_w_hh[l][dd], (size_t)4 * Hd * Hd));
        }
        UIC_TRY(add(w->dec_w_ih[l], L.c_dec_w_ih[l], (size_t)4 * H * din));
        UIC_TRY(add(w->dec_w_hh[l], L.c_dec_w_hh[l], (size_t)4 * H * H));
      }
      UIC_TRY(add(w->attn_in_w, L.c_attn_in_w, (size_t)H * H));
      UIC_TRY(add(w->attn_out_w, L.c_attn_out_w, (size_t)H * 2 * H));
      UIC_TRY(flush());
    }
    {
      UicTransposeJob jobs[UIC_TRANSPOSE_MULTI];
      int k = 0;
      auto flush = [&]() -> int { if (k) UIC_TRY(uic_transpose_multi_launch(dt, k, jobs, s)); k = 0; return UIC_OK; };
      auto add = [&](const void* src, int rows, int cols, int lds, void* dst, int ldd) -> int {
        jobs[k] = UicTransposeJob{src, dst, rows, cols, lds, ldd};
        if (++k == UIC_TRANSPOSE_MULTI) return flush();
        return UIC_OK;
      };
      UIC_TRY(add(L.enc_lin_w, W, W, W, L.enc_lin_wT, W));
      for (int l = 0; l < NL; ++l) {
        const int in = l == 0 ? W : H, din = l == 0 ? W + H : H;
        for (int dd = 0; dd < 2; ++dd) {
          UIC_TRY(add(L.enc_w_ih[l][dd], 4 * Hd, in, in, L.enc_w_ihT[l][dd], 4 * Hd));
          UIC_TRY(add(L.enc_w_hh[l][dd], 4 * Hd, Hd, Hd, L.enc_w_hhT[l][dd], 4 * Hd));
        }
        UIC_TRY(add(L.dec_w_ih[l], 4 * H, din, din, L.dec_wT[l], 4 * H));
        UIC_TRY(add(L.dec_w_hh[l], 4 * H, H, H, offw(L.dec_wT[l], (size_t)din * 4 * H, dt), 4 * H));
      }
      UIC_TRY(add(L.attn_in_w, H, H, H, L.attn_in_wT, H));
      UIC_TRY(add(L.attn_out_w, H, 2 * H, 2 * H, L.attn_out_wT, H));
      UIC_TRY(flush());
    }
    return UIC_OK;
  }

  // input of encoder layer l as seen by its W_ih GEMM (slots 1..S): the dropped copy between layers in training
  const void* enc_in(int l) const {
    const int in = l == 0 ? W : H;
    return off(l > 0 && drop_p > 0.f ? L.xd[l] : L.xl[l], (size_t)B * in, dt);
  }

  // Persistent launch i of a step (encoder layers forward 0..NL-1, decoder forward NL, decoder BPTT NL+1, encoder layers backward
  // NL+2..) has a sync block of its own, so that one launch clears all of them together with the step's other zero-initialised
  // buffers (uic_zero_list_launch) instead of one memset per launch and per buffer.
  unsigned* sync_block(int i) const { return L.rnn_sync + (size_t)i * (uic_rnn_persist_sync_bytes() / 4); }
  int zero_forward_buffers(hipStream_t s) {
    void* ptr[4 * UIC_NMT_MAX_LAYERS + 4];
    size_t nbytes[4 * UIC_NMT_MAX_LAYERS + 4];
    int n = 0;
    auto add = [&](void* q, size_t by) { ptr[n] = q; nbytes[n] = (by + 15) & ~(size_t)15; ++n; };
    for (int l = 0; l < NL; ++l) {
      add(L.xl[l + 1], (size_t)(S + 2) * BH * Sz);
      for (int dd = 0; dd < 2; ++dd) add(L.c_e[l][dd], (size_t)(S + 2) * BHd * 4);
      if (l + 1 < NL && drop_p > 0.f) add(L.xd[l + 1], (size_t)(S + 2) * BH * Sz);
    }
    add(L.out_all, BH * Sz);                                         // init_input_feed: zeros (:454-458)
    add(sync_block(0), (size_t)(NL + 1) * uic_rnn_persist_sync_bytes());
    return uic_zero_list_launch(ptr, nbytes, n, s);
  }
  size_t gen_rows() const { return ((size_t)Td * B + 127) & ~(size_t)127; }
  int zero_backward_buffers(hipStream_t s) {
    void* ptr[2 * UIC_NMT_MAX_LAYERS + 5];
    size_t nbytes[2 * UIC_NMT_MAX_LAYERS + 5];
    int n = 0;
    if (live) { ptr[n] = L.d_out_all; nbytes[n] = (size_t)Td * B * H * 4; ++n; }
    for (int l = 0; l < NL; ++l)
      for (int dd = 0; dd < 2; ++dd) { ptr[n] = L.dg_e[l][dd]; nbytes[n] = ((size_t)S * B * 4 * Hd * Sz + 15) & ~(size_t)15; ++n; }
    ptr[n] = sync_block(NL + 1); nbytes[n] = (size_t)(NL + 1) * uic_rnn_persist_sync_bytes(); ++n;
    const size_t pad = gen_rows() - (size_t)Td * B;      // zero rows behind the generator weight gradient's operands (nmt_layout)
    if (pad) {
      ptr[n] = offw(L.dlogits, (size_t)Td * B * Vtp, dt); nbytes[n] = pad * Vtp * Sz; ++n;
      ptr[n] = offw(L.out_all, (size_t)(Td + 1) * B * H, dt); nbytes[n] = pad * H * Sz; ++n;
    }
    return uic_zero_list_launch(ptr, nbytes, n, s);
  }

  int encoder_fwd(const int32_t* lengths_dev, hipStream_t s) {
    UIC_TRY(zero_forward_buffers(s));
    // relu(linear(word_lut[src]))  (NMT_Models.py:63-67)
    UIC_TRY(uic_embed_fwd_launch(dt, w->enc_lut, Vs, W, src, 1, S * B, 1, 0.f, 0, 0, 0, 0, L.xe, s));
    {
      UicGemmParams g = gemm_base(dt, S * B, W);
      add_seg(g, L.xe, W, L.enc_lin_w, W, W);
      g.C = offw(L.xl[0], (size_t)B * W, dt); g.ldc = W; g.bias = w->enc_lin_b; g.flags = UIC_GEMM_RELU;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    const bool enc_persist = !(d.recurrence & UIC_REC_FWD_CHAIN) && uic_nmt_enc_persist_eligible(dt, B, S, H);
    for (int l = 0; l < NL; ++l) {
      const int in = l == 0 ? W : H;
      if (enc_persist) {
        // both directions' S recurrent steps as ONE persistent launch (nmt_persist.hip) behind the two batched input GEMMs
        UicNmtEncParams p;
        memset(&p, 0, sizeof(p));
        p.B = B; p.S = S;
        for (int st = 0; st < S; ++st) p.nb[st] = nb[st];
        p.x_out = L.xl[l + 1];
        for (int dd = 0; dd < 2; ++dd) {
          UicGemmParams g = gemm_base(dt, S * B, 4 * Hd);
          add_seg(g, enc_in(l), in, L.enc_w_ih[l][dd], in, in);
          g.C = L.gx_e[l][dd]; g.ldc = 4 * Hd; g.bias = w->enc_b_ih[l][dd]; g.bias2 = w->enc_b_hh[l][dd]; g.flags = UIC_GEMM_OUT_F32;
          UIC_TRY(uic_gemm_launch(g, s));
          p.w_hh[dd] = L.enc_w_hh[l][dd]; p.gx[dd] = L.gx_e[l][dd]; p.c[dd] = L.c_e[l][dd]; p.gates[dd] = L.gates_e[l][dd];
        }
        p.sync = sync_block(l); p.sync_zeroed = 1; p.status = d.rnn_status; p.force_safe = (d.recurrence & UIC_REC_SAFE) != 0;
        p.row0 = 0; p.Nrows = B;
        UIC_TRY(uic_nmt_enc_fwd_persist_launch(p, s));
        if (l + 1 < NL && drop_p > 0.f) {
          NMT_T(dropout_apply_kernel, gridn((size_t)S * BH), 0, (const void*)off(L.xl[l + 1], BH, dt), (void*)offw(L.xd[l + 1], BH, dt),
                (size_t)S * BH, drop_p, seed, SITE_NMT_ENC(l));
        }
        continue;
      }
      // the layer's input and the zeroed output buffer are ready: the backward direction runs on the side stream
      NmtSide* ss = nullptr;
      UIC_TRY(nmt_side(&ss));
      UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_go, s), "hipEventRecord"));
      UIC_TRY(uic_check_hip(hipStreamWaitEvent(ss->stream, ss->ev_go, 0), "hipStreamWaitEvent"));
      for (int dd = 1; dd >= 0; --dd) {
        hipStream_t sd = dd == 1 ? ss->stream : s;
        {  // W_ih x + b_ih + b_hh for every (s, b)
          UicGemmParams g = gemm_base(dt, S * B, 4 * Hd);
          add_seg(g, enc_in(l), in, L.enc_w_ih[l][dd], in, in);
          g.C = L.gx_e[l][dd]; g.ldc = 4 * Hd; g.bias = w->enc_b_ih[l][dd]; g.bias2 = w->enc_b_hh[l][dd]; g.flags = UIC_GEMM_OUT_F32;
          UIC_TRY(uic_gemm_launch(g, sd));
        }
        for (int k = 0; k < S; ++k) {
          const int st = dd == 0 ? k : S - 1 - k;          // time step; its slot is st + 1
          const int prev = dd == 0 ? st : st + 2;          // slot of the previous state in this direction
          if (nb[st] == 0) continue;                       // S longer than the longest sentence
          UicGemmParams g = gemm_base(dt, nb[st], 4 * Hd);
          g.lstm = 1; g.H = Hd;
          add_seg(g, off(L.xl[l + 1], (size_t)prev * BH + dd * Hd, dt), H, L.enc_w_hh[l][dd], Hd, Hd);
          g.pre1 = L.gx_e[l][dd] + (size_t)st * B * 4 * Hd; g.ldpre1 = 4 * Hd;
          g.c_prev = L.c_e[l][dd] + (size_t)prev * BHd; g.c_out = L.c_e[l][dd] + (size_t)(st + 1) * BHd;
          g.h_out = offw(L.xl[l + 1], (size_t)(st + 1) * BH + dd * Hd, dt); g.ldh = H;
          g.gates_out = offw(L.gates_e[l][dd], (size_t)st * B * 4 * Hd, dt);
          UIC_TRY(uic_gemm_launch(g, sd));
        }
        if (dd == 1) UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_done, sd), "hipEventRecord"));
      }
      UIC_TRY(uic_check_hip(hipStreamWaitEvent(s, ss->ev_done, 0), "hipStreamWaitEvent"));
      if (l + 1 < NL && drop_p > 0.f) {  // nn.LSTM's inter-layer dropout (element index over the [S*B, H] slots 1..S)
        NMT_T(dropout_apply_kernel, gridn((size_t)S * BH), 0, (const void*)off(L.xl[l + 1], BH, dt), (void*)offw(L.xd[l + 1], BH, dt),
              (size_t)S * BH, drop_p, seed, SITE_NMT_ENC(l));
      }
    }
    // decoder initial state (_fix_enc_hidden + init_decoder_state, :284-295): slot 0 of the decoder state buffers
    for (int l = 0; l < NL; ++l)
      NMT_T(enc_final_kernel, gridn(BH), 0, (const void*)L.xl[l + 1], (const float*)L.c_e[l][0], (const float*)L.c_e[l][1],
            lengths_dev, B, H, Hd, (void*)L.hd[l], L.cd[l]);
    return UIC_OK;
  }

  int decoder_fwd(hipStream_t s) {
    const int H4 = 4 * H;
    hipLaunchKernelGGL(nmt_prep_kernel, dim3(gridn((size_t)Td * B)), dim3(NT), 0, s, tgt, Td + 1, B, L.target_bt, L.mask_bt, L.tgt_in);
    UIC_LAUNCH_CHECK("nmt_prep_kernel");
    UIC_TRY(uic_embed_fwd_launch(dt, w->dec_lut, Vt, W, L.tgt_in, 1, Td * B, 1, 0.f, 0, 0, 0, 0, L.emb_d, s));
    {  // layer-0 input projection of the embedding part for all steps (the input-feed part stays in the loop)
      UicGemmParams g = gemm_base(dt, Td * B, H4);
      add_seg(g, L.emb_d, W, L.dec_w_ih[0], W + H, W);
      g.C = L.gx_d0; g.ldc = H4; g.bias = w->dec_b_ih[0]; g.bias2 = w->dec_b_hh[0]; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    {  // ctxw[s, b, :] = context[s, b, :] W_in, all source positions at once (see gattn_fwd_kernel)
      UicGemmParams g = gemm_base(dt, S * B, H);
      add_seg(g, off(L.xl[NL], BH, dt), H, L.attn_in_wT, H, H);
      g.C = L.ctxw; g.ldc = H; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    if (!(d.recurrence & UIC_REC_FWD_CHAIN) && uic_nmt_dec_persist_eligible(dt, B, S, H, NL)) {
      // the whole target-step loop as ONE persistent launch (nmt_persist.hip): same buffers, same dropout sites
      UicNmtDecParams p;
      memset(&p, 0, sizeof(p));
      p.B = B; p.S = S; p.Td = Td; p.NL = NL;
      p.out_all = L.out_all; p.out_pre = L.out_pre; p.gx_d0 = L.gx_d0;
      for (int l = 0; l < NL; ++l) {
        p.hd[l] = L.hd[l]; p.cd[l] = L.cd[l]; p.hdrop[l] = L.hdrop[l]; p.gates_d[l] = L.gates_d[l];
        p.w_ih[l] = l == 0 ? off(L.dec_w_ih[0], W, dt) : L.dec_w_ih[l]; p.ld_ih[l] = l == 0 ? W + H : H;
        p.w_hh[l] = L.dec_w_hh[l];
        p.b_ih[l] = w->dec_b_ih[l]; p.b_hh[l] = w->dec_b_hh[l];
      }
      p.ctx = off(L.xl[NL], BH, dt); p.ctxw = L.ctxw; p.attn_all = L.attn_all; p.cvec_all = L.cvec_all; p.attn_out_w = L.attn_out_w;
      p.drop_p = drop_p; p.seed = seed;
      p.sync = sync_block(NL); p.sync_zeroed = 1; p.status = d.rnn_status; p.force_safe = (d.recurrence & UIC_REC_SAFE) != 0;
      p.row0 = 0; p.Nrows = B;
      p.dbg = (d.recurrence & UIC_REC_STAMPS) ? L.dec_fwd_dbg : nullptr;
      return uic_nmt_dec_persist_launch(p, s);
    }
    const size_t lds_att = sizeof(float) * ((size_t)H + S + 4 * (size_t)H);
    for (int t = 0; t < Td; ++t) {
      const void* x = nullptr;
      for (int l = 0; l < NL; ++l) {                           // StackedLSTM.forward (StackedRNN.py:20-34)
        UicGemmParams g = gemm_base(dt, B, H4);
        g.lstm = 1; g.H = H;
        if (l == 0) {
          add_seg(g, off(L.out_all, (size_t)t * BH, dt), H, off(L.dec_w_ih[0], W, dt), W + H, H);   // input feed (:248-249)
          g.pre1 = L.gx_d0 + (size_t)t * B * H4; g.ldpre1 = H4;
        } else {
          add_seg(g, x, H, L.dec_w_ih[l], H, H);
          g.bias = w->dec_b_ih[l]; g.bias2 = w->dec_b_hh[l];
        }
        add_seg(g, off(L.hd[l], (size_t)t * BH, dt), H, L.dec_w_hh[l], H, H);
        g.c_prev = L.cd[l] + (size_t)t * BH; g.c_out = L.cd[l] + (size_t)(t + 1) * BH;
        g.h_out = offw(L.hd[l], (size_t)(t + 1) * BH, dt); g.ldh = H;
        g.gates_out = offw(L.gates_d[l], (size_t)t * B * H4, dt);
        if (l + 1 < NL) {                                      // dropout between layers only
          g.h_drop = offw(L.hdrop[l], (size_t)t * BH, dt); g.ldhd = H;
          g.drop_p = drop_p; g.seed = seed; g.site = SITE_NMT_DEC(l, t);
          x = g.h_drop;
        }
        UIC_TRY(uic_gemm_launch(g, s));
      }
      const void* q = off(L.hd[NL - 1], (size_t)(t + 1) * BH, dt);       // rnn_output = top layer's h
      // scores = context . linear_in(rnn_output) (GlobalAttention.py:114-116) = (context W_in) . rnn_output: L.ctxw was
      // made once before the loop, so no linear_in GEMM sits in the per-step chain
      NMT_GATTN_FWD(B, lds_att, (const void*)off(L.xl[NL], BH, dt), (const float*)nullptr, S, B, H,
                    L.attn_all + (size_t)t * B * S, (void*)offw(L.cvec_all, (size_t)t * BH, dt), (const int64_t*)nullptr,
                    (const float*)L.ctxw, q);
      {  // tanh(linear_out([c ; rnn_output])) (:165-167)
        UicGemmParams g = gemm_base(dt, B, H);
        add_seg(g, off(L.cvec_all, (size_t)t * BH, dt), H, L.attn_out_w, 2 * H, H);
        add_seg(g, q, H, off(L.attn_out_w, H, dt), 2 * H, H);
        // output = dropout(attn_output) = next step's input feed (NMT_Models.py:258-259): the epilogue writes both the
        // tanh output (kept for the backward pass) and its dropped copy
        g.C_pre = offw(L.out_pre, (size_t)t * BH, dt); g.ldc_pre = H;
        g.C = offw(L.out_all, (size_t)(t + 1) * BH, dt); g.ldc = H; g.flags = UIC_GEMM_TANH;
        g.drop_p = drop_p; g.seed = seed; g.site = SITE_NMT_OUT(t);
        UIC_TRY(uic_gemm_launch(g, s));
      }
    }
    return UIC_OK;
  }

  // generator + NMTCriterion + NMT_loss.score (criterion.py:126-136,175-184)
  int loss_fwd(float* loss_out, int32_t* stats_out, hipStream_t s) {
    UIC_TRY(wait_gen(s));
    if (live && !d.tgt_live_rows) UIC_TRY(uic_live_list_launch(L.mask_bt, Td, 0, B, Td * B, L.live_map, live_pad, s));   // (mask_bt: nmt_prep_kernel, this stream)
    if (live) UIC_TRY(uic_gather_rows_launch(off(L.out_all, BH, dt), live_rows, Td * B, L.out_live, live_n, live_pad, (size_t)H * Sz, s));
    if (!live || live_pad > 0) {
      UicGemmParams g = gemm_base(dt, live ? live_pad : Td * B, Vt);
      add_seg(g, live ? L.out_live : off(L.out_all, BH, dt), H, L.gen_w, H, H);
      g.C = L.logits; g.ldc = Vtp; g.bias = w->gen_b; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    hipLaunchKernelGGL(fill_f32_kernel, dim3(1), dim3(64), 0, s, L.scalars, 1.0f, 1);      // size_average=False: no denominator
    UIC_LAUNCH_CHECK("fill_f32_kernel");
    UicXeParams x;
    memset(&x, 0, sizeof(x));
    x.dtype = dt; x.M = Td * B; x.V1 = Vt; x.ldv = Vtp; x.logits = L.logits; x.dlogits = L.dlogits; x.N = B;
    x.target = L.target_bt; x.ldtarget = Td; x.target_col0 = 0;
    x.mask = L.mask_bt; x.ldmask = Td; x.mask_col0 = 0;             // weight[PAD] = 0
    x.inv_den = L.scalars; x.row_loss = L.row_loss; x.write_grad = 1;
    if (stats_out) {       // NMT_loss.score's counters ride on the criterion kernel's pass over the logits
      UIC_TRY(uic_fill_launch(L.stats, 0, 8, s));
      x.score_stats = L.stats;
    }
    if (live) { x.M = live_pad; x.row_map = live_rows; x.row_map_limit = Td * B; }   // (rows of logits / d logits / row_loss by list position)
    UIC_TRY(uic_xe_launch(x, s));
    UIC_TRY(uic_reduce_sum_launch(L.row_loss, live ? (size_t)live_n : (size_t)Td * B, 0.f, nullptr, loss_out, s));
    if (stats_out) {
      UIC_TRY(uic_copy_launch(stats_out, L.stats, 8, s));
    }
    return UIC_OK;
  }


  int backward(hipStream_t s) {
    const int H4 = 4 * H, Md = Td * B, Ms = S * B;
    UIC_TRY(zero_backward_buffers(s));
    // ---- generator
    {
      // d out = d logits W_gen: [T B, H] outputs over K = the target vocabulary -- few tiles and a very long reduction: split-K
      // over workgroups with the deterministic slab reduction (wgrad_multi; one launch of 64 x 64 tiles walked all 50 048 columns
      // in 347 us)
      const WDest d1{live ? L.d_out_live : L.d_out_all, (int)H, 0, (int)H};
      // (d out of the positions the list leaves out is zero: zero_backward_buffers cleared the buffer; the split-K reduce places
      // the listed rows itself, a direct GEMM leaves compact rows to scatter)
      WRows wr{live_rows, live_n, Md, L.d_out_all, (int)H, false};
      if (!live || live_pad > 0) UIC_TRY(wgrad_multi(L.slab, L.slab_bytes, dt, L.dlogits, live ? live_pad : Md, L.gen_wT, H, Vtp, &d1, 1, s, false, live ? &wr : nullptr));
      if (live && !wr.used) UIC_TRY(uic_scatter_rows_launch(L.d_out_live, live_rows, L.d_out_all, Md, live_n, (size_t)H * 4, s));
    }
    // the generator's weight / bias gradients (the largest GEMM of the backward pass, [Vt, H] over all target rows) need nothing
    // from the BPTT loop and the loop -- ~6 dependent launches of 64 rows per step -- leaves the chip idle: they run on the side
    // stream beside it and are joined before the main stream next touches the shared scratch buffers
    NmtSide* ssg = nullptr;
    UIC_TRY(nmt_side(&ssg));
    UIC_TRY(uic_check_hip(hipEventRecord(ssg->ev_go, s), "hipEventRecord"));
    UIC_TRY(uic_check_hip(hipStreamWaitEvent(ssg->stream, ssg->ev_go, 0), "hipStreamWaitEvent"));
    if (live && live_pad == 0) {
      // (a batch without a single target word: no gradient)
      UIC_TRY(uic_check_hip(hipMemsetAsync(G->gen_w, 0, (size_t)Vt * H * 4, ssg->stream), "hipMemsetAsync(d generator.weight)"));
      UIC_TRY(uic_check_hip(hipMemsetAsync(G->gen_b, 0, (size_t)Vt * 4, ssg->stream), "hipMemsetAsync(d generator.bias)"));
    } else {
      const UicGemmTnSeg seg{live ? L.out_live : off(L.out_all, BH, dt), H, H};
      const WDest d1{G->gen_w, H, 0, H};
      UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.dlogits, Vtp, Vt, &seg, 1, live ? live_pad : (int)gen_rows(), &d1, 1, ssg->stream, false, L.tA, L.tB));
      UIC_TRY(uic_colsum_launch(dt, L.dlogits, live ? live_pad : Md, Vt, Vtp, G->gen_b, L.colscratch, L.colscratch_floats, ssg->stream));
    }
    UIC_TRY(uic_check_hip(hipEventRecord(ssg->ev_done, ssg->stream), "hipEventRecord"));
    UIC_TRY(uic_check_hip(hipEventRecord(ssg->ev_grad_gen, ssg->stream), "hipEventRecord"));   // gradient group 0: generator.*
    // ---- decoder BPTT
    const bool bptt_persist = !(d.recurrence & UIC_REC_FWD_CHAIN) && uic_nmt_dec_bwd_persist_eligible(dt, B, S, H, NL);
    if (bptt_persist) {
      // ONE persistent launch (nmt_persist.hip) instead of 7 launches per step.  It needs every CU (one 160 KB workgroup each), so
      // the generator's weight gradient cannot run beside it: the main stream waits for the side stream first.
      UIC_TRY(uic_check_hip(hipStreamWaitEvent(s, ssg->ev_done, 0), "hipStreamWaitEvent"));
      UicNmtDecBwdParams p;
      memset(&p, 0, sizeof(p));
      p.B = B; p.S = S; p.Td = Td;
      p.d_out_all = L.d_out_all; p.out_pre = L.out_pre; p.d_pre_all = L.d_pre_all; p.d_cq_all = L.d_cq_all;
      p.attn_all = L.attn_all; p.ctx = off(L.xl[NL], BH, dt); p.ctxw = L.ctxw; p.dscore_all = L.dscore_all;
      for (int l = 0; l < 2; ++l) { p.gates_d[l] = L.gates_d[l]; p.cd[l] = L.cd[l]; p.dg_d[l] = L.dg_d[l]; p.dc_init[l] = L.dcd[l]; }
      p.woutT = L.attn_out_wT; p.w1T = L.dec_wT[1]; p.w0T = off(L.dec_wT[0], (size_t)W * H4, dt);
      p.dfeed_x = L.dfeed_x; p.dq_att_x = L.dq_att_x;
      p.dh_init[0] = L.dfeed; p.dh_init[1] = L.dx_lstm[1];
      p.drop_p = drop_p; p.seed = seed;
      p.sync = sync_block(NL + 1); p.sync_zeroed = 1; p.status = d.rnn_status; p.force_safe = (d.recurrence & UIC_REC_SAFE) != 0;
      p.row0 = 0; p.Nrows = B;
      p.dbg = (d.recurrence & UIC_REC_STAMPS) ? L.dec_bwd_dbg : nullptr;
      UIC_TRY(uic_nmt_dec_bwd_persist_launch(p, s));
    }
    for (int l = 0; l < NL && !bptt_persist; ++l) {
      UIC_TRY(uic_fill_launch(L.dhrec_d[l], 0, BH * 4, s));
      UIC_TRY(uic_fill_launch(L.dcd[l], 0, BH * 4, s));
    }
    const size_t lds_att = sizeof(float) * ((size_t)H + 2 * (size_t)S + 4 * (size_t)H);
    for (int t = Td - 1; t >= 0 && !bptt_persist; --t) {
      const bool last = t == Td - 1;
      void* d_pre = offw(L.d_pre_all, (size_t)t * BH, dt);
      float* d_cq = L.d_cq_all + (size_t)t * B * 2 * H;
      NMT_T(tanh_drop_bwd_kernel, gridn(BH), 0, (const float*)(L.d_out_all + (size_t)t * BH), (const float*)(last ? nullptr : L.dfeed),
            2 * H, H, (const void*)off(L.out_pre, (size_t)t * BH, dt), (int)BH, drop_p, seed, SITE_NMT_OUT(t), d_pre);
      {  // d[c ; q] = d_pre W_out
        UicGemmParams g = gemm_base(dt, B, 2 * H);
        add_seg(g, d_pre, H, L.attn_out_wT, H, H);
        g.C = d_cq; g.ldc = 2 * H; g.flags = UIC_GEMM_OUT_F32;
        UIC_TRY(uic_gemm_launch(g, s));
      }
      // d q = d_cq[:, H:] (linear_out's share) + sum_s d_score[s] ctxw[s]: added by the attention kernel itself
      if (dt == UIC_BF16 && H == 512 && S <= 64) {
        const bf16_t* cx = (const bf16_t*)off(L.xl[NL], BH, dt);
        const float* at = L.attn_all + (size_t)t * B * S;
        float* dsc = L.dscore_all + (size_t)t * B * S;
        if (S <= 32)
          hipLaunchKernelGGL(gattn_bwd_step_fast_kernel<8>, dim3(B), dim3(NT), lds_att, s, cx, at, (const float*)d_cq, 2 * H, S, B, dsc,
                             (const float*)L.ctxw, d_cq + H, 2 * H);
        else
          hipLaunchKernelGGL(gattn_bwd_step_fast_kernel<16>, dim3(B), dim3(NT), lds_att, s, cx, at, (const float*)d_cq, 2 * H, S, B, dsc,
                             (const float*)L.ctxw, d_cq + H, 2 * H);
        UIC_LAUNCH_CHECK("gattn_bwd_step_fast_kernel");
      } else
      NMT_T(gattn_bwd_step_kernel, B, lds_att, (const void*)off(L.xl[NL], BH, dt), (const float*)(L.attn_all + (size_t)t * B * S),
            (const float*)d_cq, 2 * H, S, B, H, L.dscore_all + (size_t)t * B * S, (const float*)L.ctxw, d_cq + H, 2 * H);
      for (int l = NL - 1; l >= 0; --l) {
        UicLstmBwdParams p;
        memset(&p, 0, sizeof(p));
        p.dtype = dt; p.M = B; p.H = H;
        if (l == NL - 1) { p.dh0 = d_cq + H; p.lddh0 = 2 * H; }
        else {                                               // gradient w.r.t. the dropped h of layer l (input of layer l+1)
          p.dh0 = L.dx_lstm[l + 1]; p.lddh0 = 2 * H;
          p.drop_p = drop_p; p.seed = seed; p.site = SITE_NMT_DEC(l, t);
        }
        if (!last) {       // d h_l(t) from step t + 1: layer 0 has its own buffer, layers > 0 read the second half of their dX
          if (l == 0) { p.dh1 = L.dfeed + H; p.lddh1 = 2 * H; }
          else { p.dh1 = L.dx_lstm[l] + H; p.lddh1 = 2 * H; }
        }
        p.dc = L.dcd[l]; p.gates = off(L.gates_d[l], (size_t)t * B * H4, dt);
        p.c_prev = L.cd[l] + (size_t)t * BH; p.c = L.cd[l] + (size_t)(t + 1) * BH;
        p.dgates = offw(L.dg_d[l], (size_t)t * B * H4, dt);
        UIC_TRY(uic_lstm_bwd_launch(p, s));
        if (l > 0) {   // d[x_l | h_l_prev] = dG [W_ih | W_hh]
          UicGemmParams g = gemm_base(dt, B, 2 * H);
          add_seg(g, p.dgates, H4, L.dec_wT[l], H4, H4);
          g.C = L.dx_lstm[l]; g.ldc = 2 * H; g.flags = UIC_GEMM_OUT_F32;
          UIC_TRY(uic_gemm_launch(g, s));
        } else {       // layer 0: d[feed | h_0_prev] = dG [W_ih[:, W:] | W_hh] in ONE GEMM (rows W .. W + 2H of dec_wT[0] are
                       // contiguous; the embedding part is batched below): L.dfeed is [B, 2H] = [d feed | d h_0(t-1)]
          UicGemmParams g = gemm_base(dt, B, 2 * H);
          add_seg(g, p.dgates, H4, off(L.dec_wT[0], (size_t)W * H4, dt), H4, H4);
          g.C = L.dfeed; g.ldc = 2 * H; g.flags = UIC_GEMM_OUT_F32;
          UIC_TRY(uic_gemm_launch(g, s));
        }
      }
    }
    UIC_TRY(uic_check_hip(hipStreamWaitEvent(s, ssg->ev_done, 0), "hipStreamWaitEvent"));   // generator gradients done: scratch is free
    // ---- decoder weights over all steps
    for (int l = 0; l < NL; ++l) {
      if (l == 0) {   // inputs [emb | feed_prev | h_prev]
        const UicGemmTnSeg segs[3] = {{L.emb_d, W, W}, {L.out_all, H, H}, {L.hd[0], H, H}};
        const WDest dd[2] = {{G->dec_w_ih[0], W + H, 0, W + H}, {G->dec_w_hh[0], H, W + H, H}};
        UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.dg_d[0], H4, H4, segs, 3, Md, dd, 2, s, false, L.tA, L.tB));
      } else {        // inputs [dropped h of layer l-1 | h_prev]
        const UicGemmTnSeg segs[2] = {{L.hdrop[l - 1], H, H}, {L.hd[l], H, H}};
        const WDest dd[2] = {{G->dec_w_ih[l], H, 0, H}, {G->dec_w_hh[l], H, H, H}};
        UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.dg_d[l], H4, H4, segs, 2, Md, dd, 2, s, false, L.tA, L.tB));
      }
      UIC_TRY(uic_colsum_launch(dt, L.dg_d[l], Md, H4, H4, G->dec_b_ih[l], L.colscratch, L.colscratch_floats, s));
      UIC_TRY(uic_copy_launch(G->dec_b_hh[l], G->dec_b_ih[l], (size_t)H4 * 4, s));
    }
    {  // decoder embeddings (padding_idx = PAD keeps that row's gradient at zero)
      UicGemmParams g = gemm_base(dt, Md, W);
      add_seg(g, L.dg_d[0], H4, L.dec_wT[0], H4, H4);
      g.C = L.demb_d; g.ldc = W; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
      // (bucketed by word, one owner per table row: bit-reproducible, no floating-point atomics -- csrc/pointwise.hip)
      UIC_TRY(uic_embed_bwd_sorted_launch(dt, L.demb_d, nullptr, L.tgt_in, 1, Md, 1, Vt, W, 0.f, 0, G->dec_lut, L.embed_scratch, s));
    }
    // ---- deferred attention gradients: d context (direct share) and d ctxw, then linear_in through ctxw = context W_in:
    // dW_in = context^T d ctxw,  d context += d ctxw W_in^T
    float* d_top = L.d_lay;
    NMT_T(gattn_bwd_accum_kernel, gridn((size_t)S * BH), 0, (const float*)L.attn_all, (const float*)L.dscore_all, (const float*)L.d_cq_all,
          2 * H, (const void*)off(L.hd[NL - 1], BH, dt), Td, S, B, H, d_top, (void*)L.dctxw);
    {
      const UicGemmTnSeg seg{L.dctxw, H, H};
      const WDest d1{G->attn_in_w, H, 0, H};
      UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, off(L.xl[NL], BH, dt), H, H, &seg, 1, Ms, &d1, 1, s, false, L.tA, L.tB));
      UicGemmParams g = gemm_base(dt, Ms, H);
      add_seg(g, L.dctxw, H, L.attn_in_w, H, H);
      g.C = d_top; g.ldc = H; g.flags = UIC_GEMM_OUT_F32 | UIC_GEMM_ACCUM;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    // linear_out from (d_pre, [c | q])
    {
      const UicGemmTnSeg segs[2] = {{L.cvec_all, H, H}, {off(L.hd[NL - 1], BH, dt), H, H}};
      const WDest d2{G->attn_out_w, 2 * H, 0, 2 * H};
      UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.d_pre_all, H, H, segs, 2, Md, &d2, 1, s, false, L.tA, L.tB));
    }
    // gradient group 1 (uic_nmt_grad_ready_wait): decoder LSTMs, decoder embeddings, linear_in, linear_out are final
    UIC_TRY(uic_check_hip(hipEventRecord(ssg->ev_grad_dec, s), "hipEventRecord"));
    ssg->grads_recorded = true;
    // ---- encoder: d context = deferred attention gradient; d h0/c0 of the decoder enter at each row's final steps
    const bool enc_persist = !(d.recurrence & UIC_REC_FWD_CHAIN) && uic_nmt_enc_persist_eligible(dt, B, S, H);
    for (int l = NL - 1; l >= 0; --l) {
      const int in = l == 0 ? W : H;
      NmtSide* ss = nullptr;
      UIC_TRY(nmt_side(&ss));
      if (enc_persist) {   // the layer's BPTT, both directions, as ONE persistent launch (nmt_persist.hip)
        UicNmtEncParams p;
        memset(&p, 0, sizeof(p));
        p.B = B; p.S = S;
        for (int st = 0; st < S; ++st) p.nb[st] = nb[st];
        for (int dd = 0; dd < 2; ++dd) {
          p.w_hh[dd] = L.enc_w_hhT[l][dd]; p.c[dd] = L.c_e[l][dd]; p.gates[dd] = L.gates_e[l][dd]; p.dgates[dd] = L.dg_e[l][dd];
        }
        p.d_top = d_top;
        p.dh_init = (l == 0 ? L.dfeed : L.dx_lstm[l]) + H; p.ld_dh_init = 2 * H;
        p.dc_init = L.dcd[l]; p.ld_dc_init = H;
        p.sync = sync_block(NL + 2 + l); p.sync_zeroed = 1; p.status = d.rnn_status; p.force_safe = (d.recurrence & UIC_REC_SAFE) != 0;
        p.row0 = 0; p.Nrows = B;
        UIC_TRY(uic_nmt_enc_bwd_persist_launch(p, s));
      }
      UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_go, s), "hipEventRecord"));          // d_top of this layer is final
      UIC_TRY(uic_check_hip(hipStreamWaitEvent(ss->stream, ss->ev_go, 0), "hipStreamWaitEvent"));
      for (int dd = 1; dd >= 0 && !enc_persist; --dd) {                                  // backward direction on the side stream
        hipStream_t sd = dd == 1 ? ss->stream : s;
        float* dhrec = dd == 1 ? L.dhrec_e1 : L.dhrec_e;
        float* dcc = dd == 1 ? L.dc_e1 : L.dc_e;
        // carried dh / dc start from the decoder-initial-state gradient halves (rows join the BPTT when they become active)
        // (d h_l(-1) of decoder layers > 0 sits in the second half of that layer's last dX)
        const float* dh_init = (l == 0 ? L.dfeed : L.dx_lstm[l]) + H;
        const size_t dh_pitch = (size_t)2 * H * 4;
        UIC_TRY(uic_check_hip(hipMemcpy2DAsync(dhrec, (size_t)Hd * 4, dh_init + dd * Hd, dh_pitch, (size_t)Hd * 4, B,
                                               hipMemcpyDeviceToDevice, sd), "memcpy2d dh0"));
        UIC_TRY(uic_check_hip(hipMemcpy2DAsync(dcc, (size_t)Hd * 4, L.dcd[l] + dd * Hd, (size_t)H * 4, (size_t)Hd * 4, B,
                                               hipMemcpyDeviceToDevice, sd), "memcpy2d dc0"));
        for (int k = S - 1; k >= 0; --k) {
          const int st = dd == 0 ? k : S - 1 - k;
          const int prev = dd == 0 ? st : st + 2;
          if (nb[st] == 0) continue;
          UicLstmBwdParams p;
          memset(&p, 0, sizeof(p));
          p.dtype = dt; p.M = nb[st]; p.H = Hd;
          p.dh0 = d_top + (size_t)st * BH + dd * Hd; p.lddh0 = H;
          p.dh1 = dhrec; p.lddh1 = Hd;
          p.dc = dcc; p.gates = off(L.gates_e[l][dd], (size_t)st * B * 4 * Hd, dt);
          p.c_prev = L.c_e[l][dd] + (size_t)prev * BHd; p.c = L.c_e[l][dd] + (size_t)(st + 1) * BHd;
          p.dgates = offw(L.dg_e[l][dd], (size_t)st * B * 4 * Hd, dt);
          UIC_TRY(uic_lstm_bwd_launch(p, sd));
          UicGemmParams g = gemm_base(dt, nb[st], Hd);
          add_seg(g, p.dgates, 4 * Hd, L.enc_w_hhT[l][dd], 4 * Hd, 4 * Hd);
          g.C = dhrec; g.ldc = Hd; g.flags = UIC_GEMM_OUT_F32;
          UIC_TRY(uic_gemm_launch(g, sd));
        }
        if (dd == 1) UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_done, sd), "hipEventRecord"));
      }
      if (!enc_persist) UIC_TRY(uic_check_hip(hipStreamWaitEvent(s, ss->ev_done, 0), "hipStreamWaitEvent"));
      for (int dd = 0; dd < 2; ++dd) {
        // weights of this direction: dG^T [4Hd, S*B] x [x_l | h_prev]^T  (padded rows of dG are zero)
        {
          const UicGemmTnSeg segs[2] = {{enc_in(l), in, in}, {off(L.xl[l + 1], (size_t)(dd == 0 ? 0 : 2) * BH + dd * Hd, dt), H, Hd}};
          const WDest dw[2] = {{G->enc_w_ih[l][dd], in, 0, in}, {G->enc_w_hh[l][dd], Hd, in, Hd}};
          UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.dg_e[l][dd], 4 * Hd, 4 * Hd, segs, 2, Ms, dw, 2, s, false, L.tA, L.tB));
        }
        UIC_TRY(uic_colsum_launch(dt, L.dg_e[l][dd], Ms, 4 * Hd, 4 * Hd, G->enc_b_ih[l][dd], L.colscratch, L.colscratch_floats, s));
        UIC_TRY(uic_copy_launch(G->enc_b_hh[l][dd], G->enc_b_ih[l][dd], (size_t)4 * Hd * 4, s));
      }
      {  // d input of layer l = sum over directions dG W_ih
        UicGemmParams g = gemm_base(dt, Ms, in);
        add_seg(g, L.dg_e[l][0], 4 * Hd, L.enc_w_ihT[l][0], 4 * Hd, 4 * Hd);
        add_seg(g, L.dg_e[l][1], 4 * Hd, L.enc_w_ihT[l][1], 4 * Hd, 4 * Hd);
        g.C = L.dx_e; g.ldc = in; g.flags = UIC_GEMM_OUT_F32;
        UIC_TRY(uic_gemm_launch(g, s));
      }
      if (l > 0) {
        if (drop_p > 0.f) {
          hipLaunchKernelGGL(dropout_grad_kernel, dim3(gridn((size_t)Ms * H)), dim3(NT), 0, s, L.dx_e, (size_t)Ms * H, drop_p, seed, SITE_NMT_ENC(l - 1));
          UIC_LAUNCH_CHECK("dropout_grad_kernel");
        }
        UIC_TRY(uic_copy_launch(L.d_lay, L.dx_e, (size_t)Ms * H * 4, s));
        d_top = L.d_lay;
      }
    }
    // encoder embeddings: x0 = relu(linear(emb))
    UIC_TRY(uic_relu_mask_bwd_launch(dt, L.dx_e, off(L.xl[0], (size_t)B * W, dt), 1.f, L.dpre_e, (size_t)Ms * W, s));
    {
      const UicGemmTnSeg seg{L.xe, W, W};
      const WDest d1{G->enc_lin_w, W, 0, W};
      UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.dpre_e, W, W, &seg, 1, Ms, &d1, 1, s, false, L.tA, L.tB));
    }
    UIC_TRY(uic_colsum_launch(dt, L.dpre_e, Ms, W, W, G->enc_lin_b, L.colscratch, L.colscratch_floats, s));
    {
      UicGemmParams g = gemm_base(dt, Ms, W);
      add_seg(g, L.dpre_e, W, L.enc_lin_wT, W, W);
      g.C = L.dxe; g.ldc = W; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    return uic_embed_bwd_sorted_launch(dt, L.dxe, nullptr, src, 1, Ms, 1, Vs, W, 0.f, 0, G->enc_lut, L.embed_scratch, s);
  }
};

}  // namespace

extern "C" {

size_t uic_nmt_workspace_bytes(const uic_nmt_dims* d) {
  if (nmt_check(d)) return 0;
  return nmt_layout(*d, nullptr, nullptr).total;
}

void* uic_nmt_workspace_ptr(const uic_nmt_dims* d, void* workspace, const char* name) {
  if (nmt_check(d) || !workspace || !name) return nullptr;
  const NmtLayout L = nmt_layout(*d, nullptr, workspace);
  struct { const char* n; void* p; } tab[] = {{"dec_bwd_dbg", L.dec_bwd_dbg}, {"dec_fwd_dbg", L.dec_fwd_dbg}, {"d_cq", L.d_cq_all}, {"dscore", L.dscore_all}, {"d_pre", L.d_pre_all}};
  for (auto& e : tab)
    if (!strcmp(e.n, name)) return e.p;
  return nullptr;
}

int uic_nmt_forward_loss(const uic_nmt_dims* d, const uic_nmt_weights* w, const int64_t* src, const int32_t* lengths_host,
                         const int32_t* lengths_dev, const int64_t* tgt, int32_t training, uint32_t seed, void* workspace,
                         float* loss_out, int32_t* stats_out, float* outputs_out, float* attn_out, float* context_out,
                         void* stream) {
  UIC_TRY(nmt_check(d));
  UIC_REQUIRE(w && src && lengths_host && lengths_dev && tgt && workspace && loss_out, "nmt_forward_loss: null pointer");
  hipStream_t s = (hipStream_t)stream;
  static thread_local Nmt st;
  UIC_TRY(st.init(d, w, src, lengths_host, tgt, training, seed, workspace, nullptr));
  UIC_TRY(st.refresh(s));
  UIC_TRY(st.encoder_fwd(lengths_dev, s));
  UIC_TRY(st.decoder_fwd(s));
  UIC_TRY(st.loss_fwd(loss_out, stats_out, s));
  const int dt = st.dt;
  if (outputs_out) UIC_TRY(uic_to_f32_launch(dt, off(st.L.out_all, st.BH, dt), outputs_out, (size_t)st.Td * st.BH, s));
  if (attn_out) UIC_TRY(uic_copy_launch(attn_out, st.L.attn_all, (size_t)st.Td * st.B * st.S * 4, s));
  if (context_out) UIC_TRY(uic_to_f32_launch(dt, off(st.L.xl[st.NL], st.BH, dt), context_out, (size_t)st.S * st.BH, s));
  return UIC_OK;
}

int uic_nmt_backward(const uic_nmt_dims* d, const uic_nmt_weights* w, const int64_t* src, const int32_t* lengths_host,
                     const int64_t* tgt, int32_t training, uint32_t seed, void* workspace, const uic_nmt_weights* grads,
                     void* stream) {
  UIC_TRY(nmt_check(d));
  UIC_REQUIRE(w && src && lengths_host && tgt && workspace && grads, "nmt_backward: null pointer");
  static thread_local Nmt st;
  UIC_TRY(st.init(d, w, src, lengths_host, tgt, training, seed, workspace, grads));
  return st.backward((hipStream_t)stream);
}

int uic_nmt_grad_ready_wait(void* stream, int32_t group) {
  NmtSide* ss = nullptr;
  UIC_TRY(nmt_side(&ss));
  UIC_REQUIRE(group == 0 || group == 1, "nmt_grad_ready_wait: group=%d must be 0 (generator) or 1 (decoder side)", group);
  UIC_REQUIRE(ss->grads_recorded, "nmt_grad_ready_wait: no uic_nmt_backward has run on this device yet");
  return uic_check_hip(hipStreamWaitEvent((hipStream_t)stream, group == 0 ? ss->ev_grad_gen : ss->ev_grad_dec, 0), "hipStreamWaitEvent");
}


// ---------------------------------------------------------------- translator (NMTModel.translateBatch, beam search)
}  // extern "C"

namespace {

struct TrLayout {
  int64_t* src_rep; int32_t* lens_dev;
  void* h[ML][2]; float* c[ML][2]; void* feed[2]; void* emb;
  float* targetq; float* attn; void* cvec; float* logits;
  float* cand_val; int* cand_idx; float* scores; int64_t* tok; int* prev_ks; int64_t* next_ys; float* attn_hist;
  int* done; int* flags;
  void* enc_ws; size_t enc_bytes;
  size_t total;
};

TrLayout tr_layout(const uic_nmt_dims& d, int K, int max_steps, void* ws) {
  TrLayout T;
  memset(&T, 0, sizeof(T));
  Bump b{(char*)ws, 0};
  const size_t Sz = uic_dtype_size(d.dtype);
  const size_t B = d.B, R = B * K, S = d.S, H = d.H, W = d.W, NL = d.layers, Vtp = vpad(d.Vt);
  T.src_rep = (int64_t*)b.take(S * R * 8);
  T.lens_dev = (int32_t*)b.take(R * 4);
  for (size_t l = 0; l < NL; ++l)
    for (int i = 0; i < 2; ++i) {
      T.h[l][i] = b.take(R * H * Sz);
      T.c[l][i] = (float*)b.take(R * H * 4);
    }
  for (int i = 0; i < 2; ++i) T.feed[i] = b.take(R * H * Sz);
  T.emb = b.take(R * W * Sz);
  T.targetq = (float*)b.take(R * H * 4);
  T.attn = (float*)b.take(R * S * 4);
  T.cvec = b.take(R * H * Sz);
  T.logits = (float*)b.take(R * Vtp * 4);
  T.cand_val = (float*)b.take(R * UIC_BEAM_MAX * 4);
  T.cand_idx = (int*)b.take(R * UIC_BEAM_MAX * 4);
  T.scores = (float*)b.take(R * 4);
  T.tok = (int64_t*)b.take(R * 8);
  T.prev_ks = (int*)b.take((size_t)max_steps * R * 4);
  T.next_ys = (int64_t*)b.take((size_t)max_steps * R * 8);
  T.attn_hist = (float*)b.take((size_t)max_steps * R * S * 4);
  T.done = (int*)b.take(B * 4);
  T.flags = (int*)b.take(64);
  // the encoder (and the weight copies) run through the training path's workspace on the replicated batch
  uic_nmt_dims d2 = d;
  d2.B = (int)R; d2.T = 2;
  T.enc_bytes = nmt_layout(d2, nullptr, nullptr).total;
  T.enc_ws = b.take(T.enc_bytes);
  T.total = (b.off + 255) & ~(size_t)255;
  return T;
}

}  // namespace

extern "C" {

size_t uic_nmt_translate_workspace_bytes(const uic_nmt_dims* d, int32_t beam_size, int32_t max_steps) {
  if (nmt_check(d) || beam_size < 1 || beam_size > UIC_BEAM_MAX || max_steps < 1) return 0;
  return tr_layout(*d, beam_size, max_steps, nullptr).total;
}

int uic_nmt_translate(const uic_nmt_dims* d, const uic_nmt_weights* w, const int64_t* src, int32_t beam_size, int32_t max_steps,
                      void* workspace, int64_t* hyp_out, float* score_out, float* attn_out, int32_t* n_iter_out, void* stream) {
  UIC_TRY(nmt_check(d));
  UIC_REQUIRE(w && src && workspace && hyp_out && score_out && n_iter_out, "nmt_translate: null pointer");
  UIC_REQUIRE(beam_size >= 1 && beam_size <= UIC_BEAM_MAX && beam_size <= d->Vt, "nmt_translate: beam_size=%d outside [1, %d]", beam_size, UIC_BEAM_MAX);
  UIC_REQUIRE(max_steps >= 1, "nmt_translate: max_steps=%d", max_steps);
  hipStream_t s = (hipStream_t)stream;
  const int K = beam_size, B = d->B, R = B * K, S = d->S, H = d->H, W = d->W, NL = d->layers, Vt = d->Vt, dt = d->dtype;
  const int Vtp = (int)vpad(Vt), H4 = 4 * H;
  const size_t Sz = uic_dtype_size(dt), RH = (size_t)R * H;
  const TrLayout T = tr_layout(*d, K, max_steps, workspace);
  // (1) encoder on the replicated batch, WITHOUT lengths: every position of every row is processed (translateBatch :327)
  hipLaunchKernelGGL(nmt_replicate_src_kernel, dim3(gridn((size_t)S * R)), dim3(NT), 0, s, src, S, B, K, T.src_rep);
  UIC_LAUNCH_CHECK("nmt_replicate_src_kernel");
  uic_nmt_dims d2 = *d;
  d2.B = R; d2.T = 2; d2.drop_p = 0.f;
  static thread_local Nmt st;
  {
    static thread_local int32_t lens_host[4096 * UIC_BEAM_MAX];
    UIC_REQUIRE(R <= 4096 * UIC_BEAM_MAX, "nmt_translate: too many rows (%d)", R);
    for (int i = 0; i < R; ++i) lens_host[i] = S;
    UIC_TRY(st.init(&d2, w, T.src_rep, lens_host, nullptr, 0, 0, T.enc_ws, nullptr));
    UIC_TRY(uic_check_hip(hipMemcpyAsync(T.lens_dev, lens_host, (size_t)R * 4, hipMemcpyHostToDevice, s), "memcpy lengths"));
  }
  UIC_TRY(st.refresh(s));
  UIC_TRY(st.encoder_fwd(T.lens_dev, s));
  UIC_TRY(st.wait_gen(s));                  // the generator's copies (side stream) are read from the first decode step on
  const NmtLayout& L = st.L;
  // (2) decoder state = encoder final states (slot 0 of the training path's state buffers), zero input feed
  for (int l = 0; l < NL; ++l) {
    UIC_TRY(uic_copy_launch(T.h[l][0], L.hd[l], RH * Sz, s));
    UIC_TRY(uic_copy_launch(T.c[l][0], L.cd[l], RH * 4, s));
  }
  UIC_TRY(uic_fill_launch(T.feed[0], 0, RH * Sz, s));
  hipLaunchKernelGGL(nmt_beam_init_kernel, dim3(gridn((size_t)R)), dim3(NT), 0, s, B, K, T.tok, T.scores, T.done, T.flags);
  UIC_LAUNCH_CHECK("nmt_beam_init_kernel");
  UIC_TRY(uic_fill_launch(T.prev_ks, 0, (size_t)max_steps * R * 4, s));   // (speculative steps of a frozen search gather parent 0)
  const void* ctx = off(L.xl[NL], (size_t)R * H, dt);            // memory bank [S, R, H] (slot 1 onwards)
  const size_t lds_att = sizeof(float) * ((size_t)H + S + 4 * (size_t)H);
  UicBeamParams bp;
  memset(&bp, 0, sizeof(bp));
  bp.n_img = B; bp.B = K; bp.L = max_steps; bp.V1 = Vt; bp.ldv = Vtp; bp.plain = 1;
  bp.logits = T.logits; bp.cand_val = T.cand_val; bp.cand_idx = T.cand_idx;
  int n_iter = 0;
  // (3) the main loop (:349-378): state slot 0 -> 1, then re-threaded back into slot 0 by the back-pointers
  for (int step = 0; step < max_steps; ++step) {
    UIC_TRY(uic_embed_fwd_launch(dt, w->dec_lut, Vt, W, T.tok, 1, R, 1, 0.f, 0, 0, 0, 0, T.emb, s));
    const void* x = nullptr;
    for (int l = 0; l < NL; ++l) {
      UicGemmParams g = gemm_base(dt, R, H4);
      g.lstm = 1; g.H = H;
      if (l == 0) {
        add_seg(g, T.emb, W, L.dec_w_ih[0], W + H, W);
        add_seg(g, T.feed[0], H, off(L.dec_w_ih[0], W, dt), W + H, H);
      } else {
        add_seg(g, x, H, L.dec_w_ih[l], H, H);
      }
      add_seg(g, T.h[l][0], H, L.dec_w_hh[l], H, H);
      g.bias = w->dec_b_ih[l]; g.bias2 = w->dec_b_hh[l];
      g.c_prev = T.c[l][0]; g.c_out = T.c[l][1]; g.h_out = T.h[l][1]; g.ldh = H;
      UIC_TRY(uic_gemm_launch(g, s));
      x = T.h[l][1];
    }
    {
      UicGemmParams g = gemm_base(dt, R, H);
      add_seg(g, x, H, L.attn_in_w, H, H);
      g.C = T.targetq; g.ldc = H; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    float* attn_t = T.attn_hist + (size_t)step * R * S;
    NMT_GATTN_FWD(R, lds_att, ctx, (const float*)T.targetq, S, R, H, attn_t, (void*)T.cvec, (const int64_t*)T.src_rep,
                  (const float*)nullptr, (const void*)nullptr);
    {
      UicGemmParams g = gemm_base(dt, R, H);
      add_seg(g, T.cvec, H, L.attn_out_w, 2 * H, H);
      add_seg(g, x, H, off(L.attn_out_w, H, dt), 2 * H, H);
      g.C = T.feed[1]; g.ldc = H; g.flags = UIC_GEMM_TANH;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    {
      UicGemmParams g = gemm_base(dt, R, Vt);
      add_seg(g, T.feed[1], H, L.gen_w, H, H);
      g.C = T.logits; g.ldc = Vtp; g.bias = w->gen_b; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    bp.t = step;
    UIC_TRY(uic_beam_topk_launch(bp, s));
    hipLaunchKernelGGL(nmt_beam_begin_kernel, dim3(1), dim3(1), 0, s, step, T.flags);
    UIC_LAUNCH_CHECK("nmt_beam_begin_kernel");
    hipLaunchKernelGGL(nmt_beam_advance_kernel, dim3(B), dim3(64), 0, s, B, K, step, T.cand_val, T.cand_idx, T.scores,
                       T.prev_ks, T.next_ys, T.tok, T.done, T.flags);
    UIC_LAUNCH_CHECK("nmt_beam_advance_kernel");
    // decStates.beamUpdate_ (:376): re-thread every state tensor of every sentence to the surviving parents
    const int* parents = T.prev_ks + (size_t)step * R;
    for (int l = 0; l < NL; ++l)
      UIC_TRY(uic_beam_gather_launch(dt, parents, R, K, H, T.h[l][1], T.h[l][0], nullptr, nullptr, T.c[l][1], T.c[l][0], nullptr, nullptr, s));
    UIC_TRY(uic_beam_gather_launch(dt, parents, R, K, H, T.feed[1], T.feed[0], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s));
    // `if not active: break` (:377-378): the device freezes the search by itself (nmt_beam_begin_kernel); the host looks at the
    // flags every SYNC_EVERY steps only -- at most SYNC_EVERY - 1 speculative steps run on a finished search, and the GPU no
    // longer idles through a host round trip + 13 launches after every step
    constexpr int SYNC_EVERY = 4;
    if ((step + 1) % SYNC_EVERY == 0 && step + 1 < max_steps) {
      int fl[4] = {0, 0, 0, 0};
      UIC_TRY(uic_check_hip(hipMemcpyAsync(fl, T.flags, 16, hipMemcpyDeviceToHost, s), "memcpy flags"));
      UIC_TRY(uic_check_hip(hipStreamSynchronize(s), "hipStreamSynchronize"));
      if (fl[2] || fl[1] == 0) break;                    // frozen already, or this very step finished the last sentence
    }
  }
  {
    // steps that really ran: flags[3] (a search that never froze ran them all)
    int fl[4] = {0, 0, 0, 0};
    UIC_TRY(uic_check_hip(hipMemcpyAsync(fl, T.flags, 16, hipMemcpyDeviceToHost, s), "memcpy flags"));
    UIC_TRY(uic_check_hip(hipStreamSynchronize(s), "hipStreamSynchronize"));
    n_iter = fl[3];
  }
  *n_iter_out = n_iter;
  // (4) read-out
  hipLaunchKernelGGL(nmt_beam_readout_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, K, S, n_iter, max_steps, T.scores, T.prev_ks, T.next_ys,
                     T.attn_hist, src, hyp_out, score_out, attn_out);
  UIC_LAUNCH_CHECK("nmt_beam_readout_kernel");
  return UIC_OK;
}

}  // extern "C"
