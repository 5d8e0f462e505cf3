// Internal declarations shared by the HIP translation units of libuic_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, MFMA 32x32 tiles, 160 KB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define UIC_F32 0
#define UIC_BF16 1

#define UIC_OK 0
#define UIC_EARG (-1)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void uic_set_error(const char* fmt, ...);
int uic_check_hip(hipError_t e, const char* what);

#define UIC_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      uic_set_error(__VA_ARGS__);              \
      return UIC_EARG;                         \
    }                                          \
  } while (0)

#define UIC_LAUNCH_CHECK(what)                                   \
  do {                                                           \
    int _e = uic_check_hip(hipGetLastError(), what);             \
    if (_e) return _e;                                           \
  } while (0)

#define UIC_TRY(expr)          \
  do {                         \
    int _e = (expr);           \
    if (_e) return _e;         \
  } while (0)

static inline size_t uic_dtype_size(int dtype) { return dtype == UIC_BF16 ? 2 : 4; }
static inline int uic_round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__
template <typename T> struct uic_vec;   // 16-byte vector of T
template <> struct uic_vec<float> { static constexpr int N = 4; };
template <> struct uic_vec<bf16_t> { static constexpr int N = 8; };

__device__ __forceinline__ float uic_to_f(float x) { return x; }
__device__ __forceinline__ float uic_to_f(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T uic_from_f(float x);
template <> __device__ __forceinline__ float uic_from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t uic_from_f<bf16_t>(float x) { return (bf16_t)x; }

// unpack a 16-byte chunk into floats (4 for f32, 8 for bf16)
template <typename T> __device__ __forceinline__ void uic_unpack(const uint4& v, float* f);
template <> __device__ __forceinline__ void uic_unpack<float>(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
  f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void uic_unpack<bf16_t>(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
// (as a vector conversion: ONE v_cvt_pk_bf16_f32 for the pair; two scalar conversions compile to two of them plus the packing)
__device__ __forceinline__ unsigned uic_pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
template <typename T> __device__ __forceinline__ uint4 uic_pack(const float* f);
template <> __device__ __forceinline__ uint4 uic_pack<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 uic_pack<bf16_t>(const float* f) {
  return make_uint4(uic_pack_bf16x2(f[0], f[1]), uic_pack_bf16x2(f[2], f[3]),
                    uic_pack_bf16x2(f[4], f[5]), uic_pack_bf16x2(f[6], f[7]));
}

// Counter-based dropout: the keep decision of element `idx` at dropout site `site` is a pure
// function of (seed, site, idx), so forward, backward and the exported test masks agree.
__device__ __forceinline__ float uic_drop_scale(unsigned seed, unsigned site, unsigned idx, float p, float inv_keep) {
  unsigned x = idx * 0x9E3779B1u ^ (seed + site * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  float u = (float)(x >> 8) * (1.0f / 16777216.0f);
  return u < p ? 0.f : inv_keep;
}

__device__ __forceinline__ float uic_uniform(unsigned seed, unsigned site, unsigned idx) {
  unsigned x = idx * 0x9E3779B1u ^ (seed + site * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float uic_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// tanh: exact libm form on the f32 path (parity), exp-based fast form on the bf16 path
template <typename T> __device__ __forceinline__ float uic_tanh(float x);
template <> __device__ __forceinline__ float uic_tanh<float>(float x) { return tanhf(x); }
template <> __device__ __forceinline__ float uic_tanh<bf16_t>(float x) {
  // 1 - 2 / (1 + e^{2x}) with v_exp_f32 / v_rcp_f32 (1 ulp-class hardware ops; saturates correctly at +-inf).
  // NB: __fdividef expands to the full IEEE division sequence on gfx950 (10 instructions), hence the builtins.
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// sigmoid: libm exp on the f32 path, hardware exp + fast reciprocal on the bf16 path
template <typename T> __device__ __forceinline__ float uic_sigmoid_t(float x);
template <> __device__ __forceinline__ float uic_sigmoid_t<float>(float x) { return 1.f / (1.f + expf(-x)); }
template <> __device__ __forceinline__ float uic_sigmoid_t<bf16_t>(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

__device__ __forceinline__ float uic_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over each row of 16 lanes with DPP modifiers only (no LDS-pipe shuffles): xor 1, xor 2 (quad_perm),
// row_half_mirror, row_mirror.  Every lane of a row ends up with the row's sum.
__device__ __forceinline__ float uic_row16_sum(float v) {
#define UIC_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
  UIC_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  UIC_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  UIC_DPP_ADD(0x141);   // row_half_mirror
  UIC_DPP_ADD(0x140);   // row_mirror
#undef UIC_DPP_ADD
  return v;
}

__device__ __forceinline__ float uic_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif  // __HIPCC__

// gemm_tn.hip: set (by the calling host thread, around its launches) to keep uic_gemm_tn_launch on the 2-stage kernel even
// for grids of at most one workgroup per CU -- for launches that must share CUs with another stream's LDS-heavy workgroups
extern thread_local int g_uic_tn_ring_off;
// measurement knobs of the fused training step (bits 8-15 of uic_topdown_dims.recurrence; tools/ab_knobs.py): NOT part of the
// interface, every bit off = the shipped behaviour
extern thread_local int g_uic_knobs;
#define UIC_KNOB_TWO_WG_STREAMS 0x100  // every chunk's two weight-gradient shares on TWO extra streams (default: one; two for the last chunk only)
#define UIC_KNOB_CHUNK_TN128 0x200     // beside the BPTT chain: the LSTM / h2att chunk gradients on the 128 x 128 kernel
#define UIC_KNOB_LOGIT_TN128 0x400     // beside the BPTT chain: the logit weight gradient on the 128 x 128 kernel
#define UIC_KNOB_CHUNK_SK2 0x800       // beside the BPTT chain: two K slices for the chunk gradients
#define UIC_KNOB_LAST_BESIDE 0x1000    // the last chunk's gradients (after the loop) dispatched like the others (default: as if alone on the chip)
#define UIC_KNOB_LAST_ONE_STREAM 0x2000 // the last chunk's two shares on one stream
#define UIC_KNOB_SHORT_FIRST 0x4000     // the first decode step is a chunk of its own (the BPTT loop ends on it): see uic_topdown_xe_train_step
#define UIC_KNOB_MASK 0xff00

// ---------------------------------------------------------------- GEMM (gemm.hip)
#define UIC_GEMM_RELU 1      // v = max(v, 0)
#define UIC_GEMM_ACCUM 2     // C += v
#define UIC_GEMM_OUT_F32 4   // C is float regardless of the operand dtype
#define UIC_GEMM_TANH 8      // v = tanh(v)
// kernel choice for the large-GEMM path, per call (measurement tools; 0 = the dispatcher's own choice):
#define UIC_GEMM_FORCE_128 0x100   // the 128 x 128 LDS-DMA kernel (gemm.hip)
#define UIC_GEMM_FORCE_256 0x200   // the ping-pong kernel (gemm_pp.hip), 256-row tile; an ineligible problem is an error
#define UIC_GEMM_FORCE_192 0x400   // the same with its 192-row tile
#define UIC_GEMM_FORCE_PP128 0x800 // the same with its 128-row tile (128 x 256)
#define UIC_GEMM_FORCE_MASK 0xF00
#define UIC_GEMM_NO_RING 0x1000     // the 128 x 128 kernel with two LDS buffers even where the dispatcher would take its three-buffer ring (A/B)
#define UIC_GEMM_MAX_SEG 4

// One K-segment: C += A[M,K] * B[Nrows,K]^T.  Segments are summed, which expresses
// torch.cat([...],1) followed by a Linear without materialising the concatenation.
struct UicGemmSeg {
  const void* A; const void* B;
  int K, lda, ldb;
};

struct UicGemmParams {
  int dtype;                // operand dtype (UIC_F32 / UIC_BF16)
  int M, N;
  int nseg;
  UicGemmSeg seg[UIC_GEMM_MAX_SEG];
  void* C; int ldc;
  const float* bias;        // [N] or null
  const float* bias2;       // [N] or null
  const float* addend; int add_mod, ld_add;   // optional f32 [add_mod, ld_add]: C[row, col] += addend[row % add_mod, col] (before ReLU etc.)
  int flags;
  // pack_wrapper semantics (AttModel.py:44-53): row m = n*R + r is live iff r < row_len[n]
  const int* row_len; int R;
  // dropout applied after ReLU
  float drop_p; unsigned seed; unsigned site;
  int drop_row0;            // the dropout element index of C[row, col] is (row + drop_row0) * N + col
  void* C_pre; int ldc_pre; // optional (skinny path, operand dtype): the value BEFORE dropout goes here, C gets the dropped one
  // ---- fused LSTM cell epilogue (N must be 4*H) ----
  int lstm; int H;
  const float* pre1; int ldpre1;   // [M,4H] added to the gate pre-activations (may be null)
  const float* pre2; int ldpre2;
  const float* c_prev;             // [M,H] or null (= 0)
  float* c_out;                    // [M,H]
  void* h_out; int ldh;            // [M,H] operand dtype
  void* h_drop; int ldhd;          // dropout(h) copy, or null
  void* gates_out;                 // [M,4H] activated gates (i,f,g,o), operand dtype, or null
  // ---- split-K over workgroups (large-GEMM path only): slice z of `splitk` writes its raw partial tile to
  // slab[z][M][N] (f32, dense); uic_splitk_reduce_launch sums the slabs in a fixed order (deterministic)
  int splitk; float* slab;
  // ---- fused backward-of-ReLU epilogue (ping-pong kernel only, gemm_pp.hip): v = (acc + acc_src[row, col]); then, with
  // mask_act (operand dtype, the forward activation), v = mask_act[row, col] > 0 ? v * mask_scale : 0 -- d att' = d p_att W_ctx2att
  // + the attention's own share, masked by att' > 0 and scaled by the dropout keep factor, written as the bf16 operand of
  // att_embed's weight gradient in one pass (was: GEMM into f32, then a 94 MB relu_mask_bwd pass)
  const float* acc_src; int ld_acc_src;
  const void* mask_act; int ld_mask_act; float mask_scale;
  // ---- f32 A operand (ping-pong kernel only, dtype bf16): seg[0].A is f32 [M, K] (lda in floats) and is rounded to bf16 on its
  // way to LDS, exactly as uic_cast_f32_launch rounds; with a_copy the column-0 workgroups also store that bf16 image
  // [M, ld_a_copy] (the weight gradient's operand) -- att_embed on the loader's f32 region features without the cast pass
  int a_f32; void* a_copy; int ld_a_copy;
  // ping-pong kernel, bf16 output: a second output [M, ldc] = bf16(2^clamp(2 log2(e) c, +-UIC_E2_CLAMP)) of the ROUNDED bf16 value c that
  // goes to C -- e^{2 p_att} for the persistent recurrence's attention, made where p_att is made (uic_exp2x2_launch's values)
  void* C_exp2;
};
// C[row, c - col0] = sum_z slab[z][row, c] for c in [col0, col0 + ncols)
// ... with the output rows placed by a list (rows r < map_rows -> row map[r] of C; entries outside [0, map_limit) dropped)
bool uic_splitk_reduce_rows_ok(const float* slab, int M, int N, const float* C, int ldc);
int uic_splitk_reduce_rows_launch(const float* slab, int splitk, int M, int N, const int* map, int map_rows, int map_limit, float* C, int ldc,
                                  hipStream_t s);
int uic_splitk_reduce_launch(const float* slab, int splitk, int M, int N, int col0, int ncols, float* C, int ldc, hipStream_t s,
                             int accumulate = 0);   // accumulate: C += sum_z slab[z]
// true if the large-GEMM (LDS-DMA) path accepts this single-segment problem
bool uic_gemm_glds_eligible(int dtype, int K);

int uic_gemm_launch(const UicGemmParams& p, hipStream_t stream);
// the 256 x 256 x 64 ping-pong kernel (gemm_pp.hip): bf16, one K segment, K a multiple of 128 per split-K slice
bool uic_gemm_pp_eligible(const UicGemmParams& p);
int uic_gemm_pp_rows(int M, int N, int tallest = 256, int K = 512);
int uic_gemm_pp_launch(const UicGemmParams& p, int rows, hipStream_t stream);

// ---------------------------------------------------------------- TN GEMM (gemm_tn.hip): C[i,j] = sum_k A[k,i] B[k,j]
#define UIC_GEMM_TN_MAX_SEG 4
struct UicGemmTnSeg { const void* B; int ldb; int ncols; };   // [K, ncols] bf16 row-major; ncols % 128 == 0
struct UicGemmTnParams {
  const void* A; int lda;       // [K, M] bf16 row-major
  int M, N, K;                  // N = sum of the segments' ncols; K % 64 == 0
  int nseg;
  UicGemmTnSeg seg[UIC_GEMM_TN_MAX_SEG];
  int splitk; float* slab;      // splitk > 1 (or ndst == 0): raw f32 partial tiles into slab[z][M][N]
  // splitk == 1 and ndst > 0: the epilogue writes (or, with `accumulate`, adds to) the destinations directly:
  // columns [col0, col0 + ncols) of the product go to C[row * ldc + (col - col0)]
  int ndst; int accumulate;
  struct { float* C; int ldc, col0, ncols; } dst[UIC_GEMM_TN_MAX_SEG];
};
// C_i[row, c - col0_i] (+)= sum_z slab[z][row, c] for every destination i, one launch
struct UicSlabDest { float* C; int ldc, col0, ncols; };
int uic_splitk_reduce_multi_launch(const float* slab, int splitk, int M, int N, const UicSlabDest* dst, int nd, int accumulate, hipStream_t s);
bool uic_gemm_tn_eligible(const UicGemmTnParams& p);
int uic_gemm_tn_launch(const UicGemmTnParams& p, hipStream_t s);
// the 256 x 256 ping-pong form (gemm_tn_pp.hip): whole pairs of 64-row K tiles per split-K slice
bool uic_gemm_tnpp_eligible(const UicGemmTnParams& p);
int uic_gemm_tnpp_launch(const UicGemmTnParams& p, hipStream_t s);
// kernel choice of a weight-gradient call (wgrad_tn's `how`; uic_linear_wgrad passes bits 8-9 and 16-23 of its last argument):
#define UIC_TN_FORCE_128 0x100     // the 128 x 128 kernel (gemm_tn.hip)
#define UIC_TN_FORCE_256 0x200     // the 256 x 256 ping-pong kernel (gemm_tn_pp.hip); an ineligible problem is an error
#define UIC_TN_SPLITK(n) (((n) & 0xff) << 16)   // K slices (0: the dispatcher's choice)

// ---------------------------------------------------------------- attention (attention.hip)
struct UicAttnParams {
  int dtype, N, R, A, H;
  const float* att_h;      // [N,A]  h2att(h) incl. bias
  const void* p_att;       // [N,R,A]
  const void* att;         // [N,R,H]
  const float* w_alpha;    // [A]
  const float* b_alpha;    // [1]
  const float* mask; int ldmask;   // [N,R] or null
  float* alpha;            // [N,R] out (fwd) / in (bwd)
  void* ctx; int ldctx;    // [N,H] operand dtype (fwd out)
  // backward step
  const float* dctx; int lddctx;   // [N,H]
  int dctx_nslab; size_t dctx_slab_stride;   // > 1: d ctx = sum_z dctx[z * stride + n * lddctx + h] (split-K partial slabs) ...
  float* dctx_sum; int ld_dctx_sum;          // ... and the sum is left here (read by the deferred accumulation pass)
  float* de;               // [N,R]
  void* d_att_h;           // [N,A] operand dtype
  // backward step, optional: rows whose caption has no live position at or behind this decode step (step >= row_len[n]) carry an
  // all-zero gradient -- the fast kernel writes their zeros without reading their 74 KB of operands
  const int* row_len; int step;
};
int uic_attention_fwd_launch(const UicAttnParams& p, hipStream_t s);
int uic_attention_bwd_step_launch(const UicAttnParams& p, hipStream_t s);

struct UicAttnAccumParams {
  int dtype, N, R, A, H, T;          // T = number of executed decode steps
  const float* att_h_all;  // [T,N,A]
  const float* alpha_all;  // [T,N,R]
  const float* de_all;     // [T,N,R]
  const float* dctx_all; int lddctx; size_t dctx_step_stride;  // [T][N,lddctx]
  const void* p_att;       // [N,R,A]
  const float* w_alpha;
  float* d_att;            // [N,R,H] fp32 out (overwritten)
  void* d_p_att;           // [N,R,A] operand dtype out
  float* d_walpha_part;    // [N,A+1] per-row partial of (d w_alpha, d b_alpha)
  const int* row_len;      // optional [N]: decode steps >= row_len[n] of row n carry zero gradients (d e = d ctx = 0: the caption
                           // ended): the round-6 kernels sum over the row's first min(T, row_len[n]) steps only -- the same sums
};
int uic_attention_bwd_accum_launch(const UicAttnAccumParams& p, hipStream_t s);

// ---------------------------------------------------------------- persistent recurrence (rnn_persist.hip)
// Decode steps [t0, t1) of the TopDown recurrence (P/models/AttModel.py:129-154, :430-446, :538-558) in ONE launch.
// All state / activation buffers are the time-major workspace slabs of topdown.hip (step stride N*H etc.).
struct UicRnnFwdParams {
  int dtype, N, R, t0, t1;
  int row0, Nrows;                   // filled by the launcher (slabs of <= 640 caption rows)
  int force_safe;                    // UIC_REC_SAFE
  const float* gx;                   // [T, N, 4H] xt W_x^T + b_ih + b_hh of att_lstm
  const float* gfc;                  // [N, 4H] fc' W_fc^T, or null
  const void* att_w_ih; int ld_att_ih;   // [4H, ld]: columns [0, H) multiply h_lang_prev
  const void* att_w_hh;              // [4H, H]
  const void* lang_w_ih;             // [4H, 2H]: [att_res | h_att]
  const void* lang_w_hh;             // [4H, H]
  const float* lang_b_ih; const float* lang_b_hh;
  const void* h2att_w; const float* h2att_b;     // [A, H]
  const float* w_alpha; const float* b_alpha;
  const void* p_att; const void* att;            // [N, R, A], [N, R, H]
  const int* live_inv; void* hdrop_live;   // optional (weight-stationary training kernel): live_inv[t * N + n] = the row of hdrop_live [., H] that
                                     // position (t, n)'s dropped h_lang is ALSO stored to, or -1 (the live-position logit layer's compact operand)
  const void* e_att;                 // [N, R, A] bf16 e^{2 p_att} (uic_exp2x2_launch): what the weight-stationary TRAINING kernel's attention reads in
                                     // place of p_att (required there); ignored by the decode and the generic kernels
  const float* mask; int ldmask;                 // [N, R] or null
  void* h_att; void* h_lang; float* c_att; float* c_lang;     // [(T+1), N, H]: slot t is the state before step t
  void* gates1; void* gates2;                    // [T, N, 4H] activated gates for the backward pass, or null
  float* att_h_all; float* alpha_all; void* ctx_all; void* hdrop_all;   // [T, N, .]
  float drop_p; unsigned seed;
  const void* xbase;                 // filled by the launcher: lowest address of h_att / h_lang / ctx_all (one buffer descriptor)
  unsigned* sync;                    // uic_rnn_persist_sync_bytes() bytes, zeroed by the launcher unless sync_zeroed
  int sync_zeroed;                   // the caller cleared the FIRST launch's sync block itself (with other buffers, off the critical path)
  unsigned long long* dbg; int dbg_T; // optional [256][dbg_T][16] phase time stamps (100 MHz), indexed by absolute step
  unsigned* status;                  // sticky status words (uic_topdown_dims.rnn_status) or null
  // ---- decode mode (AttModel._sample, P/models/AttModel.py:198-253; bf16 only): every step also embeds the row's input
  // token, runs the logit layer and picks the next token inside the launch.  gx is then unused (gfc holds fc' W^T + b_ih + b_hh)
  int dec;                           // 0: teacher-forced recurrence; 1: decode
  const void* dec_embed_relu;        // [V1, E = H] relu(embedding table) in bf16 (uic_rnn_decode_embed_relu_launch)
  const void* dec_xw; int dec_ld_xw; // att_lstm.weight_ih columns of xt: [4H, ld], K = E contiguous
  float dec_xt_drop;                 // dropout on relu(embed) (train-mode sampling pass) with (seed, UIC_SITE_EMBED)
  void* dec_xt_all;                  // [T, N, E] the embedded inputs, kept for a backward pass, or null
  const void* dec_logit_w; const float* dec_logit_b; int dec_V1, dec_V1p;   // [V1, H]
  float* dec_logits; size_t dec_logits_step;   // [N, V1p] f32 of step t at dec_logits + t * dec_logits_step (0: one buffer reused)
  float* dec_part;                   // [N, 32, 4] per-workgroup (max, sum exp, arg max) of a row's logits
  int* dec_tok; int* dec_unf;        // [N] next input token / still-unfinished flag (exchanged between the steps)
  int64_t* dec_seq; float* dec_seq_logp; int dec_ld_out;   // [N, ld_out]
  const int64_t* dec_forced; int dec_ld_forced;            // optional tokens replacing the draws
  int dec_sample_max;                // 1: arg max (lowest index on ties), 0: multinomial draw from softmax(logits)
  unsigned dec_draw_seed;
};
size_t uic_rnn_persist_sync_bytes();
// every persistent launch of this process on a device waits for the previous one (any stream).  enter .. leave is one critical
// section per device (the device of the STREAM): enter takes the device's lock and makes `s` wait for the previous launch, leave
// records this launch's event and releases the lock, abandon releases it on an error path.  Use the scope:
int uic_persist_gate_enter(hipStream_t s, int* dev_out);
int uic_persist_gate_leave(hipStream_t s, int dev);
void uic_persist_gate_abandon(int dev);
struct UicPersistGateScope {
  hipStream_t s = nullptr; int dev = -1; bool held = false;
  int enter(hipStream_t st) { s = st; const int rc = uic_persist_gate_enter(s, &dev); held = rc == 0; return rc; }
  int leave() { held = false; return uic_persist_gate_leave(s, dev); }
  ~UicPersistGateScope() { if (held) uic_persist_gate_abandon(dev); }
};
bool uic_rnn_decode_persist_eligible(int dtype, int N, int H, int A, int R, int E, int V1);
size_t uic_rnn_decode_part_floats(int N);
// after uic_rnn_fwd_persist_launch in decode mode: the reference's `if unfinished.sum() == 0: break` (AttModel.py:236-238) --
// log-probs recorded at steps after every row had finished are zeroed (their tokens already are).  status (the caller's
// rnn_status words or NULL): a timed-out persistent launch poisons the captions instead (token -1, log-prob NaN)
int uic_rnn_decode_finish_launch(int64_t* seq, float* seq_logp, int N, int L, int ld, const int* status, hipStream_t s);
int uic_rnn_decode_embed_relu_launch(const void* embed_w, int table_dtype, void* out_bf16, int V1, int E, hipStream_t s);
bool uic_rnn_persist_eligible(int dtype, int N, int H, int A, int R);
int uic_rnn_fwd_persist_launch(const UicRnnFwdParams& p, hipStream_t s);

// BPTT of decode steps [t_lo, t_hi) of the same recurrence (latest step first) in ONE launch (rnn_bwd_persist.hip; bf16):
// per step lang_lstm cell backward, d[att_res | h_att | h_lang_prev] = dG2 W2, attention backward (scores, softmax,
// d att_h), d h_att += d att_h W_h2att, att_lstm cell backward, d[h_lang_prev | h_att_prev] = dG1 W1rec -- the six dependent
// launches per step of Step::bwd_step (topdown.hip).  Reads what the forward pass saved, writes what the weight-gradient
// GEMMs and the deferred attention accumulation read afterwards (dg1_all, dg2_all, datth_all, de_all, the d ctx columns of
// dx2_all); the gradients carried from step to step enter and leave through dc_att, dc_lang, dx1 and the h_lang columns of
// dx2_all[t_hi] / dx2_all[t_lo] -- the buffers of the launch chain's UNSPLIT form only: once the chain splits its d x GEMMs
// (Step::bptt_split() > 0, the default at 640 rows) it carries d h in its split-K slabs instead, so all chunks of one training
// step have to be run by the same of the two (Step::bwd_persist_ok is constant within a step).
struct UicRnnBwdParams {
  int N, R, t_lo, t_hi;
  int first;                         // 1: step t_hi - 1 is the last executed decode step (nothing carried in)
  int row0, Nrows;                   // filled by the launcher (slabs of <= 640 caption rows)
  int force_safe;                    // UIC_REC_SAFE
  const void* w2T;                   // [3H, 4H]: rows = inputs [att_res | h_att | h_lang_prev] of lang_lstm, K = its gate columns
  const void* w1recT;                // [2H, 4H]: rows = inputs [h_lang_prev | h_att_prev] of att_lstm
  const void* h2attT;                // [H, A]
  const float* w_alpha;
  const void* p_att; const void* att;            // [N, R, A], [N, R, H]
  const void* gates1; const void* gates2;        // [T, N, 4H] activated gates saved by the forward pass
  const float* c_att; const float* c_lang;       // [(T+1), N, H]
  const float* att_h_all; const float* alpha_all;   // [T, N, A], [T, N, R]
  const float* dhdrop;               // [T, N, H] gradient w.r.t. dropout(h_lang) from the logit layer
  float drop_p; unsigned seed;
  void* dg1_all; void* dg2_all;      // [T, N, 4H] out
  float* dx2_all;                    // [T, N, 3H]: columns [0, H) of every step out (d att_res), columns [2H, 3H) carry
  float* dx1;                        // [N, 2H] carry
  float* dc_att; float* dc_lang;     // [N, H] carry
  float* de_all; void* datth_all;    // [T, N, R], [T, N, A] out
  unsigned* sync;                    // uic_rnn_persist_sync_bytes() bytes; zeroed by the launcher unless sync_zeroed
  int sync_zeroed;                   // the caller zeroed this launch's sync block itself (one memset for all chunks of a step)
  unsigned long long* dbg; int dbg_T;   // optional [256][dbg_T][16] phase time stamps (100 MHz), indexed by absolute step
  unsigned* status;                  // sticky status words (uic_topdown_dims.rnn_status) or null
};
// nmt_persist.hip: the pivot decoder's target-step loop (NMT_Models.Decoder.forward, P/models/NMT_Models.py:228-262) as ONE launch.
// All slabs are time-major [step][B][512] in the operand dtype (bf16) unless noted; layer 0's embedding share of the gates
// (with both biases) comes precomputed in gx_d0, its weight_ih pointer is already offset to the input-feed columns.
#define UIC_NMT_PERSIST_MAX_LAYERS 4
struct UicNmtDecParams {
  int B, S, Td, NL;
  const void* out_all;               // [(Td + 1)] : slot t = the input feed of step t (slot 0 zeros), slot t + 1 = step t's output
  void* out_pre;                     // [Td] tanh output before dropout (backward pass)
  const float* gx_d0;                // [Td][B][4 x 512] f32
  void* hd[UIC_NMT_PERSIST_MAX_LAYERS]; float* cd[UIC_NMT_PERSIST_MAX_LAYERS];         // [(Td + 1)] h (bf16) / c (f32), slot 0 = initial state
  void* hdrop[UIC_NMT_PERSIST_MAX_LAYERS]; void* gates_d[UIC_NMT_PERSIST_MAX_LAYERS];  // [Td] dropped h between layers; activated gates [B][4 x 512]
  const void* w_ih[UIC_NMT_PERSIST_MAX_LAYERS]; int ld_ih[UIC_NMT_PERSIST_MAX_LAYERS]; const void* w_hh[UIC_NMT_PERSIST_MAX_LAYERS];
  const float* b_ih[UIC_NMT_PERSIST_MAX_LAYERS]; const float* b_hh[UIC_NMT_PERSIST_MAX_LAYERS];   // layers >= 1
  const void* ctx; const float* ctxw;   // [S][B][512] encoder context (bf16) and context x W_in (f32)
  float* attn_all;                   // [Td][B][S]
  void* cvec_all;                    // [Td] attention context
  const void* attn_out_w;            // [512][1024]
  float drop_p; unsigned seed;
  unsigned* sync;                    // uic_rnn_persist_sync_bytes() bytes (zeroed by the launcher)
  int sync_zeroed;                  // the caller has already cleared `sync` for this launch (uic_zero_list_launch with its other buffers)
  unsigned* status; int force_safe;
  int row0, Nrows;                   // 0, B
  unsigned long long* dbg;           // optional (UIC_REC_STAMPS): [256 workgroups][Td][16] s_memrealtime stamps (100 MHz)
};
// its BPTT (2 layers, batch <= 128): the chain's per-step buffers, plus two step-indexed exchange slabs
struct UicNmtDecBwdParams {
  int B, S, Td;
  const float* d_out_all;            // [Td][B][512] f32: d outputs from the generator
  const void* out_pre;               // [Td] tanh outputs (bf16)
  void* d_pre_all;                   // [Td] bf16 (out: linear_out's weight-gradient operand)
  float* d_cq_all;                   // [Td][B][2 x 512] f32 (out: the d c half; read by the deferred attention accumulation)
  const float* attn_all; const void* ctx; const float* ctxw;
  float* dscore_all;                 // [Td][B][S] (out)
  const void* gates_d[2]; const float* cd[2]; void* dg_d[2];   // activated gates, c states [(Td + 1)], d gates (out)
  const void* woutT;                 // [2 x 512][512]  = linear_out.weight^T
  const void* w1T;                   // [2 x 512][4 x 512] = [W_ih_1^T ; W_hh_1^T]
  const void* w0T;                   // [2 x 512][4 x 512] = [W_ih_0[:, W:]^T ; W_hh_0^T]
  float* dfeed_x; float* dq_att_x;   // [Td][B][512] f32 exchange slabs (d input feed, the attention's share of d q)
  float* dh_init[2];                 // [B][2 x 512] f32: second halves receive d h_l(-1)
  float* dc_init[2];                 // [B][512] f32: d c_l(-1)
  float drop_p; unsigned seed;
  unsigned* sync; unsigned* status; int force_safe;
  int sync_zeroed;                  // the caller has already cleared `sync` for this launch (uic_zero_list_launch with its other buffers)
  int row0, Nrows;
  unsigned long long* dbg;           // optional (UIC_REC_STAMPS): [256 workgroups][Td][16] s_memrealtime stamps (100 MHz)
};
// One layer of the pivot encoder's packed bidirectional LSTM (NMT_Models.Encoder, P/models/NMT_Models.py:95-135; nn.LSTM over a
// pack_padded_sequence) as ONE launch, both directions side by side: workgroups 0-15 of a row group own the forward direction's
// 256 units, 16-31 the backward direction's.  Iteration k runs time step k of the forward and S - 1 - k of the backward direction
// (the BPTT launch walks k the other way).  Rows are length-sorted: row b is alive at step st iff b < nb[st].
#define UIC_NMT_ENC_MAX_S 64
struct UicNmtEncParams {
  int B, S;
  int nb[UIC_NMT_ENC_MAX_S];         // rows alive per time step
  void* x_out;                       // [(S + 2)][B][512] bf16: slot st + 1 = h of step st, [fwd 256 | bwd 256] (slots 0 / S + 1 zero; all zeroed by the caller)
  const void* w_hh[2];               // forward: [4 x 256][256] recurrent weights;  backward pass: [256][4 x 256] = W_hh^T
  const float* gx[2];                // [S][B][4 x 256] f32: W_ih x + b_ih + b_hh (forward pass)
  float* c[2];                       // [(S + 2)][B][256] f32 cell states (zeroed by the caller)
  void* gates[2];                    // [S][B][4 x 256] bf16 activated gates (written forward, read backward)
  // backward pass only
  void* dgates[2];                   // [S][B][4 x 256] bf16 (zeroed by the caller: padded positions stay zero)
  const float* d_top;                // [S][B][512] f32: gradient w.r.t. this layer's outputs
  const float* dh_init; int ld_dh_init;   // [B][ld] f32, columns [dir * 256, +256): d h of the final state (the decoder's initial state)
  const float* dc_init; int ld_dc_init;
  unsigned* sync; unsigned* status; int force_safe;
  int sync_zeroed;                  // the caller has already cleared `sync` for this launch (uic_zero_list_launch with its other buffers)
  int row0, Nrows;
};
bool uic_nmt_enc_persist_eligible(int dtype, int B, int S, int H);
int uic_nmt_enc_fwd_persist_launch(const UicNmtEncParams& p, hipStream_t s);
int uic_nmt_enc_bwd_persist_launch(const UicNmtEncParams& p, hipStream_t s);
bool uic_nmt_dec_bwd_persist_eligible(int dtype, int B, int S, int H, int NL);
int uic_nmt_dec_bwd_persist_launch(const UicNmtDecBwdParams& p, hipStream_t s);
bool uic_nmt_dec_persist_eligible(int dtype, int B, int S, int H, int NL);
int uic_nmt_dec_persist_launch(const UicNmtDecParams& p, hipStream_t s);
bool uic_rnn_bwd_persist_eligible(int dtype, int N, int H, int A, int R);
int uic_rnn_bwd_persist_launch(const UicRnnBwdParams& p, hipStream_t s);

// ---------------------------------------------------------------- pointwise (pointwise.hip)
int uic_cast_f32_launch(int dtype, const float* src, void* dst, size_t n, hipStream_t s);
int uic_to_f32_launch(int dtype, const void* src, float* dst, size_t n, hipStream_t s);
#define UIC_CAST_MULTI 6
int uic_copy_multi_launch(int count, const void* const* src, void* const* dst, const size_t* bytes, hipStream_t s);
int uic_cast_f32_multi_launch(int dtype, int count, const float* const* src, void* const* dst, const size_t* n, hipStream_t s);   // several casts, one launch
int uic_fill_launch(void* dst, int value_byte, size_t bytes, hipStream_t s);
// dst[i] = bf16(2^clamp(2 log2(e) src[i], -UIC_E2_CLAMP, UIC_E2_CLAMP)) = e^{2 src[i]} for |src| < 20.8: the factored tanh of the
// persistent recurrence's attention phase, tanh(p + h) = 1 - 2 / (1 + e^{2p} e^{2h}).  n % 8 == 0, 16-byte aligned.
#define UIC_E2_CLAMP 60.f
int uic_exp2x2_launch(const void* src_bf16, void* dst_bf16, size_t n, hipStream_t s);
int uic_copy_launch(void* dst, const void* src, size_t bytes, hipStream_t s);
int uic_fill_value_launch(int dtype, void* dst, size_t n, float value, hipStream_t s);   // n elements of the operand dtype = value   // device -> device, 4-byte granules
// zero up to four buffers (16-byte aligned, sizes multiples of 16) in ONE launch instead of one memset node each
int uic_zero4_launch(void* p0, size_t b0, void* p1, size_t b1, void* p2, size_t b2, void* p3, size_t b3, hipStream_t s);
#define UIC_ZERO_LIST 16
// the same for a list of buffers (16-byte aligned / sized each), one launch per UIC_ZERO_LIST of them
int uic_zero_list_launch(void* const* ptrs, const size_t* bytes, int n, hipStream_t s);
// dst[cols, ldd] = src[rows, lds]^T, zero-filling dst columns rows..ldd-1
int uic_transpose_launch(int dtype, const void* src, int rows, int cols, int lds, void* dst, int ldd, hipStream_t s);
// up to UIC_TRANSPOSE_MULTI transposes (same argument meaning, one entry each) in ONE launch
#define UIC_TRANSPOSE_MULTI 12
struct UicTransposeJob { const void* src; void* dst; int rows, cols, lds, ldd; };
int uic_transpose_multi_launch(int dtype, int count, const UicTransposeJob* jobs, hipStream_t s);
// out[c] = sum_r src[r, c]  (deterministic two-stage; src operand dtype or f32)
int uic_colsum_launch(int src_dtype, const void* src, int rows, int cols, int lds, float* out, float* scratch,
                      size_t scratch_floats, hipStream_t s);
// dst[n, c] = sum_t src[t, n, c]
int uic_sum_steps_launch(int dtype, const void* src, int T, size_t step_elems, void* dst, hipStream_t s);
// table in f32 (table_dtype = UIC_F32) or, for bf16 outputs only, in bf16
int uic_embed_fwd_t_launch(int dtype, const void* table, int table_dtype, int V1, int E, const int64_t* tokens, int ldtok, int N, int T,
                           float drop_p, unsigned seed, unsigned site, size_t idx_base, int relu, void* out, hipStream_t s);
int uic_embed_fwd_launch(int dtype, const float* table, int V1, int E, const int64_t* tokens, int ldtok, int N, int T,
                         float drop_p, unsigned seed, unsigned site, size_t idx_base, int relu, void* out, hipStream_t s);
// the embedding gradient (nn.Embedding backward; skip_token < 0: none, nn.Embedding padding_idx otherwise) with the positions bucketed by token first (a stable counting sort), so that runs of equal tokens are summed in
// registers and every table row is STORED by one owner in a fixed order -- no floating-point atomics, bit-reproducible:
// dtable is overwritten; scratch = uic_embed_bwd_sorted_scratch_ints(N, T, V1, E) ints
size_t uic_embed_bwd_sorted_scratch_ints(int N, int T, int V1, int E);
int uic_embed_bwd_sorted_launch(int dtype, const float* dxt, const void* xt, const int64_t* tokens, int ldtok, int N, int T,
                                int V1, int E, float drop_p, long skip_token, float* dtable, int* scratch, hipStream_t s);
// the two halves of it: `prepare` needs only the tokens (zeroes dtable, buckets the positions), `gather` the gradients.
// split > 0: positions bucketed by (t >= split, token); gather then adds the share of decode steps [0, split) (half 0) or
// [split, T) (half 1) per call
int uic_embed_bwd_sorted_prepare(const int64_t* tokens, int ldtok, int N, int T, int V1, int E, float* dtable, int* scratch, hipStream_t s,
                                 int split = 0);
int uic_embed_bwd_sorted_gather(int dtype, const float* dxt, const void* xt, const int64_t* tokens, int ldtok, int N, int T,
                                int V1, int E, float drop_p, long skip_token, float* dtable, int* scratch, hipStream_t s,
                                int split = 0, int half = 0);
// column sums of a small f32 [rows, ncols] matrix into two destinations (columns [0, n0) -> out0, the rest -> out1), one launch
int uic_colsum_small_launch(const float* part, int rows, int ncols, int n0, float* out0, float* out1, hipStream_t s);
// dst = (act > 0 ? scale : 0) * grad ; grad f32, act/dst operand dtype
int uic_relu_mask_bwd_launch(int dtype, const float* grad, const void* act, float scale, void* dst, size_t n, hipStream_t s);
// seq_per_img > 1 (features given per image, caption rows = image * S + j):
//   expand_rows: dst[(i*S + j) * row + e] = cast(src[i * row + e])                       (dst dtype: dtype_out)
//   expand_drop: out[(i*S + j), e] = cast(y[i, e] * dropout keep-scale of element ((i*S + j) * row + e) at `site`)
//   relu_mask_bwd_fold: dst[i, e] = cast(sum_j (act[(i*S+j), e] > 0 ? grad[(i*S+j), e] * scale : 0))
int uic_expand_rows_launch(int dtype_out, const float* src, void* dst, int n_img, int S, size_t row, hipStream_t s);
int uic_expand_drop_launch(int dtype, const float* y, void* out, int n_img, int S, size_t row, float drop_p, unsigned seed,
                           unsigned site, hipStream_t s);
int uic_relu_mask_bwd_fold_launch(int dtype, const float* grad, const void* act, float scale, void* dst, int n_img, int S,
                                  size_t row, hipStream_t s);

// scheduled sampling (AttModel.py:130-143): used[n, t] = u_mask(n) < ss_prob ? draw from softmax(logits_prev[n]) : labels[n, t]
int uic_ss_sample_launch(const float* logits_prev, int N, int V1, int ldv, const int64_t* labels, int ld_labels, int t,
                         float ss_prob, unsigned seed, int64_t* used, int ld_used, hipStream_t s);
int uic_copy_tokens_launch(const int64_t* src, int ld_src, int N, int T, int64_t* dst, int ld_dst, hipStream_t s);

// ---------------------------------------------------------------- BatchNorm1d of att_embed (batchnorm.hip)
size_t uic_bn_scratch_floats(int NR, int C);
// batch statistics over the live rows (row (n, r) live iff r < row_len[n]; all rows if row_len is null):
// stat[0:C] = mean, stat[C:2C] = 1/sqrt(var + eps); running stats updated in place when non-null
int uic_bn_stats_launch(int in_dtype, const void* x, int NR, int R, int C, const int* row_len, float* part, float momentum,
                        float eps, float* stat, float* run_mean, float* run_var, hipStream_t s, float rep = 1.f);
int uic_bn_stats_running_launch(const float* run_mean, const float* run_var, int C, float eps, float* stat, hipStream_t s);
int uic_bn_apply_launch(int in_dtype, int out_dtype, const void* x, int NR, int R, int C, const int* row_len, const float* stat,
                        const float* gamma, const float* beta, int zero_padded, void* out, hipStream_t s);
// d (f32, in place) <- gradient w.r.t. the BatchNorm input y; dgamma / dbeta out; red: 3C floats of scratch
int uic_bn_bwd_launch(int dtype, float* d, const void* y, int NR, int R, int C, const int* row_len, const float* stat,
                      const float* gamma, int training, float* part, float* red, float* dgamma, float* dbeta, hipStream_t s);
int uic_bn_fold_weight_launch(int dtype, const float* W, const float* gamma, const float* beta, const float* b, int H, int D,
                              void* Weff, float* beff, hipStream_t s);
int uic_bn_fold_grad_launch(const float* W, const float* gamma, const float* beta, float* dW, const float* db, int H, int D,
                            float* dgamma, float* dbeta, hipStream_t s);

struct UicLstmBwdParams {
  int dtype, M, H;
  const float* dh0; int lddh0;   // up to three dh sources (null = absent)
  const float* dh1; int lddh1;
  const float* dh2; int lddh2;
  float drop_p; unsigned seed; unsigned site; size_t drop_base;  // dropout applied to dh0 only (out-dropout)
  // two more dh sources given as split-K partial slabs (uic_linear_partials): dh += sum_z slabX[z * strideX + m * ldX + u]
  const float* slabA; int ldA, nA; size_t strideA;
  const float* slabB; int ldB, nB; size_t strideB;
  float* dc;                     // [M,H] in: dc from step t+1 ; out: dc for step t-1  (in place)
  const void* gates;             // [M,4H] activated (i,f,g,o)
  const float* c_prev;           // [M,H] or null
  const float* c;                // [M,H]
  void* dgates;                  // [M,4H] operand dtype out
};
int uic_lstm_bwd_launch(const UicLstmBwdParams& p, hipStream_t s);
// d h = d att_h[N, A] h2attT[H, A]^T + the two slab sources, then the nn.LSTMCell backward of UicLstmBwdParams -- the `h2att`
// input-gradient GEMM and the att_lstm cell backward of a BPTT step in ONE launch (bptt_fused.hip; bf16)
struct UicH2attCellParams {
  int dtype, N, H, A;
  const void* datth;             // [N, A]
  const void* h2attT;            // [H, A]
  const float* slabA; int ldA, nA; size_t strideA;
  const float* slabB; int ldB, nB; size_t strideB;
  float* dc;                     // [N, H] in / out
  const void* gates;             // [N, 4H] activated (i, f, g, o)
  const float* c_prev;           // [N, H] or null
  const float* c;                // [N, H]
  void* dgates;                  // [N, 4H] out
};
bool uic_h2att_cell_bwd_eligible(const UicH2attCellParams& p);
int uic_h2att_cell_bwd_launch(const UicH2attCellParams& p, hipStream_t s);
int uic_maxout_lstm_bwd_launch(const UicLstmBwdParams& p, hipStream_t s);   // gates/dgates are [M,5H]; dh = (dh0 + dh1) * dropout
int uic_sample_fixup_launch(int N, int L, int ld, const int* n_unfinished, int64_t* seq, float* seq_logp, hipStream_t s);

struct UicXeParams {
  int dtype, M, V1, ldv;         // logits f32 [M, ldv] in; dlogits [M, ldv] operand dtype out
  const float* logits;
  void* dlogits;
  const int64_t* target; int ldtarget; int target_col0;   // target[n, target_col0 + t]
  const float* mask; int ldmask; int mask_col0;
  int N;                         // row m = t*N + n
  const float* inv_den;          // device scalar: 1 / sum(mask)
  const float* grad_scale; int ldscale; int scale_col0;   // optional: gradient weight of row (n,t) instead of mask/den (SCST)
  float* row_loss;               // [M]
  float* logprobs; size_t lp_step_stride, lp_row_stride;  // optional full log-probs out [n][t][v]
  int write_grad;
  int* score_stats;              // optional: [0] += rows whose arg-max (lowest index on ties) is the target, [1] += rows with target != 0
  const int* row_map;            // optional [M]: row m of logits / dlogits is position row_map[m] = t * N + n (target, mask, grad_scale and
                                 // are indexed by THAT, with the column offsets of step 0; row_loss by m).  Entries outside
                                 // [0, row_map_limit) -- the list's -1 padding -- are rows with zero gradient and no loss entry
  int row_map_limit;
};
int uic_xe_launch(const UicXeParams& p, hipStream_t s);
// rows by index: out[m] = src[map[m]] (rows [M, Mpad) of out, and rows whose index is outside [0, src_rows), cleared) /
// dst[map[m]] = src[m] (indices outside [0, dst_rows) skipped); row_bytes % 16 == 0
// the ascending list of the positions t * N + n (p < M) with mask[n * ld + col0 + t] != 0, padded with -1 up to out_len entries
// inv (optional, [M]): inv[p] = the list index of position p or -1; zero / zero_bytes (optional): a region cleared by the same launch
// row_len (optional, [N]): row_len[n] = 1 + the last t with mask[n, col0 + t] != 0 (0: none)
int uic_live_list_launch(const float* mask, int ld, int col0, int N, int M, int* out, int out_len, hipStream_t s, int* inv = nullptr,
                         void* zero = nullptr, size_t zero_bytes = 0, int* row_len = nullptr);
int uic_gather_rows_launch(const void* src, const int* map, int src_rows, void* out, int M, int Mpad, size_t row_bytes, hipStream_t s);
int uic_scatter_rows_launch(const void* src, const int* map, void* dst, int dst_rows, int M, size_t row_bytes, hipStream_t s);
// general log-softmax backward given dense upstream grad g [N,T,V1] (API-compat path):
// dlogits = g - softmax * sum(g)
int uic_logsoftmax_bwd_launch(int dtype, void* dlogits, int M, int V1, int ldv, int N, const float* g,
                              size_t g_step_stride, size_t g_row_stride, const float* logprobs, hipStream_t s);
int uic_masked_sum_launch(const float* x, const float* mask, int ldmask, int col0, int N, int T, float* out_sum,
                          float* out_inv, hipStream_t s);
int uic_reduce_sum_launch(const float* x, size_t n, float scale_by_dev_ptr_or_one, const float* scale, float* out, hipStream_t s);

struct UicAdamParams {
  float* p; const float* g; float* m; float* v; size_t n;
  float lr, beta1, beta2, eps, bc1, bc2, grad_scale;
  // clip_grad_norm (P/misc/optimizer.py:99): if sqnorm != null, gradients are additionally scaled by
  // min(1, max_norm / (grad_scale * sqrt(sqnorm[0]) + 1e-6)), read on the device (no host sync)
  float max_norm; const float* sqnorm;
  // uic_adam_step_guarded: if guard != null and guard[0] != 0 (any bit set: an int32 status word or a float sum of them), the
  // launch changes nothing -- parameters and moments stay those of before the step
  const int32_t* guard;
};
// out[0] = sum_i g[i]^2, deterministic two-stage reduction; scratch >= 1024 floats
int uic_sqnorm_launch(const float* g, size_t n, float* scratch, float* out, hipStream_t s);
int uic_adam_launch(const UicAdamParams& a, hipStream_t s);
constexpr int UIC_ADAM_RANGES = 24;
struct UicAdamRanges { int count; size_t total; size_t lo[UIC_ADAM_RANGES]; size_t start[UIC_ADAM_RANGES]; };
int uic_adam_ranges_launch(const UicAdamParams& a, const UicAdamRanges& r, void* w_out, int w_dtype, hipStream_t s);

constexpr int UIC_NUNF_STRIPES = 64;
struct UicSampleParams {
  int dtype, N, V1, ldv, t, L;
  const void* logits;            // [N, ldv]
  int sample_max; float temperature; unsigned seed;
  int decoding_constraint;
  int64_t* seq;                  // [N, L]
  float* seq_logp;               // [N, L]
  int64_t* it;                   // [N] next input token
  int* unfinished;               // [N]
  int* n_unfinished;             // [(L+1) * UIC_NUNF_STRIPES] live-row counters per step, striped over 64 words (row n adds to
                                 // stripe n % 64): 640 same-address atomics per launch cost ~19 us, striped ones nothing
  const int64_t* forced;         // optional [N, L] tokens replacing the multinomial draw
  float* logprobs_out;           // optional [N, V1]
  int fc_mode;                   // FCModel_NMT._sample semantics (raw token fed forward, break before write)
  int ld_out;                    // row stride of seq / seq_logp (0 = L)
  // optional: the NEXT step's embedding row of every caption row, written by the workgroup that just chose its token
  // (xt_out[n, :] = dropout(relu(embed_table[it[n]])), element index embed_idx_base + n * E + e at site embed_site, seed `seed`):
  // one launch less per decode step
  const void* embed_table; int embed_table_dtype; int embed_V1, embed_E; float embed_drop_p; unsigned embed_site; size_t embed_idx_base; void* xt_out;
};
int uic_sample_step_launch(const UicSampleParams& p, hipStream_t s);
// ---- beam search bookkeeping (beam.hip): rows = (image, beam)
#define UIC_BEAM_MAX 16
struct UicBeamParams {
  int n_img, B, L, V1, ldv, t;
  int decoding_constraint, max_ppl;
  int plain;                           // 1: plain log-softmax top-B (NMT translator): no -1000 on the last index, no constraint
  const float* logits;                 // [n_img * B, ldv] of the current step
  float* cand_val; int* cand_idx;      // [n_img * B, B]
  int64_t* beam_seq_hist[2]; float* beam_lp_hist[2];   // [n_img, L, B] x 2 generations (read t & 1, write the other)
  const int64_t* beam_seq;             // set by the launcher: the generation holding steps < t
  float* beam_sum;                     // [n_img, B]
  int* parent;                         // [n_img * B] surviving parent beam of each new beam
  int64_t* it;                         // [n_img * B] next input tokens
  int* done_count; float* done_p; int64_t* done_seq; float* done_lp;   // [n_img], [n_img, L*B], [n_img, L*B, L] x 2
};
int uic_beam_step_launch(const UicBeamParams& p, hipStream_t s);
int uic_beam_topk_launch(const UicBeamParams& p, hipStream_t s);     // only the per-row log-softmax + top-B into cand_val / cand_idx
int uic_beam_gather_launch(int dtype, const int* parent, int rows, int B, int H, const void* h1s, void* h1d, const void* h2s, void* h2d,
                           const float* c1s, float* c1d, const float* c2s, float* c2d, hipStream_t s);
int uic_beam_final_launch(const UicBeamParams& p, int64_t* seq_out, float* lp_out, hipStream_t s);
int uic_dropout_mask_launch(float* out, size_t n, float p, unsigned seed, unsigned site, size_t base, hipStream_t s);
