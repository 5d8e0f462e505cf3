// Persistent, step-fused BPTT of the TopDown captioner's recurrence for gfx950 (MI355X): ONE launch runs the backward of
// decode steps [t_lo, t_hi) of AttModel._forward's loop body (P/models/AttModel.py:129-154 -> TopDownCore.forward
// :430-446 -> Attention.forward :538-558), latest step first, instead of six dependent launches per step
// (Step::bwd_step in topdown.hip: lstm_bwd, d x2 GEMM, attention backward, h2att GEMM, lstm_bwd, d x1 GEMM).
//
// Decomposition = the forward kernel's (rnn_persist.hip): caption rows are independent, so the chip is cut into 8 row
// groups (one per XCD where the placement allows, else the SAFE protocol), and inside a group workgroup `rank` owns 16
// hidden units for all of the group's <= 80 rows: 48 output columns of d[att_res | h_att | h_lang_prev] = dG2 W2, 16 of
// d att_h W_h2att, 32 of d[h_lang_prev | h_att_prev] = dG1 W1rec, both cells' pointwise backward for its units, and 2-3 of
// the group's rows in the attention phase.  What the gate gradients of a cell need (d h, d c of the SAME units) is
// therefore always local: the only exchanged data are dG2, d att_res, d att_h and dG1 of the step, and a step is four
// group barriers:
//     lang cell backward (local) -> dG2 | B1 | d x2 GEMM -> d att_res | B2 | attention backward -> d att_h | B3 |
//     h2att GEMM + att cell backward (local) -> dG1 | B4 | d x1 GEMM (local result, carried to step t - 1)
//
// UNLIKE the forward kernel this one is NOT weight-stationary and NOT exclusive: the fused training step runs it beside
// the side stream's throughput GEMMs (logit layer, weight gradients), which is where the step's time goes if the BPTT
// chain holds the chip.  So its footprint is kept to half a CU's registers and half its LDS -- 8 waves, the weight slices
// (344 KB per workgroup and step) streamed from L2 / the Infinity Cache each step, K split over the waves with operands
// going global -> VGPR directly in MFMA fragment layout, partial tiles summed through LDS in a fixed order -- and what it
// buys over the launch chain is the launch gaps, the per-launch ramps and the latencies that now overlap the barrier
// waits (weight fragments and saved activations are requested between `arrive` and `wait`).
#include "rnn_persist_common.h"
#include <stdlib.h>

namespace {

constexpr int BW_NW = 4;                                    // one wave per SIMD, at most half of its register file
constexpr int BW_NTH = BW_NW * 64;
constexpr int BW_RED_BYTES = BW_NW * MT_MAX * 64 * 16;     // 20 KB: one column tile's partial sums of all waves
constexpr int BW_FLAG_BYTES = 256;
#ifndef BW_RED_BUFS
#define BW_RED_BUFS 1
#endif
// (20.3 KB with one buffer: two 64 KB GEMM workgroups of the other stream still fit beside it; a second buffer would save
// one workgroup barrier per reduction round and cost the other stream half of its occupancy)
constexpr int BW_LDS_BYTES = BW_FLAG_BYTES + BW_RED_BUFS * BW_RED_BYTES;
#ifndef BW_DEPTH_V
#define BW_DEPTH_V 3
#endif
#ifndef BW_NT_V
#define BW_NT_V 0
#endif
#ifndef BW_PRIO_V
#define BW_PRIO_V 0
#endif
constexpr int BW_DEPTH = BW_DEPTH_V;                         // k-steps of operands in flight per wave
constexpr bool BW_NT = BW_NT_V != 0;                        // weights and region features as non-temporal loads
constexpr int H4 = 4 * HH;
constexpr int H3 = 3 * HH;
constexpr int AR = ATT_R / BW_NW;                           // regions per wave in the attention phase

__device__ __forceinline__ f32x4 bw_mma(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One K-split GEMM of a phase: acc[i][j] (16 x 16 tile: rows 16 i.. of the group, column tile j of this workgroup) over the
// k-steps of this wave (k-step = 32 K elements = 64 bytes of a row; wave w takes k-steps w + 4 m, walked from an offset of
// the workgroup's own so that the 32 workgroups of a group do not ask the L2 for the same lines at the same time).
// Operands go global -> VGPR directly in MFMA fragment layout, BW_DEPTH k-steps in flight.
template <int NCT, int NKS>
struct BwGemm {
  u32x4 fa[BW_DEPTH][MT_MAX], fb[BW_DEPTH][NCT];
  __amdgpu_buffer_rsrc_t ra, rb;
  unsigned aoff[MT_MAX], boff[NCT];
  int wave, rot;
  __device__ __forceinline__ unsigned kk(int m) const { return (unsigned)((wave + BW_NW * ((m + rot) & (NKS - 1))) * 64); }
  __device__ __forceinline__ void loadB(int buf, int m) {
#pragma unroll
    for (int j = 0; j < NCT; ++j) fb[buf][j] = bload<false, BW_NT>(rb, boff[j], kk(m));
  }
  __device__ __forceinline__ void loadA(int buf, int m) {
#pragma unroll
    for (int i = 0; i < MT_MAX; ++i) fa[buf][i] = bload<true>(ra, aoff[i], kk(m));
  }
  // weights of the first k-steps: requested before the barrier wait (they do not depend on the exchange)
  __device__ __forceinline__ void prefetch() {
#pragma unroll
    for (int m = 0; m < BW_DEPTH; ++m) loadB(m, m);
  }
  __device__ __forceinline__ void run(f32x4 (&acc)[MT_MAX][NCT]) {
#pragma unroll
    for (int m = 0; m < BW_DEPTH; ++m) loadA(m, m);
#pragma unroll
    for (int m = 0; m < NKS; ++m) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT_MAX; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = bw_mma(fa[m % BW_DEPTH][i], fb[m % BW_DEPTH][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      if (m + BW_DEPTH < NKS) { loadB(m % BW_DEPTH, m + BW_DEPTH); loadA(m % BW_DEPTH, m + BW_DEPTH); }
    }
  }
};

// The (row, unit) pairs a lane finishes: the 4 rows 4 lq + r of row tile `wave` in the D layout of a 16 x 16 tile, and
// row 4 lq + wave of the fifth tile (which the four waves share).  e[k] = element index in an [N, HH] slab (rows past the
// group's share clamped: their results are never stored), live mask in bit k.
constexpr int NP = 5;
struct RowSet { unsigned e[NP]; unsigned live; };
__device__ __forceinline__ RowSet bw_rows(const Ctx& c) {
  RowSet q;
  q.live = 0;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int rr = k < 4 ? 16 * c.wave + 4 * c.lq + k : 16 * BW_NW + 4 * c.lq + c.wave;
    if (rr < c.nrow) q.live |= 1u << k;
    q.e[k] = (unsigned)((c.rbegin + (rr < c.nrow ? rr : c.nrow - 1)) * HH + c.u0 + c.l15);
  }
  return q;
}

// saved activations the pointwise backward of one cell needs for the lane's pairs
struct PwOps {
  unsigned g[NP][2];        // activated gates (i | f << 16), (g | o << 16), bf16 bits
  float c[NP], cp[NP], dh0[NP];
};

// stores of exchanged data: plain in the XCD-local protocol, write-through (sc1) in the SAFE one.  `safe` is uniform; one
// instantiation of the step loop serves both protocols (two of them double the code and the register pressure).
__device__ __forceinline__ void bst_bf16(bool safe, __amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  const bf16_t b = (bf16_t)v;
  if (safe) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, b), r, off, 0, 16);
  else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, b), r, off, 0, 0);
}
__device__ __forceinline__ void bst_f32(bool safe, __amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  if (safe) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 16);
  else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0);
}
__device__ __forceinline__ float bld_f32(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// all addressing below is a uniform slab base (buffer descriptor) + a 32-bit lane offset derived from e[k]:
//   [N, HH] f32: 4 e      [N, 4 HH] bf16 (gate q): 8 e - 6 u + 2 q HH      [N, 3 HH] f32 (column block b): 12 e - 8 u + 4 b HH
__device__ __forceinline__ void bw_load_pw(const Ctx& c, PwOps& o, const void* gates, const float* cnew, const float* cold, const float* dh0) {
  const RowSet q = bw_rows(c);
  const unsigned u = (unsigned)(c.u0 + c.l15);
  const __amdgpu_buffer_rsrc_t rg = rsrc_of(gates), rc = rsrc_of(cnew), rp = rsrc_of(cold), rd = rsrc_of(dh0 ? dh0 : cnew);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const unsigned og = 8u * q.e[k] - 6u * u;
    const unsigned g0 = __builtin_amdgcn_raw_buffer_load_b16(rg, og, 0, 0), g1 = __builtin_amdgcn_raw_buffer_load_b16(rg, og + 2 * HH, 0, 0);
    const unsigned g2 = __builtin_amdgcn_raw_buffer_load_b16(rg, og + 4 * HH, 0, 0), g3 = __builtin_amdgcn_raw_buffer_load_b16(rg, og + 6 * HH, 0, 0);
    o.g[k][0] = g0 | (g1 << 16);
    o.g[k][1] = g2 | (g3 << 16);
    o.c[k] = bld_f32(rc, 4u * q.e[k]);
    o.cp[k] = bld_f32(rp, 4u * q.e[k]);
    o.dh0[k] = dh0 ? bld_f32(rd, 4u * q.e[k]) : 0.f;
  }
}
// nn.LSTMCell backward for the lane's (row, unit) pairs (same formulas as lstm_bwd_kernel, pointwise.hip): gate
// gradients to dg (exchanged: the next GEMM's A operand), d c updated in place
__device__ __forceinline__ void bw_cell(const Ctx& c, bool safe, const PwOps& o, const float (&dh)[NP], float (&dc)[NP], void* dg) {
  const RowSet q = bw_rows(c);
  const unsigned u = (unsigned)(c.u0 + c.l15);
  const __amdgpu_buffer_rsrc_t rg = rsrc_of(dg);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const float gi = __uint_as_float(o.g[k][0] << 16), gf = __uint_as_float(o.g[k][0] & 0xffff0000u);
    const float gg = __uint_as_float(o.g[k][1] << 16), go = __uint_as_float(o.g[k][1] & 0xffff0000u);
    const float tc = uic_tanh<bf16_t>(o.c[k]);
    const float d = dc[k] + dh[k] * go * (1.f - tc * tc);
    const float d_o = dh[k] * tc;
    if (q.live & (1u << k)) {
      const unsigned og = 8u * q.e[k] - 6u * u;
      bst_bf16(safe, rg, og, d * gg * gi * (1.f - gi));
      bst_bf16(safe, rg, og + 2 * HH, d * o.cp[k] * gf * (1.f - gf));
      bst_bf16(safe, rg, og + 4 * HH, d * gi * (1.f - gg * gg));
      bst_bf16(safe, rg, og + 6 * HH, d_o * go * (1.f - go));
    }
    dc[k] = d * gf;
  }
}
// sum of the waves' partial tiles of column tile j (fixed order: deterministic); out[k] = the value of the lane's pair k
template <int NCT>
__device__ __forceinline__ void bw_reduce(const Ctx& c, int& rb_next, const f32x4 (&acc)[MT_MAX][NCT], int j, float (&out)[NP]) {
  f32x4* red = (f32x4*)(c.smem + BW_FLAG_BYTES + rb_next * BW_RED_BYTES);
  if (BW_RED_BUFS > 1) rb_next ^= 1;
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i) red[(c.wave * MT_MAX + i) * 64 + c.lane] = acc[i][j];
  __syncthreads();
  f32x4 s = red[(0 * MT_MAX + c.wave) * 64 + c.lane];
  float s5 = ((const float*)(red + (0 * MT_MAX + BW_NW) * 64 + c.lane))[c.wave];
#pragma unroll
  for (int w = 1; w < BW_NW; ++w) {
    s += red[(w * MT_MAX + c.wave) * 64 + c.lane];
    s5 += ((const float*)(red + (w * MT_MAX + BW_NW) * 64 + c.lane))[c.wave];
  }
  out[0] = s[0]; out[1] = s[1]; out[2] = s[2]; out[3] = s[3]; out[4] = s5;
  if (BW_RED_BUFS == 1) __syncthreads();        // the buffer is rewritten by the next round
}
template <int NCT, int NKS>
__device__ __forceinline__ void bw_gemm_setup(const Ctx& c, BwGemm<NCT, NKS>& g, const void* A, int lda, const void* B, int ldb) {
  g.wave = c.wave; g.rot = c.rank & (NKS - 1);
  g.ra = rsrc_of(A);
  g.rb = rsrc_of(B);
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i) {
    int ar = 16 * i + c.l15;
    ar = ar < c.nrow ? ar : c.nrow - 1;
    g.aoff[i] = (unsigned)((ar * lda + c.lq * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < NCT; ++j) g.boff[j] = (unsigned)(((j * HH + c.u0 + c.l15) * ldb + c.lq * 8) * 2);
}

// att' / p_att chunks of one caption row: regions wave + 4 k, 16 bytes per lane
struct BSlot { u32x4 va[AR], vp[AR]; };
__device__ __forceinline__ void bw_load_va(const Ctx& c, BSlot& q, const void* att, int n, int R) {
  const __amdgpu_buffer_rsrc_t rv = rsrc_of((const bf16_t*)att + (size_t)n * R * HH);
#pragma unroll
  for (int k = 0; k < AR; ++k) q.va[k] = bload<false, BW_NT>(rv, (unsigned)(c.lane * 16), (unsigned)(min(c.wave + BW_NW * k, R - 1) * HH * 2));
}
__device__ __forceinline__ void bw_load_vp(const Ctx& c, BSlot& q, const void* p_att, int n, int R) {
  const __amdgpu_buffer_rsrc_t rp = rsrc_of((const bf16_t*)p_att + (size_t)n * R * HH);
#pragma unroll
  for (int k = 0; k < AR; ++k) q.vp[k] = bload<false, BW_NT>(rp, (unsigned)(c.lane * 16), (unsigned)(min(c.wave + BW_NW * k, R - 1) * HH * 2));
}

// attention backward of caption row n (Attention.forward :544-556 backward): d alpha = d ctx . att', softmax backward -> de,
// d att_h = w_alpha . sum_r de_r (1 - tanh^2(p_att + att_h)); all waves, AR regions each.  q holds row n's chunks on entry
// and row n_next's on exit (the caller passes n again when there is no next row: an unconditional refill keeps the
// chunks in ONE register set): each half is re-requested as soon as it has been consumed.
__device__ __forceinline__ void bw_attn_row(const Ctx& c, bool safe, const UicRnnBwdParams& p, BSlot& q, int n, int n_next, const float* dx2,
                                            const float* atth_t, const float* alpha_t, float* de_t, void* datth) {
  const int R = p.R;
  float* s_da4 = (float*)(c.smem + BW_FLAG_BYTES);     // [ATT_R][4] row-of-16 partial d alpha
  float* s_red = s_da4 + 4 * ATT_R;                    // [BW_NW][HH]
  // this row's d att_res (exchanged), att_h, w_alpha: 8 consecutive elements per lane
  const __amdgpu_buffer_rsrc_t rd = rsrc_of(dx2 + (size_t)n * H3), rh = rsrc_of(atth_t + (size_t)n * HH), rw = rsrc_of(p.w_alpha);
  const unsigned lo = (unsigned)(c.lane * 32);
  const u32x4 d0 = bload<true>(rd, lo, 0), d1 = bload<true>(rd, lo + 16, 0);
  const u32x4 a0 = bload<false>(rh, lo, 0), a1 = bload<false>(rh, lo + 16, 0);
  const float al_l = c.lane < R ? alpha_t[(size_t)n * R + c.lane] : 0.f;
  const float dc8[8] = {__uint_as_float(d0.x), __uint_as_float(d0.y), __uint_as_float(d0.z), __uint_as_float(d0.w),
                        __uint_as_float(d1.x), __uint_as_float(d1.y), __uint_as_float(d1.z), __uint_as_float(d1.w)};
#pragma unroll
  for (int k = 0; k < AR; ++k) {
    const int r = c.wave + BW_NW * k;
    float f[8];
    uic_unpack<bf16_t>(make_uint4(q.va[k].x, q.va[k].y, q.va[k].z, q.va[k].w), f);
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) part += dc8[j] * f[j];
    part = uic_row16_sum(part);
    if (c.l15 == 0 && r < R) s_da4[r * 4 + c.lq] = part;
    __builtin_amdgcn_sched_barrier(0);   // (one region at a time: left alone hipcc unpacks all of them first, 200+ registers)
  }
  bw_load_va(c, q, p.att, n_next, R);
  __syncthreads();
  float da_l = 0.f;
  if (c.lane < R) {
    const float4 qq = *(const float4*)(s_da4 + c.lane * 4);
    da_l = (qq.x + qq.y) + (qq.z + qq.w);
  }
  const float wbar = uic_wave_sum(al_l * da_l);
  const float de_l = al_l * (da_l - wbar);
  if (c.wave == 0 && c.lane < R) de_t[(size_t)n * R + c.lane] = de_l;
  const float ah[8] = {__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z), __uint_as_float(a0.w),
                       __uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z), __uint_as_float(a1.w)};
  float acc8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc8[j] = 0.f;
#pragma unroll
  for (int k = 0; k < AR; ++k) {
    const int r = c.wave + BW_NW * k;
    // (every wave holds de_r in lane r and r is uniform: a scalar read, no cross-lane traffic)
    float de = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, de_l), r < R ? r : 0));
    if (r >= R) de = 0.f;
    float f[8];
    uic_unpack<bf16_t>(make_uint4(q.vp[k].x, q.vp[k].y, q.vp[k].z, q.vp[k].w), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float th = uic_tanh<bf16_t>(f[j] + ah[j]);
      acc8[j] += de * (1.f - th * th);
    }
    // (the sums are needed HERE: left alone hipcc keeps every region's 8 factors and accumulates at the end)
    asm volatile("" : "+v"(acc8[0]), "+v"(acc8[1]), "+v"(acc8[2]), "+v"(acc8[3]), "+v"(acc8[4]), "+v"(acc8[5]), "+v"(acc8[6]), "+v"(acc8[7]));
    __builtin_amdgcn_sched_barrier(0);
  }
  bw_load_vp(c, q, p.p_att, n_next, R);
  const u32x4 w0 = bload<false>(rw, lo, 0), w1 = bload<false>(rw, lo + 16, 0);
  const float w[8] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w),
                      __uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z), __uint_as_float(w1.w)};
  float* dst = s_red + c.wave * HH + c.lane * 8;
  *(float4*)dst = make_float4(acc8[0] * w[0], acc8[1] * w[1], acc8[2] * w[2], acc8[3] * w[3]);
  *(float4*)(dst + 4) = make_float4(acc8[4] * w[4], acc8[5] * w[5], acc8[6] * w[6], acc8[7] * w[7]);
  __syncthreads();
  {
    const __amdgpu_buffer_rsrc_t ro = rsrc_of((bf16_t*)datth + (size_t)n * HH);
#pragma unroll
    for (int h = 0; h < HH / BW_NTH; ++h) {
      const int a = c.tid + h * BW_NTH;
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < BW_NW; ++wv) v += s_red[wv * HH + a];
      bst_bf16(safe, ro, (unsigned)(a * 2), v);
    }
  }
  __syncthreads();
}

// keeps the per-lane values a phase derives from the lane coordinates out of the other phases' live ranges (left alone
// hipcc hoists every step-invariant offset out of the step loop and spills them)
#define BW_OPAQUE(c)                                                               \
  asm volatile("" : "+v"((c).lane), "+v"((c).l15), "+v"((c).lq), "+v"((c).tid)); \
  (c).wave = __builtin_amdgcn_readfirstlane((c).wave); (c).u0 = __builtin_amdgcn_readfirstlane((c).u0);       \
  (c).rbegin = __builtin_amdgcn_readfirstlane((c).rbegin); (c).nrow = __builtin_amdgcn_readfirstlane((c).nrow); \
  (c).rank = __builtin_amdgcn_readfirstlane((c).rank);                                                        \
  asm volatile("" : "+s"((c).wave), "+s"((c).u0), "+s"((c).rbegin), "+s"((c).nrow), "+s"((c).rank))

__device__ __forceinline__ void bw_run(const UicRnnBwdParams& p, Ctx& c, bool safe) {
  const int N = p.N, R = p.R;
  const size_t NH = (size_t)N * HH;
  int* flag = (int*)c.smem;
  int rb_next = 0;                                  // LDS partial-sum buffer of the next reduction round
  // gradients carried from step to step, for this lane's (row, unit) pairs
  float dcl[NP], dca[NP], hl2[NP], hl1[NP], ha1[NP], dx2ha[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) { dcl[k] = dca[k] = hl2[k] = hl1[k] = ha1[k] = dx2ha[k] = 0.f; }
  if (!p.first) {
    const RowSet q = bw_rows(c);
    const unsigned u = (unsigned)(c.u0 + c.l15);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      dcl[k] = p.dc_lang[q.e[k]];
      dca[k] = p.dc_att[q.e[k]];
      hl2[k] = p.dx2_all[(size_t)p.t_hi * N * H3 + 3u * q.e[k] - 2u * u + 2 * HH];
      hl1[k] = p.dx1[2u * q.e[k] - u];
      ha1[k] = p.dx1[2u * q.e[k] - u + HH];
    }
  }
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * p.dbg_T + (p.t_hi - 1)) * 16 : nullptr;

  PwOps pw;
  {
    const int t = p.t_hi - 1;
    bw_load_pw(c, pw, (const bf16_t*)p.gates2 + (size_t)t * N * H4, p.c_lang + (size_t)(t + 1) * NH, p.c_lang + (size_t)t * NH, p.dhdrop + (size_t)t * NH);
  }
  for (int t = p.t_hi - 1; t >= p.t_lo; --t) {
    bf16_t* dg2 = (bf16_t*)p.dg2_all + (size_t)t * N * H4;
    bf16_t* dg1 = (bf16_t*)p.dg1_all + (size_t)t * N * H4;
    float* dx2 = p.dx2_all + (size_t)t * N * H3;
    bf16_t* datth = (bf16_t*)p.datth_all + (size_t)t * NH;
    BW_OPAQUE(c);
    if (dbg && c.tid == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
    // ---- A: lang_lstm cell backward (P/models/AttModel.py:441 backward; output dropout :443)
    {
      const RowSet q = bw_rows(c);
      float dh[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        float v = pw.dh0[k];
        if (p.drop_p > 0.f) v *= uic_drop_scale(p.seed, UIC_SITE_OUT0 + (unsigned)t, q.e[k], p.drop_p, inv_keep);
        dh[k] = v + hl2[k] + hl1[k];
      }
      bw_cell(c, safe, pw, dh, dcl, dg2);
    }
    if (dbg && c.tid == 0) dbg[1] = __builtin_amdgcn_s_memrealtime();
    BW_OPAQUE(c);
    // ---- B: d[att_res | h_att | h_lang_prev] = dG2 [W_ih | W_hh]  (48 columns of it)
    {
      BwGemm<3, 16> g;
      bw_gemm_setup(c, g, dg2 + (size_t)c.rbegin * H4, H4, p.w2T, H4);
      group_arrive(c);
      g.prefetch();
      if (!group_wait(c, flag)) return;
      if (dbg && c.tid == 0) dbg[2] = __builtin_amdgcn_s_memrealtime();
      f32x4 acc[MT_MAX][3];
#pragma unroll
      for (int i = 0; i < MT_MAX; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      g.run(acc);
      if (dbg && c.tid == 0) dbg[3] = __builtin_amdgcn_s_memrealtime();
      float s[NP];
      bw_reduce<3>(c, rb_next, acc, 0, s);
      {                  // d att_res: the attention phase of other workgroups reads it
        const RowSet q = bw_rows(c);
        const unsigned u = (unsigned)(c.u0 + c.l15);
        const __amdgpu_buffer_rsrc_t rx = rsrc_of(dx2);
#pragma unroll
        for (int k = 0; k < NP; ++k)
          if (q.live & (1u << k)) bst_f32(safe, rx, 12u * q.e[k] - 8u * u, s[k]);
      }
      group_arrive(c);
      bw_reduce<3>(c, rb_next, acc, 1, dx2ha);
      bw_reduce<3>(c, rb_next, acc, 2, hl2);
    }
    if (dbg && c.tid == 0) dbg[4] = __builtin_amdgcn_s_memrealtime();
    BW_OPAQUE(c);
    // ---- C: attention backward for rows rank, rank + 32, rank + 64 of the group
    {
      const float* atth_t = p.att_h_all + (size_t)t * NH;
      const float* alpha_t = p.alpha_all + (size_t)t * N * R;
      float* de_t = p.de_all + (size_t)t * N * R;
      BSlot slot;
      {   // (addresses do not depend on the exchange: in flight across the wait.  Unconditional -- a workgroup without a row
          // re-reads the group's last one -- so that the chunks are ONE register set whatever the control flow)
        const int n0 = c.rbegin + (c.rank < c.nrow ? c.rank : c.nrow - 1);
        bw_load_va(c, slot, p.att, n0, R);
        bw_load_vp(c, slot, p.p_att, n0, R);
      }
      if (!group_wait(c, flag)) return;
      if (dbg && c.tid == 0) dbg[5] = __builtin_amdgcn_s_memrealtime();
      for (int r0 = c.rank; r0 < c.nrow; r0 += PW)
        bw_attn_row(c, safe, p, slot, c.rbegin + r0, c.rbegin + (r0 + PW < c.nrow ? r0 + PW : r0), dx2, atth_t, alpha_t, de_t, datth);
    }
    if (dbg && c.tid == 0) dbg[6] = __builtin_amdgcn_s_memrealtime();
    BW_OPAQUE(c);
    // ---- D: d h_att += d att_h W_h2att (16 columns), then att_lstm cell backward (:434 backward)
    {
      BwGemm<1, 4> g;
      bw_gemm_setup(c, g, datth + (size_t)c.rbegin * HH, HH, p.h2attT, HH);
      group_arrive(c);
      g.prefetch();
      bw_load_pw(c, pw, (const bf16_t*)p.gates1 + (size_t)t * N * H4, p.c_att + (size_t)(t + 1) * NH, p.c_att + (size_t)t * NH, nullptr);
      if (!group_wait(c, flag)) return;
      if (dbg && c.tid == 0) dbg[7] = __builtin_amdgcn_s_memrealtime();
      f32x4 acc[MT_MAX][1];
#pragma unroll
      for (int i = 0; i < MT_MAX; ++i) acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      g.run(acc);
      float s[NP], dh[NP];
      bw_reduce<1>(c, rb_next, acc, 0, s);
#pragma unroll
      for (int k = 0; k < NP; ++k) dh[k] = dx2ha[k] + ha1[k] + s[k];
      bw_cell(c, safe, pw, dh, dca, dg1);
    }
    if (dbg && c.tid == 0) dbg[8] = __builtin_amdgcn_s_memrealtime();
    BW_OPAQUE(c);
    // ---- E: d[h_lang_prev | h_att_prev] = dG1 [W_ih[:, :H] | W_hh]  (32 columns; stays in this workgroup)
    if (t > 0) {
      BwGemm<2, 16> g;
      bw_gemm_setup(c, g, dg1 + (size_t)c.rbegin * H4, H4, p.w1recT, H4);
      group_arrive(c);
      g.prefetch();
      // the next step's lang cell operands (unused after the launch's last step)
      bw_load_pw(c, pw, (const bf16_t*)p.gates2 + (size_t)(t - 1) * N * H4, p.c_lang + (size_t)t * NH, p.c_lang + (size_t)(t - 1) * NH, p.dhdrop + (size_t)(t - 1) * NH);
      if (!group_wait(c, flag)) return;
      if (dbg && c.tid == 0) dbg[9] = __builtin_amdgcn_s_memrealtime();
      f32x4 acc[MT_MAX][2];
#pragma unroll
      for (int i = 0; i < MT_MAX; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      g.run(acc);
      bw_reduce<2>(c, rb_next, acc, 0, hl1);
      bw_reduce<2>(c, rb_next, acc, 1, ha1);
    }
    if (dbg && c.tid == 0) dbg[10] = __builtin_amdgcn_s_memrealtime();
    if (dbg) dbg -= 16;
  }
  // what the next chunk of steps (this kernel or the launch chain) continues from
  {
    const RowSet q = bw_rows(c);
    const unsigned u = (unsigned)(c.u0 + c.l15);
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (q.live & (1u << k)) {
        p.dc_lang[q.e[k]] = dcl[k];
        p.dc_att[q.e[k]] = dca[k];
        if (p.t_lo > 0) {
          p.dx2_all[(size_t)p.t_lo * N * H3 + 3u * q.e[k] - 2u * u + 2 * HH] = hl2[k];
          p.dx1[2u * q.e[k] - u] = hl1[k];
          p.dx1[2u * q.e[k] - u + HH] = ha1[k];
        }
      }
  }
}

// (one wave per SIMD with at most half of its registers, 41 KB of LDS: the rest of the CU stays available to the other stream)
__global__ __launch_bounds__(BW_NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void rnn_bwd_persist_kernel(const UicRnnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  if (BW_PRIO_V) __builtin_amdgcn_s_setprio(BW_PRIO_V);
  const int mode = setup_ctx(p, smem, c);
  if (mode == 0) return;
  bw_run(p, c, mode == 2);
}

}  // namespace

bool uic_rnn_bwd_persist_eligible(int dtype, int N, int H, int A, int R) {
  return dtype == UIC_BF16 && uic_rnn_persist_eligible(dtype, N, H, A, R);
}

int uic_rnn_bwd_persist_launch(const UicRnnBwdParams& p0, hipStream_t s) {
  UIC_REQUIRE(p0.sync && p0.t_hi > p0.t_lo && p0.t_lo >= 0 && p0.N > 0, "rnn_bwd_persist: bad arguments");
  static bool configured = false;
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)rnn_bwd_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BW_LDS_BYTES),
                          "hipFuncSetAttribute(rnn bwd persist)"));
    configured = true;
  }
  const int G = 8, cap = G * 16 * MT_MAX;     // caption rows one launch covers
  UicPersistGateScope gate;                   // (its workgroups have to be resident together: never beside another persistent launch)
  UIC_TRY(gate.enter(s));
  for (int r0 = 0; r0 < p0.N; r0 += cap) {
    UicRnnBwdParams p = p0;
    p.row0 = r0;
    p.Nrows = p0.N - r0 < cap ? p0.N - r0 : cap;
    p.sync = p0.sync + (size_t)(r0 / cap) * SY_WORDS;
    if (!p0.sync_zeroed) UIC_TRY(uic_check_hip(hipMemsetAsync(p.sync, 0, (size_t)SY_WORDS * 4, s), "hipMemsetAsync(rnn bwd sync)"));
    hipLaunchKernelGGL(rnn_bwd_persist_kernel, dim3(G * PW), dim3(BW_NTH), BW_LDS_BYTES, s, p);
    UIC_LAUNCH_CHECK("rnn_bwd_persist_kernel");
  }
  return gate.leave();
}
