// Persistent, step-fused recurrence of the TopDown captioner for gfx950 (MI355X): ONE launch runs decode steps
// [t0, t1) of AttModel._forward's loop body (P/models/AttModel.py:129-154 -> TopDownCore.forward :430-446 ->
// Attention.forward :538-558) -- att_lstm GEMM + cell, h2att, attention, lang_lstm GEMM + cell + output dropout --
// instead of four dependent launches per step.
//
// Why it can be split: every caption row's recurrence is independent of every other row's, only the weights are
// shared.  So the chip is cut into row GROUPS, one per XCD (32 CUs behind one 4 MB L2): group g owns caption rows
// [g*Rg, (g+1)*Rg) for all steps and never talks to another group.  Inside a group workgroup `rank` (one per CU) owns
// 16 hidden units = 64 gate columns of both LSTMs and 16 columns of h2att for all of the group's rows, and between 2
// and 3 of the group's rows in the attention phase; the only exchanged data are the rows' h_att / att_h / ctx /
// h_lang vectors (<= 80 rows x 512), which stay in that XCD's L2.  Phases are separated by a GROUP barrier (32
// arrivals on one counter) instead of a kernel boundary or a grid barrier.
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility): a CU's vector L1 is never refreshed by another CU's
// stores, the L2 of one XCD is shared by its CUs, the L2s of different XCDs are not coherent.  Groups are formed from
// the hardware's own XCC_ID register (never from blockIdx), so in the normal case all members of a group sit behind
// one L2: exchanged data are written with plain stores (they land in that L2), every storing wave drains vmcnt(0)
// before the workgroup arrives at the barrier, and EVERY load of exchanged data is an `sc1` load (bypasses the
// reader's L1).  If the dispatcher ever places the grid differently (an XCD with != 32 workgroups) the kernel runs in
// SAFE mode: groups by arrival ticket, exchanged data stored write-through (`sc1`) as well -- slower, still correct,
// so results never depend on placement.  Every spin is bounded; a timeout sets sync[SY_ERR] and the launch ends.
//
// GEMM phases: an output tile [<=80 rows] x [64 gate columns] per workgroup, K split over the 8 waves (wave w takes
// k-steps w, w+8, ...), operands go global -> VGPR directly in MFMA 16x16 fragment layout (no LDS staging: every
// operand byte is used by exactly one wave), the 8 partial tiles are summed through LDS and the cell update runs on
// the wave that owns the 16-row tile.  bf16 operands -> v_mfma_f32_16x16x32_bf16, f32 -> v_mfma_f32_16x16x4_f32.
#include "rnn_persist_common.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

// att_h = h2att(h_att) (P/models/AttModel.py:543): 16 columns of the group's rows
template <typename T, bool SAFE>
__device__ __forceinline__ void h2att_phase(Ctx& c, const T* h_att_new, const T* w, const float* b, float* att_h) {
  const bool owner = c.wave < c.MT;
  const int a = c.u0 + c.l15;
  const float bias = owner && b ? b[a] : 0.f;
  f32x4 acc[MT_MAX][1];
  zero_acc<1>(acc);
  const void* const As[1] = {h_att_new};
  const void* const Bs[1] = {w};
  const int ldb[1] = {HH};
  gemm_ksplit<T, 1, 1, false>(c, acc, As, Bs, ldb);
  f32x4* red = (f32x4*)c.smem;      // [wave][tile][lane]
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i)
    if (i < c.MT) red[(c.wave * MT_MAX + i) * 64 + c.lane] = acc[i][0];
  __syncthreads();
  if (owner) {
    f32x4 s = red[(0 * MT_MAX + c.wave) * 64 + c.lane];
#pragma unroll
    for (int w2 = 1; w2 < NWAVE; ++w2) s += red[(w2 * MT_MAX + c.wave) * 64 + c.lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * c.wave + 4 * c.lq + r;
      if (rr < c.nrow) st_x<SAFE>(att_h + (unsigned)((c.rbegin + rr) * HH + a), s[r] + bias);
    }
  }
  __syncthreads();
}

// Attention.forward after h2att (P/models/AttModel.py:544-556), all 8 waves per caption row (attention.hip's
// attn_fwd_fast_kernel with the row's att_h read past the L1).  A workgroup has up to three rows per step; their loads are
// software-pipelined through three register slots of 5 x 16 B per lane (p_att of a row, att' of a row), so that two
// slots' worth of loads are always in flight while the third is being consumed.
template <typename T> struct AttnCfg {
  static constexpr int VEC = 16 / (int)sizeof(T);
  static constexpr int CH = HH / VEC / 64;        // 16-byte chunks of a 512-wide row per lane (bf16: 1, f32: 2)
};
template <typename T, int NW> struct AttnSlot { uint4 v[ATT_R / NW][AttnCfg<T>::CH]; };
template <typename T> struct AttnHead { u32x4 ah[AttnCfg<T>::CH][AttnCfg<T>::VEC / 4]; };

// STREAM: the load is non-temporal (does not stay in the XCD's L2).  A workgroup re-reads the SAME rows of p_att and att'
// every decode step: 5.9 MB per XCD and step against a 4 MB L2, so with one policy for both the L2 thrashes; streaming att'
// lets the 2.95 MB of p_att rows stay resident.
template <typename T, int NW, bool STREAM>
__device__ __forceinline__ void attn_load_slot(const Ctx& c, int R, const T* base, AttnSlot<T, NW>& q) {
  constexpr int VEC = AttnCfg<T>::VEC, CH = AttnCfg<T>::CH;
#pragma unroll
  for (int u = 0; u < ATT_R / NW; ++u) {
    const int r = min(c.wave + u * NW, R - 1);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const u32x4* src = (const u32x4*)(base + (unsigned)(r * HH + (c.lane + 64 * k) * VEC));
      const u32x4 v = *src;
      q.v[u][k] = make_uint4(v.x, v.y, v.z, v.w);
    }
  }
}
template <typename T>
__device__ __forceinline__ void attn_load_head(const Ctx& c, const float* att_h_row, AttnHead<T>& q) {
  constexpr int VEC = AttnCfg<T>::VEC, CH = AttnCfg<T>::CH;
  const __amdgpu_buffer_rsrc_t rh = rsrc_of(att_h_row);
#pragma unroll
  for (int k = 0; k < CH; ++k)
#pragma unroll
    for (int v = 0; v < VEC / 4; ++v) q.ah[k][v] = bload<true>(rh, (unsigned)(((c.lane + 64 * k) * VEC + v * 4) * 4), 0);
}
// scores + softmax (+ mask renormalisation) of row n; returns this lane's weight (lane r < R holds alpha_r)
template <typename T, int NW>
__device__ __forceinline__ float attn_scores(const Ctx& c, const UicRnnFwdParams& p, int n, const AttnSlot<T, NW>& vp, const AttnHead<T>& hd,
                                             float* alpha) {
  constexpr int VEC = AttnCfg<T>::VEC, CH = AttnCfg<T>::CH;
  const int R = p.R;
  float* s_e = (float*)c.smem + 64;               // [R][4] row-of-16 partial scores (word 0 of smem is the barrier flag)
  float ah[CH][VEC], w[CH][VEC];
#pragma unroll
  for (int k = 0; k < CH; ++k)
#pragma unroll
    for (int v = 0; v < VEC / 4; ++v) {
      ah[k][v * 4 + 0] = __uint_as_float(hd.ah[k][v].x); ah[k][v * 4 + 1] = __uint_as_float(hd.ah[k][v].y);
      ah[k][v * 4 + 2] = __uint_as_float(hd.ah[k][v].z); ah[k][v * 4 + 3] = __uint_as_float(hd.ah[k][v].w);
      const float4 ww = *(const float4*)(p.w_alpha + (c.lane + 64 * k) * VEC + v * 4);
      w[k][v * 4 + 0] = ww.x; w[k][v * 4 + 1] = ww.y; w[k][v * 4 + 2] = ww.z; w[k][v * 4 + 3] = ww.w;
    }
  const float b_alpha = p.b_alpha ? p.b_alpha[0] : 0.f;
#pragma unroll
  for (int u = 0; u < ATT_R / NW; ++u) {
    const int r = c.wave + u * NW;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      float f[VEC];
      uic_unpack<T>(vp.v[u][k], f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) part += w[k][j] * uic_tanh<T>(f[j] + ah[k][j]);
    }
    part = uic_row16_sum(part);
    if (c.l15 == 0 && r < R) s_e[r * 4 + c.lq] = part;
  }
  __syncthreads();
  const float* mk = p.mask ? p.mask + (size_t)n * p.ldmask : nullptr;
  float e = -INFINITY;
  if (c.lane < R) {
    const float4 qq = *(const float4*)(s_e + c.lane * 4);
    e = (qq.x + qq.y) + (qq.z + qq.w) + b_alpha;
  }
  const float mx = uic_wave_max(e);
  // (bf16 path: one v_exp_f32 / v_rcp_f32 instead of libm's exp and an IEEE division -- this kernel is made of instruction issue)
  const float ex = c.lane < R ? (sizeof(T) == 2 ? __builtin_amdgcn_exp2f((e - mx) * 1.4426950408889634f) : expf(e - mx)) : 0.f;
  float wgt = ex * (sizeof(T) == 2 ? __builtin_amdgcn_rcpf(uic_wave_sum(ex)) : 1.f / uic_wave_sum(ex));
  if (mk) {
    wgt *= c.lane < R ? mk[c.lane] : 0.f;
    wgt = wgt / uic_wave_sum(wgt);
  }
  if (c.wave == 0 && c.lane < R) alpha[(unsigned)(n * R + c.lane)] = wgt;
  return wgt;
}
template <typename T, bool SAFE, int NW>
__device__ __forceinline__ void attn_context(const Ctx& c, const UicRnnFwdParams& p, int n, const AttnSlot<T, NW>& va, float wgt, T* ctx) {
  constexpr int VEC = AttnCfg<T>::VEC, CH = AttnCfg<T>::CH;
  const int R = p.R;
  float* s_red = (float*)c.smem + 64 + 4 * ((R + 3) & ~3);        // [NW][HH]
  float acc[CH][VEC];
#pragma unroll
  for (int k = 0; k < CH; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[k][j] = 0.f;
#pragma unroll
  for (int u = 0; u < ATT_R / NW; ++u) {
    const int r = c.wave + u * NW;
    float al = __shfl(wgt, r < R ? r : 0, 64);
    if (r >= R) al = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      float f[VEC];
      uic_unpack<T>(va.v[u][k], f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[k][j] += al * f[j];
    }
  }
#pragma unroll
  for (int k = 0; k < CH; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) s_red[c.wave * HH + (c.lane + 64 * k) * VEC + j] = acc[k][j];
  __syncthreads();
#pragma unroll
  for (int h = c.tid; h < HH; h += NW * 64) {
    float sacc = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) sacc += s_red[wv * HH + h];
    st_x<SAFE>(ctx + (unsigned)(n * HH + h), sacc);
  }
  __syncthreads();
}
// the attention phase of one workgroup: rows rank, rank + 32, rank + 64 of the group (<= 3).  SLOTS = 3: p_att and att' of
// a row plus p_att of the next one in flight; SLOTS = 2 (the weight-stationary kernel, whose registers hold weights):
// one slot in flight while the other is consumed.
template <typename T, bool SAFE, int NW, int SLOTS>
__device__ __forceinline__ void attn_phase(const Ctx& c, const UicRnnFwdParams& p, const float* att_h, float* alpha, T* ctx) {
  const int R = p.R;
  const int r0 = c.rank, r1 = c.rank + PW, r2 = c.rank + 2 * PW;
  if (r0 >= c.nrow) return;
  const bool has1 = r1 < c.nrow, has2 = r2 < c.nrow;
  const int n0 = c.rbegin + r0, n1 = c.rbegin + r1, n2 = c.rbegin + r2;
  const T* P = (const T*)p.p_att;
  const T* V = (const T*)p.att;
  AttnHead<T> h0, h1;
  if constexpr (SLOTS == 3) {
    AttnSlot<T, NW> s0, s1, s2;
    attn_load_head<T>(c, att_h + (size_t)n0 * HH, h0);
    attn_load_slot<T, NW, false>(c, R, P + (size_t)n0 * R * HH, s0);
    attn_load_slot<T, NW, true>(c, R, V + (size_t)n0 * R * HH, s1);
    if (has1) {
      attn_load_head<T>(c, att_h + (size_t)n1 * HH, h1);
      attn_load_slot<T, NW, false>(c, R, P + (size_t)n1 * R * HH, s2);
    }
    float wgt = attn_scores<T, NW>(c, p, n0, s0, h0, alpha);
    if (has1) attn_load_slot<T, NW, true>(c, R, V + (size_t)n1 * R * HH, s0);
    attn_context<T, SAFE, NW>(c, p, n0, s1, wgt, ctx);
    if (has2) {
      attn_load_head<T>(c, att_h + (size_t)n2 * HH, h0);
      attn_load_slot<T, NW, false>(c, R, P + (size_t)n2 * R * HH, s1);
    }
    if (has1) {
      wgt = attn_scores<T, NW>(c, p, n1, s2, h1, alpha);
      if (has2) attn_load_slot<T, NW, true>(c, R, V + (size_t)n2 * R * HH, s2);
      attn_context<T, SAFE, NW>(c, p, n1, s0, wgt, ctx);
    }
    if (has2) {
      wgt = attn_scores<T, NW>(c, p, n2, s1, h0, alpha);
      attn_context<T, SAFE, NW>(c, p, n2, s2, wgt, ctx);
    }
  } else {
    AttnSlot<T, NW> s0, s1;
    attn_load_head<T>(c, att_h + (size_t)n0 * HH, h0);
    attn_load_slot<T, NW, false>(c, R, P + (size_t)n0 * R * HH, s0);
    attn_load_slot<T, NW, true>(c, R, V + (size_t)n0 * R * HH, s1);
    float wgt = attn_scores<T, NW>(c, p, n0, s0, h0, alpha);
    if (has1) {
      attn_load_head<T>(c, att_h + (size_t)n1 * HH, h1);
      attn_load_slot<T, NW, false>(c, R, P + (size_t)n1 * R * HH, s0);
    }
    attn_context<T, SAFE, NW>(c, p, n0, s1, wgt, ctx);
    if (has1) {
      attn_load_slot<T, NW, true>(c, R, V + (size_t)n1 * R * HH, s1);
      wgt = attn_scores<T, NW>(c, p, n1, s0, h1, alpha);
      if (has2) {
        attn_load_head<T>(c, att_h + (size_t)n2 * HH, h0);
        attn_load_slot<T, NW, false>(c, R, P + (size_t)n2 * R * HH, s0);
      }
      attn_context<T, SAFE, NW>(c, p, n1, s1, wgt, ctx);
    }
    if (has2) {
      attn_load_slot<T, NW, true>(c, R, V + (size_t)n2 * R * HH, s1);
      wgt = attn_scores<T, NW>(c, p, n2, s0, h0, alpha);
      attn_context<T, SAFE, NW>(c, p, n2, s1, wgt, ctx);
    }
  }
}

// Round 5: the attention phase of the weight-stationary TRAINING kernel with ONE WAVE PER CAPTION ROW.  The four-waves-per-row form
// above is a chain of dependent stages per row (head load, two workgroup barriers, an LDS reduction between the waves, the
// softmax, another barrier, the context reduction): ~4 us of latency per row beside 2 us of arithmetic, three rows in a row for
// half of the workgroups (17.4 us) and two for the others, who then wait in the group barrier.  Here wave w of workgroup `rank`
// takes row rank + 32 w of the group by itself (waves 0..2; 80 rows on 96 of the group's 128 waves): no workgroup barrier, no
// cross-wave reduction, every wave one row -- the phase is one row's arithmetic on one SIMD (~8 us, 4.8 of it v_exp / v_rcp) and
// the same length for every workgroup.  Region rows arrive in chunks of 10 through two register buffers; att's first chunk is
// requested before the scores so that the context starts without a load latency.  The context is one running sum over the
// regions instead of four partial sums.
constexpr int AW_CR = 10;                             // regions per chunk of attn_phase_wave
// the first chunk of the wave's p_att rows: requested between the two halves of the group barrier in front of the phase
__device__ __forceinline__ void attn_wave_preload(const Ctx& c, const UicRnnFwdParams& p, uint4 (&q)[AW_CR]) {
  const int r = c.rank + PW * c.wave;
  if (r >= c.nrow) return;
  const bf16_t* P = (const bf16_t*)p.e_att + (size_t)(c.rbegin + r) * p.R * HH + c.lane * 8;
#pragma unroll
  for (int u = 0; u < AW_CR; ++u) {
    const u32x4 v = *(const u32x4*)(P + (unsigned)(min(u, p.R - 1) * HH));
    q[u] = make_uint4(v.x, v.y, v.z, v.w);
  }
}
template <bool SAFE>
__device__ __forceinline__ void attn_phase_wave(const Ctx& c, const UicRnnFwdParams& p, const float* att_h, float* alpha, bf16_t* ctx,
                                                const uint4 (&pre)[AW_CR]) {
  typedef bf16_t T;
  const int R = p.R;
  const int r = c.rank + PW * c.wave;                 // this wave's row of the group (uniform per wave)
  if (r >= c.nrow) return;
  const int n = c.rbegin + r;
  constexpr int CR = AW_CR, NCH = ATT_R / CR;
  static_assert(NCH * CR == ATT_R, "chunks cover the region slots");
  const T* P = (const T*)p.e_att + (size_t)n * R * HH + c.lane * 8;      // e^{2 p_att} (round 6, below)
  const T* V = (const T*)p.att + (size_t)n * R * HH + c.lane * 8;
  float* s_e = (float*)c.smem + 64 + c.wave * (4 * ATT_R);      // [R][4] row-of-16 partial scores of this wave's row (word 0: barrier flag)
  auto load_chunk = [&](const T* base, int ch, uint4 (&q)[CR]) {
#pragma unroll
    for (int u = 0; u < CR; ++u) {
      const int rr = min(ch * CR + u, R - 1);
      const u32x4 v = *(const u32x4*)(base + (unsigned)(rr * HH));
      q[u] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  const __amdgpu_buffer_rsrc_t rh = rsrc_of(att_h + (size_t)n * HH);
  const u32x4 h0 = bload<true>(rh, (unsigned)(c.lane * 32), 0), h1 = bload<true>(rh, (unsigned)(c.lane * 32 + 16), 0);
  // the region mask of this lane's region, requested HERE and without a branch (no mask: any readable word, ignored below): read
  // where it is used, under `if (mask)`, it was a load behind a run-time branch followed by s_waitcnt vmcnt(0) -- an exposed L2
  // round trip between the softmax and the context, with the context's first chunks held up behind it
  const float* mkp = p.mask ? p.mask + (size_t)n * p.ldmask : p.w_alpha;
  const float mkv = mkp[c.lane < R ? c.lane : 0];
  uint4 pa[2][CR], va[2][CR];
#pragma unroll
  for (int u = 0; u < CR; ++u) pa[0][u] = pre[u];
  load_chunk(P, 1, pa[1]);
  load_chunk(V, 0, va[0]);
  float ah[8], w[8];
  ah[0] = __uint_as_float(h0.x); ah[1] = __uint_as_float(h0.y); ah[2] = __uint_as_float(h0.z); ah[3] = __uint_as_float(h0.w);
  ah[4] = __uint_as_float(h1.x); ah[5] = __uint_as_float(h1.y); ah[6] = __uint_as_float(h1.z); ah[7] = __uint_as_float(h1.w);
  {
    const float4 w0 = *(const float4*)(p.w_alpha + c.lane * 8), w1 = *(const float4*)(p.w_alpha + c.lane * 8 + 4);
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
  }
  const float b_alpha = p.b_alpha ? p.b_alpha[0] : 0.f;
  // Round 6: w tanh(p + h) = w - 2 w rc,  rc = 1 / (1 + e^{2p} e^{2h}).  e^{2p} comes from memory (uic_exp2x2_launch, once per
  // training step), e^{2h} is eight v_exp per lane and step, and TWO regions share one reciprocal: rc_0 = x_1 / (x_0 x_1),
  // rc_1 = x_0 / (x_0 x_1) -- per element 0.5 v_rcp and three packed / plain VALU instructions where the direct form spent
  // v_exp + v_rcp + four (the phase is one row's arithmetic on one SIMD: 288 elements per lane).  Both factors are clamped to
  // 2^+-60 where they are made, so x <= 2^120 + 1 is finite; a product beyond f32 gives rc = 0 for both, which is what
  // tanh = 1 means (the partner's own x would have to be < 2^8 against 2^120 for that to matter).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 eh[8], wm2[8];
  float wsum = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float e = __builtin_amdgcn_exp2f(fminf(fmaxf(ah[j] * 2.8853900817779268f, -UIC_E2_CLAMP), UIC_E2_CLAMP));
    eh[j] = (f32x2){e, e};
    wm2[j] = (f32x2){-2.f * w[j], -2.f * w[j]};
    wsum += w[j];
  }
  const f32x2 one2 = {1.f, 1.f};
  static_assert(CR % 2 == 0, "regions are taken in pairs");
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
    for (int u = 0; u < CR; u += 2) {
      const int rr = ch * CR + u;
      float f0[8], f1[8];
      uic_unpack<T>(pa[ch & 1][u], f0);
      uic_unpack<T>(pa[ch & 1][u + 1], f1);
      f32x2 part = {wsum, wsum};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x2 x = __builtin_elementwise_fma((f32x2){f0[j], f1[j]}, eh[j], one2);
        const float inv = __builtin_amdgcn_rcpf(x.x * x.y);
        part = __builtin_elementwise_fma(wm2[j], (f32x2){inv, inv} * x.yx, part);
      }
      const float p0 = uic_row16_sum(part.x), p1 = uic_row16_sum(part.y);
      if (c.l15 == 0 && rr < R) s_e[rr * 4 + c.lq] = p0;
      if (c.l15 == 0 && rr + 1 < R) s_e[(rr + 1) * 4 + c.lq] = p1;
    }
    if (ch + 2 < NCH) load_chunk(P, ch + 2, pa[ch & 1]);
#ifdef UIC_AW_RESCHED   // (A/B build, tools/build_variant.sh: att' chunks 2 / 3 requested into the score buffers as those fall free -- the
                        //  context then never waits for them, but four buffers live at once cost the kernel its last 29 free registers:
                        //  256 + 256, i.e. a whole SIMD's register file per wave, and ANY foreign wave on a CU then keeps the launch from
                        //  becoming resident -- tools/queue_probe.py: one waiting wave on another stream, step 2.9 -> 5.9 ms.  Off.)
    else load_chunk(V, ch, pa[ch & 1]);
#endif
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes are done (LDS serves a wave's requests in order)
  float e = -INFINITY;
  if (c.lane < R) {
    const float4 qq = *(const float4*)(s_e + c.lane * 4);
    e = (qq.x + qq.y) + (qq.z + qq.w) + b_alpha;
  }
  const float mx = uic_wave_max(e);
  const float ex = c.lane < R ? __builtin_amdgcn_exp2f((e - mx) * 1.4426950408889634f) : 0.f;
  float wgt = ex * __builtin_amdgcn_rcpf(uic_wave_sum(ex));
  if (p.mask) {                                       // (wave-uniform; the value has been here since the top of the phase)
    wgt *= c.lane < R ? mkv : 0.f;
    wgt = wgt / uic_wave_sum(wgt);
  }
  if (c.lane < R) alpha[(unsigned)(n * R + c.lane)] = wgt;
  load_chunk(V, 1, va[1]);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
    for (int u = 0; u < CR; ++u) {
      const int rr = ch * CR + u;
      float al = __shfl(wgt, rr < ATT_R ? (rr < 64 ? rr : 0) : 0, 64);
      if (rr >= R) al = 0.f;
      float f[8];
#ifdef UIC_AW_RESCHED
      uic_unpack<T>(ch < 2 ? va[ch][u] : pa[ch & 1][u], f);
#else
      uic_unpack<T>(va[ch & 1][u], f);
#endif
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(al, f[j], acc[j]);   // (explicit: the SAFE and the XCD-local instantiation must round alike)
    }
#ifndef UIC_AW_RESCHED
    if (ch + 2 < NCH) load_chunk(V, ch + 2, va[ch & 1]);
#endif
  }
  T* o = ctx + (unsigned)(n * HH + c.lane * 8);
  if (!SAFE) {
    *(uint4*)o = uic_pack<T>(acc);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) st_x<SAFE>(o + j, acc[j]);
  }
}

template <typename T, bool SAFE>
__device__ __forceinline__ void run_steps(const UicRnnFwdParams& p, Ctx& c) {
  const int N = p.N;
  const size_t NH = (size_t)N * HH;
  const size_t rb = (size_t)c.rbegin * HH;
  const T* att_w_ih = (const T*)p.att_w_ih;
  const T* lang_w_ih = (const T*)p.lang_w_ih;
  unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * p.dbg_T + p.t0) * 16 : nullptr;
  for (int t = p.t0; t < p.t1; ++t) {
    T* h_att_prev = (T*)p.h_att + (size_t)t * NH;
    T* h_att_new = h_att_prev + NH;
    T* h_lang_prev = (T*)p.h_lang + (size_t)t * NH;
    T* h_lang_new = h_lang_prev + NH;
    // hipcc hoists every step-invariant per-lane value (row offsets, dropout hashes, 64-bit addresses of all four phases)
    // out of this loop and then spills them; making the lane coordinates opaque once per step keeps them recomputed instead
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
    asm volatile("" : "+s"(c.wave), "+s"(c.u0), "+s"(c.rbegin), "+s"(c.nrow), "+s"(c.MT));
    c.dbg = dbg;
    if (dbg && c.tid == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
    {  // att_lstm on cat([h_lang_prev, fc', xt]) with the fc' / xt share precomputed in gx / gfc (:431-434)
      const void* const As[2] = {h_lang_prev + rb, h_att_prev + rb};
      const void* const Bs[2] = {att_w_ih, p.att_w_hh};
      const int ldb[2] = {p.ld_att_ih, HH};
      const float* gx = p.gx + (size_t)t * N * 4 * HH;
      const float* gfc = p.gfc;
      auto pre = [&](unsigned idx4, unsigned) { return gx[idx4] + (gfc ? gfc[idx4] : 0.f); };
      lstm_phase<T, SAFE, 2>(c, As, Bs, ldb, pre, p.c_att + (size_t)t * NH, p.c_att + (size_t)(t + 1) * NH, h_att_new,
                             (T*)nullptr, p.gates1 ? (T*)p.gates1 + (size_t)t * N * 4 * HH : nullptr, N, 0.f, 0u, 0u);
    }
    if (dbg && c.tid == 0) dbg[1] = __builtin_amdgcn_s_memrealtime();
    if (!group_barrier(c)) return;
    if (dbg && c.tid == 0) dbg[2] = __builtin_amdgcn_s_memrealtime();
    float* att_h = p.att_h_all + (size_t)t * NH;
    h2att_phase<T, SAFE>(c, h_att_new + rb, (const T*)p.h2att_w, p.h2att_b, att_h);
    if (dbg && c.tid == 0) dbg[3] = __builtin_amdgcn_s_memrealtime();
    if (!group_barrier(c)) return;
    if (dbg && c.tid == 0) dbg[4] = __builtin_amdgcn_s_memrealtime();
    T* ctx = (T*)p.ctx_all + (size_t)t * NH;
    attn_phase<T, SAFE, NWAVE, 3>(c, p, att_h, p.alpha_all + (size_t)t * N * p.R, ctx);
    if (dbg && c.tid == 0) dbg[5] = __builtin_amdgcn_s_memrealtime();
    if (!group_barrier(c)) return;
    if (dbg && c.tid == 0) dbg[6] = __builtin_amdgcn_s_memrealtime();
    {  // lang_lstm on cat([att_res, h_att]) (:438-441) + the output dropout (:443)
      const void* const As[3] = {ctx + rb, h_att_new + rb, h_lang_prev + rb};
      const void* const Bs[3] = {lang_w_ih, lang_w_ih + HH, p.lang_w_hh};
      const int ldb[3] = {2 * HH, 2 * HH, HH};
      const float* b1 = p.lang_b_ih;
      const float* b2 = p.lang_b_hh;
      auto pre = [&](unsigned, unsigned col) { return (b1 ? b1[col] : 0.f) + (b2 ? b2[col] : 0.f); };
      lstm_phase<T, SAFE, 3>(c, As, Bs, ldb, pre, p.c_lang + (size_t)t * NH, p.c_lang + (size_t)(t + 1) * NH, h_lang_new,
                             p.hdrop_all ? (T*)p.hdrop_all + (size_t)t * NH : nullptr,
                             p.gates2 ? (T*)p.gates2 + (size_t)t * N * 4 * HH : nullptr, N, p.drop_p, p.seed,
                             UIC_SITE_OUT0 + (unsigned)t);
    }
    if (dbg && c.tid == 0) dbg[7] = __builtin_amdgcn_s_memrealtime();
    if (!group_barrier(c)) return;
    if (dbg) dbg += 16;
  }
}

template <typename T>
__global__ __launch_bounds__(NTH) void rnn_fwd_persist_kernel(const UicRnnFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  const int mode = setup_ctx(p, smem, c);
  if (mode == 0) return;
  if (mode == 2) run_steps<T, true>(p, c);
  else run_steps<T, false>(p, c);
}

// ---------------------------------------------------------------------------------------------------
// Weight-stationary bf16 variant.  A workgroup's slice of the recurrent weights is 336 KB in bf16 (64 gate columns of
// att_lstm x K 1024, 64 of lang_lstm x K 1536, 16 columns of h2att x K 512): streamed from the Infinity Cache every step
// it is 10.7 MB per XCD and step, as much as the activations.  Here the two LSTM slices are loaded ONCE per launch by a
// workgroup of FOUR waves with the whole 512-entry register file of their SIMD each:
//   * att_lstm's slice (128 KB) lives in LDS as MFMA B fragments [k-step][gate][lane][16 B]; wave w owns the 16-row tile w
//     of the group and walks all 32 k-steps itself, so the four gate accumulators of a (row, unit) end up in one lane and
//     the cell update needs no cross-wave reduction (the fifth tile of an 80-row group is split over the waves by K and
//     summed through LDS);
//   * lang_lstm's slice (192 KB) lives in REGISTERS, K split over the 4 waves (wave w holds k-steps w, w+4, ...: 48
//     fragments = 192 registers, MFMA operands only); the partial tiles are summed through two 16 KB LDS buffers, one
//     row tile per pass and one workgroup barrier per pass;
//   * h2att's slice (16 KB) is re-read every step beside the activations (4 MB per step chip-wide).
// The c state of both cells stays in the registers of the lanes that update it.
#ifndef WS_EARLY_PRE
#define WS_EARLY_PRE 0
#endif
#ifndef WS_P1_CHUNK
#define WS_P1_CHUNK 8
#endif
#ifndef WS_P2_ALL
#define WS_P2_ALL 0
#endif
#ifndef WS_ATT_SLOTS
#define WS_ATT_SLOTS 2
#endif
#ifndef WS_P4_DEPTH
#define WS_P4_DEPTH 2
#endif
constexpr int WS_NW = 4;
constexpr int WS_NTH = WS_NW * 64;
constexpr int WS_W1_BYTES = 32 * 4 * 1024;
constexpr int WS_SCR_BYTES = 32 * 1024;
constexpr int WS_LDS_BYTES = WS_W1_BYTES + WS_SCR_BYTES;

__device__ __forceinline__ f32x4 mma_bf16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// nn.LSTMCell's pointwise part for the 4 rows x 1 unit a lane holds in the D layout of four gate accumulators
template <bool SAFE>
__device__ __forceinline__ void ws_cell(const Ctx& c, int tile, const f32x4 (&s)[4], const float (&pv)[4][4], float (&cst)[4],
                                        float* c_out, bf16_t* h_out, bf16_t* h_drop, bf16_t* gates_out, float drop_p,
                                        unsigned seed, unsigned site) {
  const unsigned u = (unsigned)(c.u0 + c.l15);
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = 16 * tile + 4 * c.lq + r;
    const float gi = uic_sigmoid_t<bf16_t>(s[0][r] + pv[r][0]);
    const float gf = uic_sigmoid_t<bf16_t>(s[1][r] + pv[r][1]);
    const float gg = uic_tanh<bf16_t>(s[2][r] + pv[r][2]);
    const float go = uic_sigmoid_t<bf16_t>(s[3][r] + pv[r][3]);
    const float cn = gf * cst[r] + gi * gg;
    const float h = go * uic_tanh<bf16_t>(cn);
    cst[r] = cn;
    if (rr < c.nrow) {
      const unsigned nn = (unsigned)((c.rbegin + rr) * HH);
      const unsigned o = nn + u;
      c_out[o] = cn;
      st_x<SAFE>(h_out + o, h);
      if (h_drop) {
        float hd = h;
        if (drop_p > 0.f) hd *= uic_drop_scale(seed, site, o, drop_p, inv_keep);
        h_drop[o] = (bf16_t)hd;
      }
      if (gates_out) {
        const unsigned og = 4u * nn + u;
        __builtin_nontemporal_store((bf16_t)gi, gates_out + og);
        __builtin_nontemporal_store((bf16_t)gf, gates_out + og + HH);
        __builtin_nontemporal_store((bf16_t)gg, gates_out + og + 2 * HH);
        __builtin_nontemporal_store((bf16_t)go, gates_out + og + 3 * HH);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Decode mode of the weight-stationary kernel (AttModel._sample, P/models/AttModel.py:198-253: greedy or multinomial, the
// sampling pass and the greedy baseline of the self-critical step, P/trainer.py:167-171): a step's input token exists only
// once the previous step has picked it, so three more pieces run inside the launch, all per row group:
//   dec_xt_gemm   the token's share of att_lstm's gates, dropout(relu(embed[token])) W_x^T, for the rows whose cell update
//                 this wave finishes -- A fragments built straight from the f32 embedding table, the result stays in the
//                 registers the cell update reads (the teacher-forced kernel gets it from the batched input GEMM);
//   dec_logits    logits of the group's rows for this workgroup's 1/32 of the vocabulary (W_logit streamed from L2 / the
//                 Infinity Cache, A = the rows' dropped h_lang), written out, plus per-row (max, sum exp, arg max) partials;
//   dec_sample    after a group barrier: every workgroup combines the 32 partials of every row (same order everywhere, so
//                 all agree); greedy: the arg max; multinomial: the workgroup whose vocabulary range holds the inverse-CDF
//                 target scans its <= 320 logits of that row.  The row's writer stores token, log-prob and finished flag.
// Three more group barriers per step than the teacher-forced loop.
constexpr int DEC_NJ = 5;                 // vocabulary tiles (16 columns) per wave: 32 workgroups x 4 waves x 5 x 16 = 10240 >= V1

template <bool SAFE> __device__ __forceinline__ void st_xi(int* p, int v) {
  if (SAFE) __hip_atomic_store(p, v, RLX_AGENT);
  else *p = v;
}
__device__ __forceinline__ int ld_xi(const int* p) { return __hip_atomic_load(p, RLX_AGENT); }       // (past the vector L1)
__device__ __forceinline__ float ld_xf(const float* p) { return __hip_atomic_load(p, RLX_AGENT); }

// (m, s, a) <- the combination of two partial (max, sum of exp(x - max), arg max) triples; `o` lies at HIGHER column indices
__device__ __forceinline__ void dec_merge(float& m, float& s, int& a, float om, float os, int oa) {
  const float m2 = fmaxf(m, om);
  const float e1 = m > -INFINITY ? __expf(m - m2) : 0.f, e2 = om > -INFINITY ? __expf(om - m2) : 0.f;
  s = s * e1 + os * e2;
  if (om > m || (om == m && oa < a)) a = oa;
  m = m2;
}

// xg[r][g] / xg5[g]: the input token's share of att_lstm's gate pre-activations of step `t`, in the layout ws_cell reads
// (own tile: rows 16 wave + 4 lq + r; fifth tile: row 64 + 4 lq + wave), unit u0 + l15, gate g.
// K is split over the waves (wave w: k-steps 4w .. 4w+3 of EVERY row tile), so a lane requests its 16 W_x fragments and its
// <= 20 embedding-row fragments all at once -- the table (relu(embed) in bf16, 10 MB, random rows) comes from the Infinity
// Cache, W_x from L2,
// one round trip each -- and the partial tiles are summed through LDS, one row tile per pass (as lang_lstm does).
__device__ __forceinline__ void dec_xt_gemm(const UicRnnFwdParams& p, const Ctx& c, int t, float (&xg)[4][4], float (&xg5)[4], const int* tok_lds = nullptr) {
  const float dp = p.dec_xt_drop, inv_keep = dp > 0.f ? 1.f / (1.f - dp) : 1.f;
  const __amdgpu_buffer_rsrc_t r_xw = rsrc_of(p.dec_xw);
  u32x4 fb[4][4];                                  // [k-step of this wave][gate]
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      fb[k][g] = bload<false>(r_xw, (unsigned)(((g * HH + c.u0 + c.l15) * p.dec_ld_xw + c.lq * 8) * 2), (unsigned)((4 * c.wave + k) * 64));
  // (tok_lds: the group's tokens as dec_sample left them in LDS -- greedy / forced, where every workgroup knows them all)
  int tok[MT_MAX], rowi[MT_MAX];
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i) {
    int r = 16 * i + c.l15;
    r = r < c.nrow ? r : c.nrow - 1;
    rowi[i] = r;
    int tk = i < c.MT ? (tok_lds ? tok_lds[r] : ld_xi(p.dec_tok + c.rbegin + r)) : 0;
    tok[i] = tk < 0 || tk >= p.dec_V1 ? 0 : tk;
  }
  if (tok_lds) __syncthreads();                    // (the token list lives in the scratch the partial tiles go to: every wave has read it)
  const unsigned ko = (unsigned)(c.wave * 128 + c.lq * 8);          // first K element of this lane's fragments
  // every fragment of the lane (<= 20 x 16 bytes of relu(embed) in bf16, made once per launch by the host side) in flight at once
  u32x4 fa[MT_MAX][4];
  const bf16_t* er = (const bf16_t*)p.dec_embed_relu;
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i)
    if (i < c.MT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) fa[i][k] = *(const u32x4*)(er + (size_t)tok[i] * HH + ko + k * 32);
    }
  bf16_t* xo = (bf16_t*)p.dec_xt_all;
  if (dp > 0.f || xo) {
#pragma unroll
    for (int i = 0; i < MT_MAX; ++i)
      if (i < c.MT) {
        // dropout index of element e of row n at step t: (t N + n) E + e, the index uic_embed_fwd_launch uses for the training layout
        const unsigned idx = (unsigned)(((size_t)t * p.N + c.rbegin + rowi[i]) * HH) + ko;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (dp > 0.f) {
            float f[8];
            uic_unpack<bf16_t>(__builtin_bit_cast(uint4, fa[i][k]), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= uic_drop_scale(p.seed, UIC_SITE_EMBED, idx + (unsigned)(k * 32 + j), dp, inv_keep);
            fa[i][k] = u32x4{uic_pack_bf16x2(f[0], f[1]), uic_pack_bf16x2(f[2], f[3]), uic_pack_bf16x2(f[4], f[5]), uic_pack_bf16x2(f[6], f[7])};
          }
          // the embedded inputs, for a backward pass: workgroup `rank` writes k-step `rank` (of the 16) of every row
          if (xo && c.rank == 4 * c.wave + k && 16 * i + c.l15 < c.nrow)
            *(u32x4*)(xo + ((size_t)t * p.N + c.rbegin + rowi[i]) * HH + ko + k * 32) = fa[i][k];
        }
      }
  }
  f32x4* scr = (f32x4*)c.smem;                     // two 16 KB halves, as in the lang_lstm phase
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i)
    if (i < c.MT) {
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mma_bf16(fa[i][k], fb[k][g], acc[g]);
      f32x4* half = scr + (i & 1) * (WS_NW * 4 * 64);
#pragma unroll
      for (int g = 0; g < 4; ++g) half[(c.wave * 4 + g) * 64 + c.lane] = acc[g];
      __syncthreads();
      if (i < WS_NW) {
        if (c.wave == i) {                         // the tile's owner: all four accumulator components of its lanes
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v = half[(0 * 4 + g) * 64 + c.lane];
#pragma unroll
            for (int w = 1; w < WS_NW; ++w) v += half[(w * 4 + g) * 64 + c.lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) xg[r][g] = v[r];
          }
        }
      } else {                                     // the fifth tile: wave w finishes row 4 lq + w
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < WS_NW; ++w) v += ((const float*)(half + (w * 4 + g) * 64 + c.lane))[c.wave];
          xg5[g] = v;
        }
      }
    }
  __syncthreads();                                 // (the scratch is the barrier flag's home, and the next pass's)
}

// LDS layout of the decode tail inside the 32 KB scratch (word 0 is the barrier flag)
constexpr int DEC_OFF_PM = 256, DEC_OFF_PS = DEC_OFF_PM + WS_NW * 80 * 4, DEC_OFF_PA = DEC_OFF_PS + WS_NW * 80 * 4;
constexpr int DEC_OFF_CNT = DEC_OFF_PA + WS_NW * 80 * 4, DEC_OFF_LIST = DEC_OFF_CNT + 16;   // list: 80 x {row, rt, M, lse}

// att_lstm's B fragments of k-steps wave + 4 jj, jj in [jj0, jj1), into the LDS image (the launch's one-time load, and the
// reload of the part dec_logits borrows)
// (vwave: the wave whose share of the image this wave loads -- normally its own)
__device__ __forceinline__ void ws_load_w1(const UicRnnFwdParams& p, const Ctx& c, char* lds, int jj0, int jj1, int vwave) {
  const unsigned bl = (unsigned)(c.lq * 8);
  const __amdgpu_buffer_rsrc_t r_a_ih = rsrc_of(p.att_w_ih), r_a_hh = rsrc_of(p.att_w_hh);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    if (jj < jj0 || jj >= jj1) continue;
    const int s = vwave + 4 * jj;                 // k-step of [h_lang_prev | h_att_prev]
    const unsigned kk = (unsigned)((s & 15) * 32);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned row = (unsigned)(g * HH + c.u0 + c.l15);
      const u32x4 v = jj < 4 ? bload<false>(r_a_ih, (row * (unsigned)p.ld_att_ih + kk + bl) * 2u, 0)
                             : bload<false>(r_a_hh, (row * (unsigned)HH + kk + bl) * 2u, 0);
      ((u32x4*)lds)[(s * 4 + g) * 64 + c.lane] = v;
    }
  }
}

// The group's dropped h_lang rows (80 x 512 bf16 = 80 KB) are every wave's A operand: they are staged ONCE per step into LDS
// as MFMA fragments, over the first 80 KB of att_lstm's weight image (k-steps 0..19), which is reloaded from L2 afterwards --
// the weights of W_logit, the operand that comes from the Infinity Cache, can then be requested DEC_LD k-steps ahead.
constexpr int DEC_LD = 3;
// arrived: the caller has ARRIVED at the group barrier behind lang_lstm but not waited yet -- W_logit's first fragments are
// requested before the wait (they do not depend on the exchange).  Returns false after a barrier timeout.
template <bool SAFE>
__device__ __forceinline__ bool dec_logits(const UicRnnFwdParams& p, Ctx& c, int t, char* lds) {
  const int V1 = p.dec_V1;
  const int ntile_all = (V1 + 15) >> 4, per = (ntile_all + PW - 1) / PW;
  const int tile0 = c.rank * per;
  const int ntile = min(per, max(0, ntile_all - tile0));
  const bf16_t* hd = (const bf16_t*)p.hdrop_all + ((size_t)t * p.N + c.rbegin) * HH;
  const __amdgpu_buffer_rsrc_t r_a = rsrc_of(hd), r_w = rsrc_of(p.dec_logit_w);
  u32x4* la = (u32x4*)lds;                         // [k-step 16][row tile 5][lane 64]
  unsigned woff[DEC_NJ];
  int col0[DEC_NJ];                                // first column of the wave's tile j (>= V1: no tile)
#pragma unroll
  for (int j = 0; j < DEC_NJ; ++j) {
    const bool tv = c.wave + WS_NW * j < ntile;
    col0[j] = tv ? 16 * (tile0 + c.wave + WS_NW * j) : V1;
    const int wr = col0[j] + c.l15 < V1 ? col0[j] + c.l15 : V1 - 1;
    woff[j] = (unsigned)((wr * HH + c.lq * 8) * 2);
  }
  u32x4 fb[DEC_LD][DEC_NJ];
  auto loadb = [&](int buf, int ks) {
#pragma unroll
    for (int j = 0; j < DEC_NJ; ++j) fb[buf][j] = bload<false>(r_w, woff[j], (unsigned)(ks * 64));
  };
  const int ks0 = c.rank & 15;                     // de-phased walk over K, as in the recurrence's GEMMs
#pragma unroll
  for (int q = 0; q < DEC_LD - 1; ++q) loadb(q, (q + ks0) & 15);
  if (!group_wait(c, (int*)c.smem)) return false;
  if (c.dbg && c.tid == 0) c.dbg[15] = __builtin_amdgcn_s_memrealtime();
  {
    // fragment f = ks * 5 + i is staged by wave f & 3: 20 loads per lane, all in flight
    u32x4 st[20];
#pragma unroll
    for (int q = 0; q < 20; ++q) {
      const int f = c.wave + 4 * q, ks = f / MT_MAX, i = f - ks * MT_MAX;
      int r = 16 * i + c.l15;
      r = r < c.nrow ? r : c.nrow - 1;
      st[q] = bload<true>(r_a, (unsigned)((r * HH + c.lq * 8) * 2), (unsigned)(ks * 64));
    }
#pragma unroll
    for (int q = 0; q < 20; ++q) la[(c.wave + 4 * q) * 64 + c.lane] = st[q];
  }
  float4 bias[DEC_NJ];
#pragma unroll
  for (int j = 0; j < DEC_NJ; ++j) {
    const int cb = col0[j] + 4 * c.lq;
    bias[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.dec_logit_b && cb < V1) {
      if (cb + 3 < V1) bias[j] = *(const float4*)(p.dec_logit_b + cb);
      else { bias[j].x = p.dec_logit_b[cb]; bias[j].y = cb + 1 < V1 ? p.dec_logit_b[cb + 1] : 0.f; bias[j].z = cb + 2 < V1 ? p.dec_logit_b[cb + 2] : 0.f; }
    }
  }
  f32x4 acc[MT_MAX][DEC_NJ];
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i)
#pragma unroll
    for (int j = 0; j < DEC_NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                 // the A image is complete
  if (c.dbg && c.tid == 0) c.dbg[11] = __builtin_amdgcn_s_memrealtime();
  // W_logit's fragment is the MFMA's FIRST operand: the lane then holds 4 consecutive vocabulary columns (4 lq + r) of ONE
  // caption row (l15) per tile -- one 16-byte store per tile, and a row's maximum / sum are mostly in-lane
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int ks = (q + ks0) & 15;
    if (q + DEC_LD - 1 < 16) loadb((q + DEC_LD - 1) % DEC_LD, (q + DEC_LD - 1 + ks0) & 15);
    u32x4 fa[MT_MAX];
#pragma unroll
    for (int i = 0; i < MT_MAX; ++i)
      if (i < c.MT) fa[i] = la[(ks * MT_MAX + i) * 64 + c.lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT_MAX; ++i)
      if (i < c.MT) {
#pragma unroll
        for (int j = 0; j < DEC_NJ; ++j) acc[i][j] = mma_bf16(fb[q % DEC_LD][j], fa[i], acc[i][j]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (c.dbg && c.tid == 0) c.dbg[12] = __builtin_amdgcn_s_memrealtime();
  // bias, store, per-row partials
  const bool store = p.dec_logits_step != 0 || !p.dec_sample_max;     // kept for a backward pass, or read back by the draw
  float* lg = p.dec_logits + (size_t)t * p.dec_logits_step;
  float* pm = (float*)(c.smem + DEC_OFF_PM);
  float* ps = (float*)(c.smem + DEC_OFF_PS);
  int* pa = (int*)(c.smem + DEC_OFF_PA);
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i)
    if (i < c.MT) {
      const int row = 16 * i + c.l15;
      const bool rv = row < c.nrow;
      float m = -INFINITY;
      int a = 0x7fffffff;
      float v[DEC_NJ][4];
#pragma unroll
      for (int j = 0; j < DEC_NJ; ++j) {
        const int cb = col0[j] + 4 * c.lq;
        v[j][0] = acc[i][j][0] + bias[j].x; v[j][1] = acc[i][j][1] + bias[j].y;
        v[j][2] = acc[i][j][2] + bias[j].z; v[j][3] = acc[i][j][3] + bias[j].w;
        if (store && rv && cb < V1) {
          float* o = lg + (size_t)(c.rbegin + row) * p.dec_V1p + cb;
          if (!SAFE && cb + 3 < V1) *(float4*)o = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (cb + r < V1) st_x<SAFE>(o + r, v[j][r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (cb + r >= V1) v[j][r] = -INFINITY;
          if (v[j][r] > m) { m = v[j][r]; a = cb + r; }           // (columns ascend with j and r inside a lane)
        }
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < DEC_NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += v[j][r] > -INFINITY ? __expf(v[j][r] - m) : 0.f;
      // the other three column quarters of the same tiles live in lanes l15 + 16, + 32, + 48
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float om = __shfl_xor(m, o, 64), os = __shfl_xor(sum, o, 64);
        const int oa = __shfl_xor(a, o, 64);
        dec_merge(m, sum, a, om, os, oa);
      }
      if (c.lq == 0 && rv) { pm[c.wave * 80 + row] = m; ps[c.wave * 80 + row] = sum; pa[c.wave * 80 + row] = a; }
    }
  if (c.dbg && c.tid == 0) c.dbg[14] = __builtin_amdgcn_s_memrealtime();
  __syncthreads();                                 // partials in LDS; every wave is through with the A image
  if (c.tid < c.nrow) {
    const int row = c.tid;
    float m = pm[row], sum = ps[row];
    int a = pa[row];
#pragma unroll
    for (int w = 1; w < WS_NW; ++w) dec_merge(m, sum, a, pm[w * 80 + row], ps[w * 80 + row], pa[w * 80 + row]);
    float* o = p.dec_part + ((size_t)(c.rbegin + row) * PW + c.rank) * 4;
    st_x<SAFE>(o, m); st_x<SAFE>(o + 1, sum); st_x<SAFE>(o + 2, __int_as_float(a));
  }
  return true;
}

template <bool SAFE>
__device__ __forceinline__ void dec_sample(const UicRnnFwdParams& p, const Ctx& c, int t, char* lds) {
  const int V1 = p.dec_V1;
  const int ntile_all = (V1 + 15) >> 4, per = (ntile_all + PW - 1) / PW;
  int* s_cnt = (int*)(c.smem + DEC_OFF_CNT);
  float* s_list = (float*)(c.smem + DEC_OFF_LIST);
  const float* lg = p.dec_logits + (size_t)t * p.dec_logits_step;
  if (c.tid == 0) *s_cnt = 0;
  __syncthreads();
  // att_lstm's weight fragments of k-steps 0..19 go back into the LDS image dec_logits borrowed (from L2): by the two waves
  // that have no row to combine below (a group has at most 80 rows = threads 0..79), beside the two that do
  if (c.wave >= 2) {
    const unsigned bl = (unsigned)(c.lq * 8);
    const __amdgpu_buffer_rsrc_t r_a_ih = rsrc_of(p.att_w_ih), r_a_hh = rsrc_of(p.att_w_hh);
    u32x4 v[40];                                   // all in flight: one L2 round trip
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int vw = c.wave - 2 * half;            // the wave whose share of the image this is
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) {
        const unsigned kk = (unsigned)(((vw + 4 * jj) & 15) * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned row = (unsigned)(g * HH + c.u0 + c.l15);
          v[half * 20 + jj * 4 + g] = jj < 4 ? bload<false>(r_a_ih, (row * (unsigned)p.ld_att_ih + kk + bl) * 2u, 0)
                                             : bload<false>(r_a_hh, (row * (unsigned)HH + kk + bl) * 2u, 0);
        }
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int jj = 0; jj < 5; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) ((u32x4*)lds)[((c.wave - 2 * half + 4 * jj) * 4 + g) * 64 + c.lane] = v[half * 20 + jj * 4 + g];
  }
  auto finish = [&](int n, int choice, float lp) {   // the row's writer (AttModel.py:240-249)
    int unf = choice > 0;
    if (t > p.t0) unf = unf && ld_xi(p.dec_unf + n) != 0;
    const int tok = unf ? choice : 0;
    st_xi<SAFE>(p.dec_unf + n, unf);
    st_xi<SAFE>(p.dec_tok + n, tok);
    p.dec_seq[(size_t)n * p.dec_ld_out + t] = tok;
    p.dec_seq_logp[(size_t)n * p.dec_ld_out + t] = lp;
  };
  if (c.tid < c.nrow) {
    const int row = c.tid, n = c.rbegin + row;
    const __amdgpu_buffer_rsrc_t r_p = rsrc_of(p.dec_part);       // (one descriptor for the wave: the row is the lane offset)
    const unsigned poff = (unsigned)n * (unsigned)(PW * 16);
    float M = -INFINITY;
    int best = 0;
    u32x4 q[PW];
#pragma unroll
    for (int w = 0; w < PW; ++w) q[w] = bload<true>(r_p, poff, (unsigned)(w * 16));
#pragma unroll
    for (int w = 0; w < PW; ++w) {
      const float mw = __uint_as_float(q[w].x);
      if (mw > M) { M = mw; best = (int)q[w].z; }       // ranks own increasing column ranges: the first maximum is the lowest index
    }
    float Z = 0.f;
#pragma unroll
    for (int w = 0; w < PW; ++w) {
      const float mw = __uint_as_float(q[w].x);
      Z += mw > -INFINITY ? __uint_as_float(q[w].y) * __expf(mw - M) : 0.f;
    }
    const float lse = M + __logf(Z);
    const bool writer = (row & (PW - 1)) == c.rank;
    if (p.dec_sample_max || p.dec_forced) {
      // every workgroup knows the choice of every row of the group: the next step's tokens go to LDS (no exchange, no barrier)
      int ch = best;
      if (!p.dec_sample_max) {
        ch = (int)p.dec_forced[(size_t)n * p.dec_ld_forced + t];
        ch = ch < 0 || ch >= V1 ? 0 : ch;
      }
      int unf = ch > 0;
      if (t > p.t0) unf = unf && ld_xi(p.dec_unf + n) != 0;       // (written by the row's writer one step -- several barriers -- ago)
      ((int*)s_list)[row] = unf ? ch : 0;
      if (writer) finish(n, ch, p.dec_sample_max ? M - lse : ld_xf(lg + (size_t)n * p.dec_V1p + ch) - lse);
    } else {
      // inverse-CDF target in units of exp(. - M); the owner is the first workgroup whose cumulative mass exceeds it
      unsigned x = (unsigned)n * 0x9E3779B1u ^ (p.dec_draw_seed + (unsigned)t * 0x85EBCA77u);
      x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
      const float target = (float)(x >> 8) * (1.0f / 16777216.0f) * Z;
      float cum = 0.f, rt = INFINITY;
      int owner = -1, lastmass = 0;
#pragma unroll
      for (int w = 0; w < PW; ++w) {
        const float mw = __uint_as_float(q[w].x);
        const float sw = mw > -INFINITY ? __uint_as_float(q[w].y) * __expf(mw - M) : 0.f;
        if (sw > 0.f) lastmass = w;
        if (owner < 0) {
          if (cum + sw > target) { owner = w; rt = target - cum; }
          else cum += sw;
        }
      }
      if (owner < 0) owner = lastmass;                // rounding left the target at or beyond the total mass: the last token with mass
      if (owner == c.rank) {
        const int k = atomicAdd(s_cnt, 1);
        s_list[4 * k] = __int_as_float(row); s_list[4 * k + 1] = rt; s_list[4 * k + 2] = M; s_list[4 * k + 3] = lse;
      }
    }
  }
  if (p.dec_sample_max || p.dec_forced) return;
  __syncthreads();
  const int cnt = *s_cnt;
  const int c_lo = 16 * c.rank * per, c_hi = min(V1, c_lo + 16 * per);
  for (int k = c.wave; k < cnt; k += WS_NW) {          // one wave per owned row: its <= 320 logits, 64 per round
    const int row = __float_as_int(s_list[4 * k]), n = c.rbegin + row;
    const float rt = s_list[4 * k + 1], M = s_list[4 * k + 2], lse = s_list[4 * k + 3];
    const float* lr = lg + (size_t)n * p.dec_V1p;
    float lv[DEC_NJ];
#pragma unroll
    for (int j = 0; j < DEC_NJ; ++j) {
      const int v = c_lo + 64 * j + c.lane;
      lv[j] = v < c_hi ? ld_xf(lr + v) : -INFINITY;
    }
    float cum = 0.f, lp = 0.f;
    int pick = -1, last = -1;
    float lastl = 0.f;
#pragma unroll
    for (int j = 0; j < DEC_NJ; ++j) {
      const float e = lv[j] > -INFINITY ? __expf(lv[j] - M) : 0.f;
      float incl = e;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(incl, o, 64);
        if (c.lane >= o) incl += up;
      }
      const unsigned long long hit = __ballot(pick < 0 && cum + incl > rt && e > 0.f);
      const unsigned long long mass = __ballot(e > 0.f);
      if (mass) {
        const int hl = 63 - __clzll((long long)mass);
        last = c_lo + 64 * j + hl;
        lastl = __shfl(lv[j], hl, 64);
      }
      if (hit && pick < 0) {
        const int fl = __ffsll((long long)hit) - 1;
        pick = c_lo + 64 * j + fl;
        lp = __shfl(lv[j], fl, 64) - lse;
      }
      cum += __shfl(incl, 63, 64);
    }
    if (pick < 0) { pick = last < 0 ? 0 : last; lp = (last < 0 ? ld_xf(lr) : lastl) - lse; }
    if (c.lane == 0) finish(n, pick, lp);
  }
}

template <bool SAFE, bool DEC>
__device__ __forceinline__ void ws_run(const UicRnnFwdParams& p, Ctx& c, char* lds) {
  typedef bf16_t T;
  const int N = p.N;
  const size_t NH = (size_t)N * HH;
  const size_t rb = (size_t)c.rbegin * HH;
  const u32x4* w1 = (const u32x4*)lds;             // [k-step 32][gate 4][lane 64]
  f32x4* scr = (f32x4*)c.smem;                     // 32 KB: two 16 KB halves
  // ---- one-time weight load
  // De-phasing: all 32 workgroups of a group read the SAME activation rows in every GEMM phase.  Walking them in the same
  // order at the same time puts every request of the XCD on the same few L2 channels, so each workgroup starts its walk
  // over K (and over the row tiles) at an offset of its own, `rank`; the stationary fragments are laid out to match.
  const int j0 = __builtin_amdgcn_readfirstlane(c.rank % 12);
  const int tr = __builtin_amdgcn_readfirstlane(c.rank % c.MT);   // the row tiles are walked starting at tile tr
  u32x4 w2[12][4];                                 // slot j: k-step wave + 4 m of [att_res | h_att | h_lang_prev], m as below
  {
    const unsigned bl = (unsigned)(c.lq * 8);
    // (one descriptor for weight_ih and weight_hh of lang_lstm: which of them slot j reads is a select of a 32-bit offset)
    const char* wlo = (const char*)p.lang_w_ih < (const char*)p.lang_w_hh ? (const char*)p.lang_w_ih : (const char*)p.lang_w_hh;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)wlo, 0, -1, 0x00020000);
    const unsigned o_ih = (unsigned)((const char*)p.lang_w_ih - wlo), o_hh = (unsigned)((const char*)p.lang_w_hh - wlo);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      // slots 0-3: h_att, 4-7: h_lang_prev, 8-11: att_res -- the segment that the attention phase produces comes LAST, so that a
      // row tile's first eight fragments can be requested (and multiplied) before the group barrier behind the attention is through;
      // inside a segment the four k-steps are walked from an offset of the workgroup's own (j0)
      const int sg = (j >> 2) == 2 ? 0 : (j >> 2) + 1;   // 0: att_res, 1: h_att, 2: h_lang_prev   (16 k-steps each)
      const int m = sg * 4 + ((j + j0) & 3);
      const unsigned kk = (unsigned)(((m & 3) * 4 + c.wave) * 32);
      const unsigned ld = sg < 2 ? 2u * HH : (unsigned)HH;
      const unsigned so = (sg < 2 ? o_ih + (unsigned)(sg * HH) * 2u : o_hh) + kk * 2u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned row = (unsigned)(g * HH + c.u0 + c.l15);
        w2[j][g] = bload<false>(r_w, (row * ld + bl) * 2u, so);
      }
    }
    ws_load_w1(p, c, lds, 0, 8, c.wave);
  }
  // lang_lstm's bias for the unit of this lane (tile owners: wave w owns row tile w, wave 0 also tile 4)
  float pb[4][4];
  {
    const unsigned u = (unsigned)(c.u0 + c.l15);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float b = (p.lang_b_ih ? p.lang_b_ih[g * HH + u] : 0.f) + (p.lang_b_hh ? p.lang_b_hh[g * HH + u] : 0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) pb[r][g] = b;
    }
  }
  // The recurrence-independent share of att_lstm's gate pre-activations (xt and fc' terms and both biases, one f32 tensor
  // made by the batched input GEMM) and the cell state of step `ts`, for the 4 rows x 1 unit this lane finishes in row tile
  // `tile`.  They come from HBM / the Infinity Cache.
  // The caption row's fc' share (Gfc, when the host keeps it out of gx) is the same at every step: read once, kept in registers.
  float gfo[4][4], gf5[4];
  {
    const unsigned u = (unsigned)(c.u0 + c.l15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * c.wave + 4 * c.lq + r;
      const unsigned n1 = (unsigned)((c.rbegin + (rr < c.nrow ? rr : c.nrow - 1)) * HH);
#pragma unroll
      for (int g = 0; g < 4; ++g) gfo[r][g] = p.gfc ? p.gfc[4u * n1 + (unsigned)(g * HH) + u] : 0.f;
    }
    const int rr = 16 * WS_NW + 4 * c.lq + c.wave;
    const unsigned n1 = (unsigned)((c.rbegin + (rr < c.nrow ? rr : c.nrow - 1)) * HH);
#pragma unroll
    for (int g = 0; g < 4; ++g) gf5[g] = p.gfc ? p.gfc[4u * n1 + (unsigned)(g * HH) + u] : 0.f;
  }
  // decode mode: the input token's share of att_lstm's gates of the coming step, made by dec_xt_gemm (the teacher-forced loop
  // reads it from gx)
  float xg[4][4], xg5[4];
  if (DEC) dec_xt_gemm(p, c, p.t0, xg, xg5);
  auto load_pre = [&](int ts, int tile, float (&pv)[4][4], float (&cp)[4]) {     // (tile == this wave's own tile)
    const float* gx = p.gx + (size_t)ts * N * 4 * HH;
    const float* c_prev = p.c_att + (size_t)ts * NH;
    const unsigned u = (unsigned)(c.u0 + c.l15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * tile + 4 * c.lq + r;
      const unsigned n1 = (unsigned)((c.rbegin + (rr < c.nrow ? rr : c.nrow - 1)) * HH);
      cp[r] = c_prev[n1 + u];
#pragma unroll
      for (int g = 0; g < 4; ++g) pv[r][g] = DEC ? xg[r][g] : gx[4u * n1 + (unsigned)(g * HH) + u];
    }
  };
  // fifth tile: its cell update is shared by the four waves, wave w takes row 4 lq + w of every 4-row group
  auto load_pre5 = [&](int ts, float (&pv)[4], float& cp) {
    const float* gx = p.gx + (size_t)ts * N * 4 * HH;
    const unsigned u = (unsigned)(c.u0 + c.l15);
    const int rr = 16 * WS_NW + 4 * c.lq + c.wave;
    const unsigned n1 = (unsigned)((c.rbegin + (rr < c.nrow ? rr : c.nrow - 1)) * HH);
    cp = p.c_att[(size_t)ts * NH + n1 + u];
#pragma unroll
    for (int g = 0; g < 4; ++g) pv[g] = DEC ? xg5[g] : gx[4u * n1 + (unsigned)(g * HH) + u];
  };
#if WS_EARLY_PRE
  float pvn[4][4], cpn[4];                          // of the NEXT att_lstm phase, own tile (tile = wave): requested one phase early
  load_pre(p.t0, c.wave, pvn, cpn);
#endif
  __syncthreads();                                  // the W1 image is complete
  // att_lstm's gate accumulators of this wave's own row tile, and the half of its K that needs no exchange at the step boundary:
  // h_att_prev of step t + 1 is step t's h_att_new, which every workgroup has had since the barrier behind att_lstm -- so its 16
  // k-steps are multiplied between the two halves of the step's LAST group barrier (after this workgroup's arrival, before it
  // looks for the others'), i.e. in time that was idle; what is left behind the barrier is the h_lang_prev half.
  f32x4 acc1[4];
  const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.xbase, 0, -1, 0x00020000);
  auto lstm1_early = [&](unsigned o_hap_t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) acc1[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c.wave < c.MT) {
      int ar = 16 * c.wave + c.l15;
      ar = ar < c.nrow ? ar : c.nrow - 1;
      const unsigned aoff = (unsigned)((ar * HH + c.lq * 8) * 2);
      const int h0 = c.rank & 15;                   // each workgroup starts its walk over the half at an offset of its own
      u32x4 fa[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) fa[q] = bload<true>(rx0, aoff, o_hap_t + (unsigned)(((q + h0) & 15) * 64));
      u32x4 fb[2][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) fb[0][g] = w1[((16 + h0) * 4 + g) * 64 + c.lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int sn = 16 + ((q + 1 + h0) & 15);    // (the last prefetch wraps and is unused)
#pragma unroll
        for (int g = 0; g < 4; ++g) fb[(q + 1) & 1][g] = w1[(sn * 4 + g) * 64 + c.lane];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc1[g] = mma_bf16(fa[q], fb[q & 1][g], acc1[g]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * p.dbg_T + p.t0) * 16 : nullptr;
  for (int t = p.t0; t < p.t1; ++t) {
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
    asm volatile("" : "+s"(c.wave), "+s"(c.u0), "+s"(c.rbegin), "+s"(c.nrow), "+s"(c.MT));
    T* h_att_prev = (T*)p.h_att + (size_t)t * NH;
    T* h_att_new = h_att_prev + NH;
    T* h_lang_prev = (T*)p.h_lang + (size_t)t * NH;
    T* h_lang_new = h_lang_prev + NH;
    float* att_h = p.att_h_all + (size_t)t * NH;
    T* ctx = (T*)p.ctx_all + (size_t)t * NH;
    c.dbg = dbg;
    // every exchanged slab is addressed through ONE buffer descriptor (base = lowest of their addresses, from the host) plus
    // a 32-bit byte offset: which slab a k-step comes from is then a scalar select of an offset, not of a pointer
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.xbase, 0, -1, 0x00020000);
    const unsigned o_hlp = (unsigned)((const char*)(h_lang_prev + rb) - (const char*)p.xbase);
    const unsigned o_hap = (unsigned)((const char*)(h_att_prev + rb) - (const char*)p.xbase);
    const unsigned o_han = (unsigned)((const char*)(h_att_new + rb) - (const char*)p.xbase);
    const unsigned o_ctx = (unsigned)((const char*)(ctx + rb) - (const char*)p.xbase);
    if (dbg && c.tid == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
    // ---- att_lstm (:431-434): wave w = row tile w (all 32 k-steps, B fragments from LDS); tile 4: k-steps w, w+4, ...
    // The h_att_prev half of K (k-steps 16..31) has been multiplied already (lstm1_early: at the top of the first step, else
    // inside the previous step's last group barrier); here the h_lang_prev half, which that barrier exchanged.
    if (DEC || t == p.t0) lstm1_early(o_hap);
    {
      // k-step s of [h_lang_prev | h_att_prev]: its slab and its byte offset inside a row
      auto a_soff = [&](int s) { return ((s >> 4) ? o_hap : o_hlp) + (unsigned)((s & 15) * 64); };
      const int s0 = c.rank;                        // (fifth tile: this workgroup walks its k-steps starting at s0)
      float pv5[4], cp5 = 0.f;
      const bool split5 = c.MT > WS_NW;             // an 80-row group: a fifth tile, shared by the waves (then every wave has a tile of its own too)
      if (c.wave < c.MT) {
        const int i = c.wave;
        int ar = 16 * i + c.l15;
        ar = ar < c.nrow ? ar : c.nrow - 1;
        const unsigned aoff = (unsigned)((ar * HH + c.lq * 8) * 2);
        int ar5 = 16 * WS_NW + c.l15;
        ar5 = ar5 < c.nrow ? ar5 : c.nrow - 1;
        const unsigned aoff5 = (unsigned)((ar5 * HH + c.lq * 8) * 2);
        const int h0 = c.rank & 15;                 // de-phased walk over the half, as lstm1_early's
        // the whole half of the tile's activations in flight: the L2 latency is paid once and the MFMAs run as the fragments
        // arrive (loads return in order); then the cell update's own operands
        u32x4 fa[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) fa[q] = bload<true>(rx, aoff, o_hlp + (unsigned)(((q + h0) & 15) * 64));
#if !WS_EARLY_PRE
        float pvn[4][4], cpn[4];
        load_pre(t, i, pvn, cpn);
#endif
        // B fragments of k-step q + 1 are read from LDS while the MFMAs of k-step q run (left alone hipcc emits
        // read - wait - MFMA per fragment: an LDS round trip per MFMA, 7 us for the 128 of a tile)
        u32x4 fb[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) fb[0][g] = w1[(h0 * 4 + g) * 64 + c.lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int sn = (q + 1 + h0) & 15;         // (the last prefetch wraps to k-step h0 and is unused)
#pragma unroll
          for (int g = 0; g < 4; ++g) fb[(q + 1) & 1][g] = w1[(sn * 4 + g) * 64 + c.lane];
#pragma unroll
          for (int g = 0; g < 4; ++g) acc1[g] = mma_bf16(fa[q], fb[q & 1][g], acc1[g]);
          if (q == 7 && (DEC ? split5 : true)) {
            // the registers that have just been consumed take this wave's share of tile 4: k-steps wave, wave + 4, ...
            // (training kernel: requested whether or not the group has a fifth tile -- the row index is clamped --, so that the
            // waits of the remaining k-steps are counted exactly; see the note at mfma_range)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int sr = c.wave + 4 * ((k + s0) & 7);
              fa[k] = bload<true>(rx, aoff5, a_soff(sr));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 (&f5)[16] = fa;
        if (dbg && c.tid == 0) dbg[8] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int g = 0; g < 4; ++g) pvn[r][g] += gfo[r][g];
        ws_cell<SAFE>(c, i, acc1, pvn, cpn, p.c_att + (size_t)(t + 1) * NH, h_att_new, (T*)nullptr,
                      p.gates1 ? (T*)p.gates1 + (size_t)t * N * 4 * HH : nullptr, 0.f, 0u, 0u);
        if (dbg && c.tid == 0) dbg[9] = __builtin_amdgcn_s_memrealtime();
        if (split5) {
          f32x4 acc[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int g = 0; g < 4; ++g) fb[0][g] = w1[((c.wave + 4 * (s0 & 7)) * 4 + g) * 64 + c.lane];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int sn = c.wave + 4 * ((q + 1 + s0) & 7);
#pragma unroll
            for (int g = 0; g < 4; ++g) fb[(q + 1) & 1][g] = w1[(sn * 4 + g) * 64 + c.lane];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mma_bf16(f5[q], fb[q & 1][g], acc[g]);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) scr[(c.wave * 4 + g) * 64 + c.lane] = acc[g];
          load_pre5(t, pv5, cp5);
          __syncthreads();                          // (split5 => MT == 5 => all four waves are here)
          {
            float sg4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v = 0.f;
#pragma unroll
              for (int w = 0; w < WS_NW; ++w) v += ((const float*)(scr + (w * 4 + g) * 64 + c.lane))[c.wave];
              sg4[g] = v;
            }
            const int rr = 16 * WS_NW + 4 * c.lq + c.wave;
            const unsigned u = (unsigned)(c.u0 + c.l15);
            const float gi = uic_sigmoid_t<bf16_t>(sg4[0] + (pv5[0] + gf5[0]));
            const float gf = uic_sigmoid_t<bf16_t>(sg4[1] + (pv5[1] + gf5[1]));
            const float gg = uic_tanh<bf16_t>(sg4[2] + (pv5[2] + gf5[2]));
            const float go = uic_sigmoid_t<bf16_t>(sg4[3] + (pv5[3] + gf5[3]));
            const float cn = gf * cp5 + gi * gg;
            const float h = go * uic_tanh<bf16_t>(cn);
            if (rr < c.nrow) {
              const unsigned nn = (unsigned)((c.rbegin + rr) * HH);
              const unsigned o = nn + u;
              p.c_att[(size_t)(t + 1) * NH + o] = cn;
              st_x<SAFE>(h_att_new + o, h);
              if (p.gates1) {
                T* G = (T*)p.gates1 + (size_t)t * N * 4 * HH + 4u * nn + u;
                __builtin_nontemporal_store((bf16_t)gi, G);
                __builtin_nontemporal_store((bf16_t)gf, G + HH);
                __builtin_nontemporal_store((bf16_t)gg, G + 2 * HH);
                __builtin_nontemporal_store((bf16_t)go, G + 3 * HH);
              }
            }
          }
        }
      }
    }
    if (dbg && c.tid == 0) dbg[1] = __builtin_amdgcn_s_memrealtime();
    if (!group_barrier(c)) return;
    if (dbg && c.tid == 0) dbg[2] = __builtin_amdgcn_s_memrealtime();
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));   // keep this phase's lane-derived values out of the others' live ranges
    // ---- h2att (:543): the waves split K, wave w (and wave 0 for tile 4) reduces its row tile
    {
      const __amdgpu_buffer_rsrc_t r_h2 = rsrc_of(p.h2att_w);
      constexpr int P2D = WS_P2_ALL ? MT_MAX : 2;   // row tiles of activations in flight
      u32x4 wh[4], fa[P2D][4];
      auto tile_of = [&](int i) { const int x = i + tr; return x >= c.MT ? x - c.MT : x; };
      auto load_tile = [&](int buf, int i) {
        int ar = 16 * tile_of(i) + c.l15;
        ar = ar < c.nrow ? ar : c.nrow - 1;
        const unsigned aoff = (unsigned)((ar * HH + c.lq * 8) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[buf][j] = bload<true>(rx, aoff, o_han + (unsigned)((c.wave + 4 * ((j + c.rank) & 3)) * 64));
      };
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wh[j] = bload<false>(r_h2, (unsigned)(((c.u0 + c.l15) * HH + c.lq * 8) * 2), (unsigned)((c.wave + 4 * ((j + c.rank) & 3)) * 64));
      // (requests without run-time conditions -- a tile the group does not have is its last real tile again -- and an early exit
      // instead of skipped blocks: the waits in front of every tile's MFMAs are then counted exactly; see the note at mfma_range)
      auto clamp_tile = [&](int i) { return i < c.MT ? i : c.MT - 1; };
#pragma unroll
      for (int i = 0; i < P2D - 1; ++i) load_tile(i, clamp_tile(i));
#pragma unroll
      for (int i = 0; i < MT_MAX; ++i) {
        if (i >= c.MT) break;
        {
          if (i + P2D - 1 < MT_MAX) load_tile((i + P2D - 1) % P2D, DEC ? (i + P2D - 1 < c.MT ? i + P2D - 1 : i) : clamp_tile(i + P2D - 1));
          __builtin_amdgcn_sched_barrier(0);
          f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j) a = mma_bf16(fa[i % P2D][j], wh[j], a);
          scr[(c.wave * MT_MAX + tile_of(i)) * 64 + c.lane] = a;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      const int a = c.u0 + c.l15;
      const float bias = p.h2att_b ? p.h2att_b[a] : 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = c.wave + WS_NW * k;
        if (i < c.MT && (k == 0 || c.wave == 0)) {
          f32x4 sum = scr[(0 * MT_MAX + i) * 64 + c.lane];
#pragma unroll
          for (int w = 1; w < WS_NW; ++w) sum += scr[(w * MT_MAX + i) * 64 + c.lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * i + 4 * c.lq + r;
            if (rr < c.nrow) st_x<SAFE>(att_h + (unsigned)((c.rbegin + rr) * HH + a), sum[r] + bias);
          }
        }
      }
    }
    if (dbg && c.tid == 0) dbg[3] = __builtin_amdgcn_s_memrealtime();
    // ---- attention (:544-556)
    if constexpr (!DEC) {
      group_arrive(c);
      asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
      uint4 apre[AW_CR];
      attn_wave_preload(c, p, apre);                  // (not exchanged data: its latency passes while the workgroup waits for the others)
      if (!group_wait(c, (int*)c.smem)) return;
      if (dbg && c.tid == 0) dbg[4] = __builtin_amdgcn_s_memrealtime();
      attn_phase_wave<SAFE>(c, p, att_h, p.alpha_all + (size_t)t * N * p.R, ctx, apre);
    } else {                                          // (the decode kernel has no registers to spare)
      if (!group_barrier(c)) return;
      if (dbg && c.tid == 0) dbg[4] = __builtin_amdgcn_s_memrealtime();
      asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));   // keep this phase's lane-derived values out of the others' live ranges
      attn_phase<T, SAFE, WS_NW, WS_ATT_SLOTS>(c, p, att_h, p.alpha_all + (size_t)t * N * p.R, ctx);
    }
    if (dbg && c.tid == 0) dbg[5] = __builtin_amdgcn_s_memrealtime();
    group_arrive(c);                                // (the wait is inside the block below, behind the requests that need no exchange)
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));   // keep this phase's lane-derived values out of the others' live ranges
    // ---- lang_lstm (:438-441) + output dropout (:443): K split over the waves (stationary fragments in registers), one row
    // tile per pass; pass i writes its partial tiles to LDS half i & 1, ONE barrier, then the tile's owner sums and finishes
    // it while the other waves already multiply tile i + 1 (whose partials go to the other half)
    {
      auto tile_of = [&](int i) { const int x = i + tr; return x >= c.MT ? x - c.MT : x; };
      // The cell update of a tile is shared by the four waves: wave w takes accumulator component w, i.e. row 4 lq + w of
      // every 4-row group, so a pass costs a quarter of the transcendental work on its critical path.
      float cl[MT_MAX];                            // c_lang of (tile i, row 4 lq + wave, unit u0 + l15)
      {
        const float* c_prev = p.c_lang + (size_t)t * NH;
        const unsigned u = (unsigned)(c.u0 + c.l15);
#pragma unroll
        for (int i = 0; i < MT_MAX; ++i) {
          const int rr = 16 * i + 4 * c.lq + c.wave;
          cl[i] = c_prev[(unsigned)((c.rbegin + (rr < c.nrow ? rr : c.nrow - 1)) * HH) + u];
        }
      }
      // One register set of 12 activation fragments: fragment j of the NEXT row tile is requested right after the MFMAs that
      // consumed fragment j of this one, so a whole pass (MFMAs, barrier, the owner's cell update) covers its latency.
      u32x4 fa[12];
      auto aoff_of = [&](int i) {
        int ar = 16 * i + c.l15;
        ar = ar < c.nrow ? ar : c.nrow - 1;
        return (unsigned)((ar * HH + c.lq * 8) * 2);
      };
      auto load_frag = [&](int j, unsigned aoff) {
        const int m = ((j >> 2) == 2 ? 0 : (j >> 2) + 1) * 4 + ((j + j0) & 3);     // (as the stationary fragments' slots)
        // (two independent selects: a three-way select chain becomes a table in private memory, and with it everything
        // this lambda captures)
        const unsigned so = o_ctx + (m >= 4 ? o_han - o_ctx : 0u) + (m >= 8 ? o_hlp - o_han : 0u);
        fa[j] = bload<true>(rx, aoff, so + (unsigned)(((m & 3) * 4 + c.wave) * 64));
      };
      {
        // the first row tile's h_att / h_lang_prev fragments (exchanged one and three barriers ago) are requested before this
        // workgroup looks for the others' arrival at the barrier behind the attention; att_res's four behind it
        const unsigned a0 = aoff_of(tile_of(0));
#pragma unroll
        for (int j = 0; j < 8; ++j) load_frag(j, a0);
      }
      // ... and multiplied too (the first tile's eight k-fragments that need no exchange): in the time between this workgroup's
      // arrival and its first look at the barrier.  mfma_range: slots [jlo, jhi) of a pass (see the note on the inline asm below).
      f32x4 acc0[4];
      // `more` is a COMPILE-TIME fact (is there a later pass in the unrolled loop at all); whether the group really has that tile
      // only picks the address (the last real tile's again).  A load under a run-time branch here makes hipcc's s_waitcnt
      // insertion assume it may NOT have been issued, and every later slot of the pass then waits with vmcnt(4) .. vmcnt(0) --
      // i.e. for the loads this very pass has just issued, an L2 round trip per pass (round 6: 1.46 us of a pass's 2.26 were this).
      auto mfma_range = [&](f32x4 (&acc)[4], int jlo, int jhi, bool more, unsigned an) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          if (j < jlo || j >= jhi) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (j == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[g]) : "v"(fa[0]), "a"(w2[0][g]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(fa[j]), "a"(w2[j][g]));
          }
          if (more) load_frag(j, an);
        }
      };
      {
        constexpr bool more0 = 1 < MT_MAX;
        if constexpr (!DEC) mfma_range(acc0, 0, 8, more0, aoff_of(tile_of(1 < c.MT ? 1 : 0)));   // (not the decode kernel: it spills as it is)
        if (!group_wait(c, (int*)c.smem)) return;
        if (dbg && c.tid == 0) dbg[6] = __builtin_amdgcn_s_memrealtime();
        const unsigned a0 = aoff_of(tile_of(0));
#pragma unroll
        for (int j = 8; j < 12; ++j) load_frag(j, a0);
      }
#pragma unroll
      for (int i = 0; i < MT_MAX; ++i) {
        if (i >= c.MT) break;                       // (an early exit, not a skipped block: pass i + 1 is reachable through pass i only)
        {
          const bool more = DEC ? (i + 1 < MT_MAX && i + 1 < c.MT) : i + 1 < MT_MAX;      // training kernel: compile-time (see mfma_range)
          const int tile = tile_of(i);
          const unsigned an = aoff_of(tile_of(i + 1 < MT_MAX && i + 1 < c.MT ? i + 1 : i));
          // (live-position logit layer: where this lane's row of the tile goes in the compact operand -- requested now, used in the
          // epilogue behind the MFMAs; a valid address whatever the row)
          int live_ci = -1;
          if constexpr (!DEC) {
            if (p.live_inv) {
              const int rq = 16 * tile + 4 * c.lq + c.wave;
              live_ci = p.live_inv[t * N + c.rbegin + (rq < c.nrow ? rq : 0)];
            }
          }
          // The stationary B fragments are named as ACCUMULATOR-file operands ("a"), which is what keeps them there for the
          // whole launch: left to itself hipcc parks them in AGPRs but copies each one back (4 x v_accvgpr_read) in front of
          // every MFMA.  Inline-asm MFMAs get no hazard padding from the compiler: gate g's chain is re-entered only after
          // the three other gates' MFMAs (the matrix pipe is in order), and the nops below cover MFMA result -> LDS store.
#if WS_EARLY_PRE
          if (!more && t + 1 < p.t1) load_pre(t + 1, c.wave, pvn, cpn);   // last pass: no later request of this phase waits behind these
#endif
          f32x4 acc[4];
          if constexpr (!DEC) {
            if (i == 0) {
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[g] = acc0[g];
              mfma_range(acc, 8, 12, more, an);
            } else {
              mfma_range(acc, 0, 12, more, an);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                if (j == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[g]) : "v"(fa[0]), "a"(w2[0][g]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(fa[j]), "a"(w2[j][g]));
              }
              if (more) load_frag(j, an);
            }
          }
          asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
          f32x4* half = scr + (i & 1) * (WS_NW * 4 * 64);
#pragma unroll
          for (int g = 0; g < 4; ++g) half[(c.wave * 4 + g) * 64 + c.lane] = acc[g];
          __syncthreads();
          if (dbg && c.tid == 0) dbg[11 + i] = __builtin_amdgcn_s_memrealtime();
          {
            // every wave sums component `wave` of the four partial tiles (fixed order: deterministic) and finishes that row
            float sg4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v = 0.f;
#pragma unroll
              for (int w = 0; w < WS_NW; ++w) v += ((const float*)(half + (w * 4 + g) * 64 + c.lane))[c.wave];
              sg4[g] = v;
            }
            float cst = cl[0];
#pragma unroll
            for (int k = 1; k < MT_MAX; ++k) cst = tile == k ? cl[k] : cst;
            const int rr = 16 * tile + 4 * c.lq + c.wave;
            const unsigned u = (unsigned)(c.u0 + c.l15);
            const float gi = uic_sigmoid_t<bf16_t>(sg4[0] + pb[0][0]);
            const float gf = uic_sigmoid_t<bf16_t>(sg4[1] + pb[0][1]);
            const float gg = uic_tanh<bf16_t>(sg4[2] + pb[0][2]);
            const float go = uic_sigmoid_t<bf16_t>(sg4[3] + pb[0][3]);
            const float cn = gf * cst + gi * gg;
            const float h = go * uic_tanh<bf16_t>(cn);
            if (rr < c.nrow) {
              const unsigned nn = (unsigned)((c.rbegin + rr) * HH);
              const unsigned o = nn + u;
              p.c_lang[(size_t)(t + 1) * NH + o] = cn;
              st_x<SAFE>(h_lang_new + o, h);
              if (p.hdrop_all) {
                float hd = h;
                if (p.drop_p > 0.f) hd *= uic_drop_scale(p.seed, UIC_SITE_OUT0 + (unsigned)t, o, p.drop_p, 1.f / (1.f - p.drop_p));
                // (decode mode: the logit phase of every workgroup of the group reads these rows -- exchanged data)
                if (DEC) st_x<SAFE>((T*)p.hdrop_all + (size_t)t * NH + o, hd);
                else {
                  ((T*)p.hdrop_all)[(size_t)t * NH + o] = (bf16_t)hd;
                  // (live-position logit layer: the row also goes to its place in the compact operand -- no gather launches)
                  if (live_ci >= 0) ((T*)p.hdrop_live)[(size_t)live_ci * HH + u] = (bf16_t)hd;
                }
              }
              if (p.gates2) {
                T* G = (T*)p.gates2 + (size_t)t * N * 4 * HH + 4u * nn + u;
                __builtin_nontemporal_store((bf16_t)gi, G);
                __builtin_nontemporal_store((bf16_t)gf, G + HH);
                __builtin_nontemporal_store((bf16_t)gg, G + 2 * HH);
                __builtin_nontemporal_store((bf16_t)go, G + 3 * HH);
              }
            }
          }
        }
      }
      __syncthreads();                              // the scratch halves are free again (barrier flag, next step)
    }
    if (dbg && c.tid == 0) dbg[7] = __builtin_amdgcn_s_memrealtime();
    if (!DEC) {
      group_arrive(c);
      if (t + 1 < p.t1) lstm1_early(o_han);         // (step t + 1's h_att_prev = this step's h_att_new)
      if (!group_wait(c, (int*)c.smem)) return;
    } else {
      group_arrive(c);
      asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
      if (!dec_logits<SAFE>(p, c, t, lds)) return;
      if (dbg && c.tid == 0) dbg[8] = __builtin_amdgcn_s_memrealtime();
      if (!group_barrier(c)) return;
      asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
      dec_sample<SAFE>(p, c, t, lds);
      if (dbg && c.tid == 0) dbg[9] = __builtin_amdgcn_s_memrealtime();
      const bool tok_in_lds = p.dec_sample_max || p.dec_forced;      // greedy / forced: dec_sample left the group's tokens in LDS
      if (tok_in_lds) __syncthreads();
      else if (!group_barrier(c)) return;
      asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
      if (t + 1 < p.t1) dec_xt_gemm(p, c, t + 1, xg, xg5, tok_in_lds ? (const int*)(c.smem + DEC_OFF_LIST) : nullptr);
      if (dbg && c.tid == 0) dbg[10] = __builtin_amdgcn_s_memrealtime();
    }
    if (dbg) dbg += 16;
  }
}

__global__ __launch_bounds__(WS_NTH) void rnn_fwd_persist_ws_kernel(const UicRnnFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  const int mode = setup_ctx(p, smem + WS_W1_BYTES, c);
  if (mode == 0) return;
  if (mode == 2) ws_run<true, false>(p, c, smem);
  else ws_run<false, false>(p, c, smem);
}
__global__ __launch_bounds__(WS_NTH) void rnn_dec_persist_ws_kernel(const UicRnnFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  const int mode = setup_ctx(p, smem + WS_W1_BYTES, c);
  if (mode == 0) return;
  if (mode == 2) ws_run<true, true>(p, c, smem);
  else ws_run<false, true>(p, c, smem);
}

// relu(embed) in bf16: what every decode step gathers its input rows from (dropout, where the pass has it, is applied to the
// gathered values; with the reference's p = 0.5 that is exactly dropout(relu(embed)) rounded once, otherwise within one bf16 ulp)
__global__ void dec_embed_relu_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = *(const float4*)(w + 4 * i);
    *(uint2*)(out + 4 * i) = make_uint2(uic_pack_bf16x2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)), uic_pack_bf16x2(fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)));
  }
}
// the same from the bf16 copy of the table (relu of a bf16 value: clear the halves whose sign bit is set)
__global__ void dec_embed_relu16_kernel(const uint2* __restrict__ w, uint2* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    uint2 v = w[i];
    if (v.x & 0x8000u) v.x &= 0xFFFF0000u;
    if (v.x & 0x80000000u) v.x &= 0x0000FFFFu;
    if (v.y & 0x8000u) v.y &= 0xFFFF0000u;
    if (v.y & 0x80000000u) v.y &= 0x0000FFFFu;
    out[i] = v;
  }
}

// the reference stops decoding once every row has finished (AttModel.py:236-238): entries after that step stay zero.
// status[0] != 0: a persistent launch gave up waiting for its workgroups (bounded spin) and left its outputs partly unwritten --
// the captions are then POISONED (token -1, log-prob NaN) so that no caller can score or print them as if they were results.
__global__ void dec_finish_kernel(int64_t* __restrict__ seq, float* __restrict__ seq_logp, int N, int L, int ld, const int* __restrict__ status) {
  __shared__ int s_max;
  if (status && status[0] != 0) {
    for (int i = threadIdx.x; i < N * L; i += blockDim.x) {
      const int n = i / L, t = i - n * L;
      seq[(size_t)n * ld + t] = -1;
      seq_logp[(size_t)n * ld + t] = __int_as_float(0x7fc00000);
    }
    return;
  }
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  int mx = 0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    int f = L;                                   // first step at which the row emitted token 0
    for (int t = 0; t < L; ++t)
      if (seq[(size_t)n * ld + t] == 0) { f = t; break; }
    mx = max(mx, f);
  }
  atomicMax(&s_max, mx);
  __syncthreads();
  const int t_dead = s_max + 1;                  // every row had finished before this step
  for (int i = threadIdx.x; i < N * L; i += blockDim.x) {
    const int n = i / L, t = i - n * L;
    if (t >= t_dead) seq_logp[(size_t)n * ld + t] = 0.f;
  }
}

}  // namespace

// Two persistent launches must not be on the chip at once (each holds every CU while it waits, bounded, for its own
// workgroups to become resident): inside one process every persistent launch therefore waits for the previous one on its
// device, whatever stream that ran on.  (Across processes the caller has to choose: UIC_REC_FWD_CHAIN.)
// enter .. leave is ONE critical section per device (host threads call in through ctypes without the GIL): the lock taken by
// enter is held until leave has recorded the launch's event -- or until abandon, on an error path (UicPersistGateScope).
namespace {
struct PersistGate { hipEvent_t ev = nullptr; bool recorded = false; std::mutex mu; };
PersistGate g_gate[16];
int gate_device(hipStream_t s, int* dev) {
  // the STREAM's device, not the calling thread's current one (the default stream belongs to the current device)
  if (s) return uic_check_hip(hipStreamGetDevice(s, dev), "hipStreamGetDevice");
  return uic_check_hip(hipGetDevice(dev), "hipGetDevice");
}
}
int uic_persist_gate_enter(hipStream_t s, int* dev_out) {
  int dev = 0;
  UIC_TRY(gate_device(s, &dev));
  UIC_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
  PersistGate& g = g_gate[dev];
  g.mu.lock();
  int rc = UIC_OK;
  if (!g.ev) rc = uic_check_hip(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming), "hipEventCreate");
  if (rc == UIC_OK && g.recorded) rc = uic_check_hip(hipStreamWaitEvent(s, g.ev, 0), "hipStreamWaitEvent(persistent gate)");
  if (rc != UIC_OK) { g.mu.unlock(); return rc; }
  *dev_out = dev;
  return UIC_OK;
}
int uic_persist_gate_leave(hipStream_t s, int dev) {
  PersistGate& g = g_gate[dev];
  const int rc = uic_check_hip(hipEventRecord(g.ev, s), "hipEventRecord(persistent gate)");
  if (rc == UIC_OK) g.recorded = true;
  g.mu.unlock();
  return rc;
}
void uic_persist_gate_abandon(int dev) { g_gate[dev].mu.unlock(); }

constexpr int MAX_SLABS = 8;       // launches of <= 640 caption rows each
size_t uic_rnn_persist_sync_bytes() { return (size_t)MAX_SLABS * SY_WORDS * 4; }

bool uic_rnn_persist_eligible(int dtype, int N, int H, int A, int R) {
  if (dtype != UIC_BF16 && dtype != UIC_F32) return false;
  if (H != HH || A != HH || R < 1 || R > ATT_R || N < 1 || N > MAX_SLABS * 8 * 16 * MT_MAX) return false;
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    cus = prop.multiProcessorCount;
  }
  return cus == 8 * PW;        // one workgroup per CU, 32 per XCD
}

int uic_rnn_fwd_persist_launch(const UicRnnFwdParams& p0, hipStream_t s) {
  UIC_REQUIRE(p0.sync && p0.t1 > p0.t0 && p0.N > 0, "rnn_fwd_persist: bad arguments");
  UIC_REQUIRE(!p0.dec || (p0.dtype == UIC_BF16 && p0.gfc && p0.dec_embed_relu && p0.dec_xw && p0.dec_logit_w && p0.dec_logits && p0.dec_part &&
                          p0.dec_tok && p0.dec_unf && p0.dec_seq && p0.dec_seq_logp && p0.hdrop_all && p0.dec_V1 >= 2 &&
                          p0.dec_V1 <= PW * WS_NW * DEC_NJ * 16),
              "rnn_fwd_persist: decode mode needs bf16, gfc, the embedding / logit operands and its exchange buffers (V1 <= %d)", PW * WS_NW * DEC_NJ * 16);
  static bool configured = false;
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)rnn_fwd_persist_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), "hipFuncSetAttribute(rnn persist)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)rnn_fwd_persist_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES), "hipFuncSetAttribute(rnn persist ws)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)rnn_dec_persist_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES), "hipFuncSetAttribute(rnn decode persist)"));
    configured = true;
  }
  const int G = 8, cap = G * 16 * MT_MAX;     // caption rows one launch covers
  UicPersistGateScope gate;
  UIC_TRY(gate.enter(s));
  for (int r0 = 0; r0 < p0.N; r0 += cap) {
    UicRnnFwdParams p = p0;
    p.row0 = r0;
    p.Nrows = p0.N - r0 < cap ? p0.N - r0 : cap;
    {
      const char* lo = (const char*)p.h_att;
      if ((const char*)p.h_lang < lo) lo = (const char*)p.h_lang;
      if ((const char*)p.ctx_all < lo) lo = (const char*)p.ctx_all;
      const char* ends[3] = {(const char*)p.h_att, (const char*)p.h_lang, (const char*)p.ctx_all};
      for (int k = 0; k < 3; ++k)
        UIC_REQUIRE((size_t)(ends[k] - lo) + (size_t)(p.t1 + 1) * p.N * HH * 4 < ((size_t)1 << 32), "rnn_fwd_persist: the state slabs must lie within 4 GB of each other");
      p.xbase = lo;
    }
    p.sync = p0.sync + (size_t)(r0 / cap) * SY_WORDS;
    if (!(p0.sync_zeroed && r0 == 0)) UIC_TRY(uic_check_hip(hipMemsetAsync(p.sync, 0, (size_t)SY_WORDS * 4, s), "hipMemsetAsync(rnn sync)"));
    // bf16: the weight-stationary kernel; f32 (the parity path): the generic one, weights re-read from L2 every step
    if (p.dec) hipLaunchKernelGGL(rnn_dec_persist_ws_kernel, dim3(G * PW), dim3(WS_NTH), WS_LDS_BYTES, s, p);
    else if (p.dtype == UIC_BF16) {
      UIC_REQUIRE(p.e_att, "rnn_fwd_persist: the weight-stationary training kernel reads e_att = e^{2 p_att} (uic_exp2x2_launch)");
      hipLaunchKernelGGL(rnn_fwd_persist_ws_kernel, dim3(G * PW), dim3(WS_NTH), WS_LDS_BYTES, s, p);
    }
    else hipLaunchKernelGGL(rnn_fwd_persist_kernel<float>, dim3(G * PW), dim3(NTH), LDS_BYTES, s, p);
    UIC_LAUNCH_CHECK("rnn_fwd_persist_kernel");
  }
  return gate.leave();
}

bool uic_rnn_decode_persist_eligible(int dtype, int N, int H, int A, int R, int E, int V1) {
  return dtype == UIC_BF16 && E == HH && V1 >= 2 && V1 <= PW * WS_NW * DEC_NJ * 16 && uic_rnn_persist_eligible(dtype, N, H, A, R);
}
size_t uic_rnn_decode_part_floats(int N) { return (size_t)N * PW * 4; }
int uic_rnn_decode_finish_launch(int64_t* seq, float* seq_logp, int N, int L, int ld, const int* status, hipStream_t s) {
  UIC_REQUIRE(seq && seq_logp && N > 0 && L > 0 && ld >= L, "rnn_decode_finish: bad arguments");
  hipLaunchKernelGGL(dec_finish_kernel, dim3(1), dim3(1024), 0, s, seq, seq_logp, N, L, ld, status);
  UIC_LAUNCH_CHECK("dec_finish_kernel");
  return UIC_OK;
}
int uic_rnn_decode_embed_relu_launch(const void* embed_w, int table_dtype, void* out_bf16, int V1, int E, hipStream_t s) {
  UIC_REQUIRE(embed_w && out_bf16 && V1 > 0 && E % 4 == 0, "rnn_decode_embed_relu: bad arguments");
  const size_t n4 = (size_t)V1 * E / 4;
  const dim3 grid((unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256));
  if (table_dtype == UIC_BF16) hipLaunchKernelGGL(dec_embed_relu16_kernel, grid, dim3(256), 0, s, (const uint2*)embed_w, (uint2*)out_bf16, n4);
  else hipLaunchKernelGGL(dec_embed_relu_kernel, grid, dim3(256), 0, s, (const float*)embed_w, (bf16_t*)out_bf16, n4);
  UIC_LAUNCH_CHECK("dec_embed_relu_kernel");
  return UIC_OK;
}
