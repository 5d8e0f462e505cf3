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
#include "uic_common.h"
#include "../../include/uic_hip.h"
#include <stdlib.h>

namespace {

constexpr int PW = 32;              // workgroups per row group (= CUs per XCD); each owns HH / PW = 16 hidden units
constexpr int NWAVE = 8;
constexpr int NTH = NWAVE * 64;
constexpr int MT_MAX = 5;           // 16-row tiles per group
constexpr int HH = 16 * PW;         // rnn_size == att_hid_size == 512 (P/opts.py:45-46 defaults)
constexpr int HALF_T = 3;           // row tiles reduced per LDS pass (8 waves x 3 tiles x 4 gates x 1 KB = 96 KB)
constexpr unsigned SPIN_MAX = 1u << 17;
constexpr int ATT_UB = 5;           // regions per wave in the attention phase: R <= 8 * 5
constexpr int LDS_BYTES = NWAVE * HALF_T * 4 * 1024;

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// words of the sync block (each counter on a 128-byte line of its own)
enum { SY_TOTAL = 0, SY_ERR = 32, SY_XCC = 64, SY_BAR = 64 + 32 * 8, SY_WORDS = 64 + 32 * 8 + 32 * 8 };

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // lane l holds A[row l&15][k0 + 4(l>>4) + j] in component j: four 16x16x4 products, K permuted identically for A and B
  static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}
template <bool SC1>
__device__ __forceinline__ u32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, SC1 ? 16 : 0));
}

// a value another workgroup of the group will read in this launch
template <bool SAFE> __device__ __forceinline__ void st_x(bf16_t* p, float v) {
  const bf16_t b = (bf16_t)v;
  if (SAFE) __hip_atomic_store((unsigned short*)p, __builtin_bit_cast(unsigned short, b), RLX_AGENT);
  else *p = b;
}
template <bool SAFE> __device__ __forceinline__ void st_x(float* p, float v) {
  if (SAFE) __hip_atomic_store(p, v, RLX_AGENT);
  else *p = v;
}

struct Ctx {
  int tid, lane, wave, l15, lq;
  int group, rank, u0;
  int rbegin, nrow, MT;
  unsigned* bar; unsigned* err; unsigned* status; unsigned bar_target;
  char* smem;
  unsigned long long* dbg; int exp;
};

// Bounded group barrier.  Every wave first drains its own stores (the payload must be in L2 / memory before the
// arrival is visible), then one lane arrives and polls.  Returns false after a timeout (uniform over the workgroup).
__device__ __forceinline__ bool group_barrier(Ctx& c) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  c.bar_target += PW;
  int* flag = (int*)c.smem;
  if (c.tid == 0) {
    __hip_atomic_fetch_add(c.bar, 1u, RLX_AGENT);
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(c.bar, RLX_AGENT) < c.bar_target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_MAX || ((spins & 255u) == 0 && __hip_atomic_load(c.err, RLX_AGENT) != 0)) {
        __hip_atomic_store(c.err, 0x100u + (unsigned)c.group, RLX_AGENT);
        if (c.status) __hip_atomic_store(c.status, 0x100u + (unsigned)c.group, RLX_AGENT);
        ok = 0;
        break;
      }
    }
    *flag = ok;
  }
  __syncthreads();
  const int ok = *flag;
  __syncthreads();      // the flag word lives in the reduction buffer
  return ok != 0;
}

// acc[i][g] += A_seg[rows of tile i, k-steps of this wave] * B_seg[rows brow(g), same k-steps]^T, summed over the segments.
// A_seg: [nrow, HH] slab of the group's rows (row stride HH), exchanged data (sc1 loads).  B_seg: weight block, row
// stride ldb elements, K contiguous.  GATES: column tile g = gate g of the workgroup's 16 units (weight row g*HH + u0 + c).
template <typename T, int NSEG, int NCT, bool GATES>
__device__ __forceinline__ void gemm_ksplit(const Ctx& c, f32x4 (&acc)[MT_MAX][NCT], const void* const (&Aseg)[NSEG],
                                            const void* const (&Bseg)[NSEG], const int (&ldb)[NSEG]) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KE = 4 * VEC;                    // K elements per k-step
  constexpr int SPS = HH / KE / NWAVE;           // k-steps of one wave per segment
  constexpr int KPW = NSEG * SPS;
  unsigned aoff[MT_MAX];
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i) {
    int r = 16 * i + c.l15;
    r = r < c.nrow ? r : c.nrow - 1;             // rows past the group's share re-read its last row (results unused)
    aoff[i] = (unsigned)((r * HH + c.lq * VEC) * (int)sizeof(T));
  }
  unsigned brow[NCT];
#pragma unroll
  for (int g = 0; g < NCT; ++g) brow[g] = (unsigned)((GATES ? g * HH : 0) + c.u0 + c.l15);

  u32x4 fa[2][MT_MAX], fb[2][NCT];
  auto load = [&](int buf, int s) {
    const int sg = s / SPS;
    const unsigned kk = (unsigned)((((s % SPS) * NWAVE + c.wave) * KE) * (int)sizeof(T));
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(Aseg[sg]);
    const __amdgpu_buffer_rsrc_t rb = rsrc_of(Bseg[sg]);
#pragma unroll
    for (int g = 0; g < NCT; ++g)
      fb[buf][g] = (c.exp & 1) ? u32x4{0, 0, 0, 0} : bload<false>(rb, (brow[g] * (unsigned)ldb[sg] + (unsigned)(c.lq * VEC)) * (unsigned)sizeof(T), kk);
#pragma unroll
    for (int i = 0; i < MT_MAX; ++i)
      if (i < c.MT) fa[buf][i] = (c.exp & 2) ? u32x4{0, 0, 0, 0} : bload<true>(ra, aoff[i], kk);
  };
  // two k-steps of operands in flight per wave (8 waves x 2 x 9 KB per CU: enough to cover the L2 latency at the
  // ~70 GB/s a CU takes in); the scheduling barriers keep hipcc from hoisting every k-step's loads to the top
  load(0, 0);
#pragma unroll
  for (int s = 0; s < KPW; ++s) {
    if (s + 1 < KPW) load((s + 1) & 1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT_MAX; ++i)
      if (i < c.MT) {
#pragma unroll
        for (int g = 0; g < NCT; ++g) acc[i][g] = Mma<T>::run(fa[s & 1][i], fb[s & 1][g], acc[i][g]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NCT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT_MAX][NCT]) {
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i)
#pragma unroll
    for (int g = 0; g < NCT; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// One nn.LSTMCell (P/models/AttModel.py:434 / :441) for the group's rows and this workgroup's 16 units.
// pre: gate pre-activations that do not depend on the recurrence, added per (row, gate, unit).
template <typename T, bool SAFE, int NSEG, typename PreFn>
__device__ __forceinline__ void lstm_phase(Ctx& c, const void* const (&Aseg)[NSEG], const void* const (&Bseg)[NSEG],
                                           const int (&ldb)[NSEG], PreFn pre, const float* c_prev, float* c_out, T* h_out,
                                           T* h_drop, T* gates_out, int N, float drop_p, unsigned seed, unsigned site) {
  // the tile-owner threads fetch what the cell update needs before the GEMM: its latency hides behind the operand stream.
  // (every address below is a uniform base + a 32-bit lane offset: no 64-bit per-lane pointers to keep alive)
  const bool owner = c.wave < c.MT;
  const unsigned u = (unsigned)(c.u0 + c.l15);
  unsigned nn[4];
  float pv[4][4], cp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = 16 * c.wave + 4 * c.lq + r;
    nn[r] = (unsigned)((c.rbegin + (rr < c.nrow ? rr : c.nrow - 1)) * HH);
  }
  if (owner) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cp[r] = c_prev[nn[r] + u];
#pragma unroll
      for (int g = 0; g < 4; ++g) pv[r][g] = pre(4u * nn[r] + (unsigned)(g * HH) + u, (unsigned)(g * HH) + u);
    }
  }
  f32x4 acc[MT_MAX][4];
  zero_acc<4>(acc);
  gemm_ksplit<T, NSEG, 4, true>(c, acc, Aseg, Bseg, ldb);
  if (c.dbg && c.tid == 0) c.dbg[8 + (NSEG == 2 ? 0 : 4)] = __builtin_amdgcn_s_memrealtime();
  f32x4* red = (f32x4*)c.smem;      // [wave][tile in pass][gate][lane]
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int tb = pass * HALF_T;
    if (tb >= c.MT) break;
#pragma unroll
    for (int i = 0; i < HALF_T; ++i) {
      if (tb + i < MT_MAX && tb + i < c.MT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) red[((c.wave * HALF_T + i) * 4 + g) * 64 + c.lane] = acc[tb + i][g];
      }
    }
    __syncthreads();
    if (c.dbg && c.tid == 0) c.dbg[9 + pass + (NSEG == 2 ? 0 : 4)] = __builtin_amdgcn_s_memrealtime();
    if (owner && c.wave >= tb && c.wave < tb + HALF_T) {
      f32x4 s[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s[g] = red[((0 * HALF_T + (c.wave - tb)) * 4 + g) * 64 + c.lane];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) s[g] += red[((w * HALF_T + (c.wave - tb)) * 4 + g) * 64 + c.lane];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * c.wave + 4 * c.lq + r;
        if (rr < c.nrow) {
          const unsigned o = nn[r] + u;                      // element (row n, unit u) of an [N, HH] slab
          const float gi = uic_sigmoid_t<T>(s[0][r] + pv[r][0]);
          const float gf = uic_sigmoid_t<T>(s[1][r] + pv[r][1]);
          const float gg = uic_tanh<T>(s[2][r] + pv[r][2]);
          const float go = uic_sigmoid_t<T>(s[3][r] + pv[r][3]);
          const float cn = gf * cp[r] + gi * gg;
          const float h = go * uic_tanh<T>(cn);
          c_out[o] = cn;
          st_x<SAFE>(h_out + o, h);
          if (h_drop) {
            float hd = h;
            if (drop_p > 0.f) hd *= uic_drop_scale(seed, site, o, drop_p, inv_keep);
            h_drop[o] = uic_from_f<T>(hd);
          }
          if (gates_out) {
            const unsigned og = 4u * nn[r] + u;              // read again only in the backward pass
            __builtin_nontemporal_store(uic_from_f<T>(gi), gates_out + og);
            __builtin_nontemporal_store(uic_from_f<T>(gf), gates_out + og + HH);
            __builtin_nontemporal_store(uic_from_f<T>(gg), gates_out + og + 2 * HH);
            __builtin_nontemporal_store(uic_from_f<T>(go), gates_out + og + 3 * HH);
          }
        }
      }
    }
    __syncthreads();
  }
  (void)N;
}

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

// Attention.forward after h2att (P/models/AttModel.py:544-556) for caption row n, all 8 waves (attention.hip's
// attn_fwd_fast_kernel with the row's att_h read past the L1)
template <typename T, bool SAFE>
__device__ __forceinline__ void attn_row(const Ctx& c, const UicRnnFwdParams& p, int n, const float* att_h, float* alpha, T* ctx) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int CH = HH / VEC / 64;               // 16-byte chunks of a 512-wide row per lane (bf16: 1, f32: 2)
  const int R = p.R;
  float* s_e = (float*)c.smem + 64;               // [R][4] row-of-16 partial scores (word 0 of smem is the barrier flag)
  float* s_red = s_e + 4 * ((R + 3) & ~3);        // [NWAVE][HH]
  const T* pa = (const T*)p.p_att + (size_t)n * R * HH;
  const T* pt = (const T*)p.att + (size_t)n * R * HH;
  uint4 vp[ATT_UB][CH], va[ATT_UB][CH];
#pragma unroll
  for (int u = 0; u < ATT_UB; ++u) {
    const int r = min(c.wave + u * NWAVE, R - 1);
#pragma unroll
    for (int k = 0; k < CH; ++k) vp[u][k] = *(const uint4*)(pa + (unsigned)(r * HH + (c.lane + 64 * k) * VEC));
  }
#pragma unroll
  for (int u = 0; u < ATT_UB; ++u) {
    const int r = min(c.wave + u * NWAVE, R - 1);
#pragma unroll
    for (int k = 0; k < CH; ++k) va[u][k] = *(const uint4*)(pt + (unsigned)(r * HH + (c.lane + 64 * k) * VEC));
  }
  float ah[CH][VEC], w[CH][VEC];
  {
    const __amdgpu_buffer_rsrc_t rh = rsrc_of(att_h + (size_t)n * HH);
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        const u32x4 v = bload<true>(rh, (unsigned)(((c.lane + 64 * k) * VEC + q * 4) * 4), 0);
        ah[k][q * 4 + 0] = __uint_as_float(v.x); ah[k][q * 4 + 1] = __uint_as_float(v.y);
        ah[k][q * 4 + 2] = __uint_as_float(v.z); ah[k][q * 4 + 3] = __uint_as_float(v.w);
        const float4 ww = *(const float4*)(p.w_alpha + (c.lane + 64 * k) * VEC + q * 4);
        w[k][q * 4 + 0] = ww.x; w[k][q * 4 + 1] = ww.y; w[k][q * 4 + 2] = ww.z; w[k][q * 4 + 3] = ww.w;
      }
  }
  const float b_alpha = p.b_alpha ? p.b_alpha[0] : 0.f;
#pragma unroll
  for (int u = 0; u < ATT_UB; ++u) {
    const int r = c.wave + u * NWAVE;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      float f[VEC];
      uic_unpack<T>(vp[u][k], f);
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
    const float4 q = *(const float4*)(s_e + c.lane * 4);
    e = (q.x + q.y) + (q.z + q.w) + b_alpha;
  }
  const float mx = uic_wave_max(e);
  const float ex = c.lane < R ? expf(e - mx) : 0.f;
  float wgt = ex * (1.f / uic_wave_sum(ex));
  if (mk) {
    wgt *= c.lane < R ? mk[c.lane] : 0.f;
    wgt = wgt / uic_wave_sum(wgt);
  }
  if (c.wave == 0 && c.lane < R) alpha[(unsigned)(n * R + c.lane)] = wgt;
  float acc[CH][VEC];
#pragma unroll
  for (int k = 0; k < CH; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[k][j] = 0.f;
#pragma unroll
  for (int u = 0; u < ATT_UB; ++u) {
    const int r = c.wave + u * NWAVE;
    float al = __shfl(wgt, r < R ? r : 0, 64);
    if (r >= R) al = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      float f[VEC];
      uic_unpack<T>(va[u][k], f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[k][j] += al * f[j];
    }
  }
#pragma unroll
  for (int k = 0; k < CH; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) s_red[c.wave * HH + (c.lane + 64 * k) * VEC + j] = acc[k][j];
  __syncthreads();
  {
    const int h = c.tid;             // NTH == HH
    float sacc = 0.f;
#pragma unroll
    for (int wv = 0; wv < NWAVE; ++wv) sacc += s_red[wv * HH + h];
    st_x<SAFE>(ctx + (unsigned)(n * HH + h), sacc);
  }
  __syncthreads();
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
    for (int rr = c.rank; rr < c.nrow; rr += PW)
      attn_row<T, SAFE>(c, p, c.rbegin + rr, att_h, p.alpha_all + (size_t)t * N * p.R, ctx);
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
  unsigned* sy = p.sync;
  int* info = (int*)smem;
  const int tid = threadIdx.x;
  if (tid == 0) {
    // Registration: every workgroup reports the XCD it actually runs on (hardware register, not blockIdx) and takes a rank
    // among that XCD's workgroups; once the whole grid has registered, all of them read the same eight counts and take
    // the same decision between the L2-local mode and the placement-independent SAFE mode.
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    const unsigned nb = gridDim.x;
    const unsigned xrank = xcc < 8u ? __hip_atomic_fetch_add(sy + SY_XCC + 32 * xcc, 1u, RLX_AGENT) : 0u;
    const unsigned ticket = __hip_atomic_fetch_add(sy + SY_TOTAL, 1u, RLX_AGENT);
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(sy + SY_TOTAL, RLX_AGENT) < nb) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_MAX) {
        __hip_atomic_store(sy + SY_ERR, 0x200u, RLX_AGENT);
        if (p.status) __hip_atomic_store(p.status, 0x200u, RLX_AGENT);
        ok = 0;
        break;
      }
    }
    bool fast = ok && xcc < 8u && nb == 8u * PW && !p.force_safe;
    for (int i = 0; i < 8; ++i) fast = fast && __hip_atomic_load(sy + SY_XCC + 32 * i, RLX_AGENT) == (unsigned)PW;
    info[0] = fast ? (int)xcc : (int)(ticket / PW);
    info[1] = fast ? (int)xrank : (int)(ticket % PW);
    info[2] = fast ? 0 : 1;
    info[3] = ok;
  }
  __syncthreads();
  Ctx c;
  c.tid = tid; c.lane = tid & 63; c.wave = __builtin_amdgcn_readfirstlane(tid >> 6); c.l15 = c.lane & 15; c.lq = c.lane >> 4;
  c.group = __builtin_amdgcn_readfirstlane(info[0]);
  c.rank = __builtin_amdgcn_readfirstlane(info[1]);
  const int safe = __builtin_amdgcn_readfirstlane(info[2]);
  const int ok = __builtin_amdgcn_readfirstlane(info[3]);
  __syncthreads();
  if (!ok) return;
  const int G = gridDim.x / PW;
  const int Rg = (p.Nrows + G - 1) / G;
  c.u0 = c.rank * 16;
  c.rbegin = p.row0 + c.group * Rg;
  c.nrow = min(Rg, p.row0 + p.Nrows - c.rbegin);
  if (c.nrow <= 0) return;
  c.MT = (c.nrow + 15) >> 4;
  c.bar = sy + SY_BAR + 32 * c.group;
  c.err = sy + SY_ERR;
  c.status = p.status;
  c.bar_target = 0;
  c.smem = smem;
  c.dbg = nullptr; c.exp = p.exp;
  if (tid == 0 && blockIdx.x == 0 && p.status) __hip_atomic_fetch_add(p.status + (safe ? 2 : 1), 1u, RLX_AGENT);   // launches per protocol
  if (safe) run_steps<T, true>(p, c);
  else run_steps<T, false>(p, c);
}

unsigned* g_status[16] = {};   // caller-allocated sticky status words per device (uic_set_persistent_status)
int g_persist_mode = -1;     // -1: read UIC_PERSIST (default on), 0: off, 1: on, 2: on + force the SAFE protocol

}  // namespace

extern "C" int uic_set_persistent_rnn(int32_t mode) {
  UIC_REQUIRE(mode >= 0 && mode <= 2, "set_persistent_rnn: mode=%d must be 0 (off), 1 (on) or 2 (on, SAFE protocol)", mode);
  g_persist_mode = mode;
  return UIC_OK;
}

extern "C" int uic_set_persistent_status(void* status) {
  int dev = 0;
  UIC_TRY(uic_check_hip(hipGetDevice(&dev), "hipGetDevice"));
  UIC_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
  g_status[dev] = (unsigned*)status;
  return UIC_OK;
}

int uic_rnn_persist_mode() {
  if (g_persist_mode < 0) {
    const char* e = getenv("UIC_PERSIST");
    g_persist_mode = e ? atoi(e) : 1;
    if (g_persist_mode < 0 || g_persist_mode > 2) g_persist_mode = 1;
  }
  return g_persist_mode;
}

constexpr int MAX_SLABS = 8;       // launches of <= 640 caption rows each
size_t uic_rnn_persist_sync_bytes() { return (size_t)MAX_SLABS * SY_WORDS * 4; }

bool uic_rnn_persist_eligible(int dtype, int N, int H, int A, int R) {
  if (!uic_rnn_persist_mode()) return false;
  if (dtype != UIC_BF16 && dtype != UIC_F32) return false;
  if (H != HH || A != HH || R < 1 || R > ATT_UB * NWAVE || N < 1 || N > MAX_SLABS * 8 * 16 * MT_MAX) return false;
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
  static bool configured = false;
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)rnn_fwd_persist_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), "hipFuncSetAttribute(rnn persist)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)rnn_fwd_persist_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), "hipFuncSetAttribute(rnn persist)"));
    configured = true;
  }
  const int G = 8, cap = G * 16 * MT_MAX;     // caption rows one launch covers
  for (int r0 = 0; r0 < p0.N; r0 += cap) {
    UicRnnFwdParams p = p0;
    p.row0 = r0;
    p.Nrows = p0.N - r0 < cap ? p0.N - r0 : cap;
    p.force_safe = uic_rnn_persist_mode() == 2;
    {
      int dev = 0;
      UIC_TRY(uic_check_hip(hipGetDevice(&dev), "hipGetDevice"));
      p.status = dev >= 0 && dev < 16 ? g_status[dev] : nullptr;
    }
    p.sync = p0.sync + (size_t)(r0 / cap) * SY_WORDS;
    UIC_TRY(uic_check_hip(hipMemsetAsync(p.sync, 0, (size_t)SY_WORDS * 4, s), "hipMemsetAsync(rnn sync)"));
    if (p.dtype == UIC_BF16) hipLaunchKernelGGL(rnn_fwd_persist_kernel<bf16_t>, dim3(G * PW), dim3(NTH), LDS_BYTES, s, p);
    else hipLaunchKernelGGL(rnn_fwd_persist_kernel<float>, dim3(G * PW), dim3(NTH), LDS_BYTES, s, p);
    UIC_LAUNCH_CHECK("rnn_fwd_persist_kernel");
  }
  return UIC_OK;
}
