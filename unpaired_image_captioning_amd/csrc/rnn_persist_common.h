// Shared protocol of the persistent recurrence kernels on gfx950 (rnn_persist.hip: the forward decode loop,
// rnn_bwd_persist.hip: its BPTT): row groups (one per XCD where the placement allows), bounded group barriers,
// loads / stores of data exchanged between workgroups inside one launch.  The visibility rules this relies on are
// stated in the header comment of rnn_persist.hip.
#pragma once
#include "uic_common.h"
#include "../../include/uic_hip.h"

namespace {

constexpr int PW = 32;              // workgroups per row group (= CUs per XCD); each owns HH / PW = 16 hidden units
constexpr int MT_MAX = 5;           // 16-row tiles per group
constexpr int HH = 16 * PW;         // rnn_size == att_hid_size == 512 (P/opts.py:45-46 defaults)
constexpr unsigned SPIN_MAX = 1u << 17;
constexpr int ATT_R = 40;           // regions the attention phase covers (8 waves x 5 or 4 waves x 10 per row)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// words of the sync block (each counter on a 128-byte line of its own)
enum { SY_TOTAL = 0, SY_ERR = 32, SY_XCC = 64, SY_BAR = 64 + 32 * 8, SY_WORDS = 64 + 32 * 8 + 32 * 8 };

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}
// SC1: exchanged data (read past the vector L1).  NT: streamed once per step (weights, region features): non-temporal, so
// that it does not push the exchanged rows -- which every workgroup of the group re-reads -- out of the XCD's L2.
template <bool SC1, bool NT = false>
__device__ __forceinline__ u32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, (SC1 ? 16 : 0) | (NT ? 2 : 0)));
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
  unsigned long long* dbg;
};

// Bounded group barrier in two halves, so that loads which do not depend on the exchange can be issued between them
// (their latency then passes while the workgroup waits for the others).
// arrive: every wave first drains its own stores (the payload must be in L2 / memory before the arrival is visible),
// then one lane announces the workgroup.
__device__ __forceinline__ void group_arrive(Ctx& c) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  c.bar_target += PW;
  if (c.tid == 0) __hip_atomic_fetch_add(c.bar, 1u, RLX_AGENT);
}
// wait: one lane polls.  Returns false after a timeout (uniform over the workgroup).  `flag` is an LDS word that nothing
// else uses between the two __syncthreads below.
__device__ __forceinline__ bool group_wait(Ctx& c, int* flag) {
  if (c.tid == 0) {
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
  const int ok = __builtin_amdgcn_readfirstlane(*flag);
  __syncthreads();
  return ok != 0;
}
// (the forward kernel's form: the flag word is the first word of its reduction buffer)
__device__ __forceinline__ bool group_barrier(Ctx& c) {
  group_arrive(c);
  return group_wait(c, (int*)c.smem);
}

// Registration + grouping.  P: a parameter block with sync / status / force_safe / row0 / Nrows.
// Returns 0 (leave), 1 (XCD-local protocol) or 2 (SAFE protocol).
template <typename P>
__device__ __forceinline__ int setup_ctx(const P& p, char* scratch, Ctx& c) {
  unsigned* sy = p.sync;
  int* info = (int*)scratch;
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
  c.tid = tid; c.lane = tid & 63; c.wave = __builtin_amdgcn_readfirstlane(tid >> 6); c.l15 = c.lane & 15; c.lq = c.lane >> 4;
  c.group = __builtin_amdgcn_readfirstlane(info[0]);
  c.rank = __builtin_amdgcn_readfirstlane(info[1]);
  const int safe = __builtin_amdgcn_readfirstlane(info[2]);
  const int ok = __builtin_amdgcn_readfirstlane(info[3]);
  __syncthreads();
  if (!ok) return 0;
  if (tid == 0 && blockIdx.x == 0 && p.status) __hip_atomic_fetch_add(p.status + (safe ? 2 : 1), 1u, RLX_AGENT);   // launches per protocol
  const int G = gridDim.x / PW;
  const int Rg = (p.Nrows + G - 1) / G;
  c.u0 = c.rank * 16;
  c.rbegin = p.row0 + c.group * Rg;
  c.nrow = min(Rg, p.row0 + p.Nrows - c.rbegin);
  if (c.nrow <= 0) return 0;
  c.MT = (c.nrow + 15) >> 4;
  c.bar = sy + SY_BAR + 32 * c.group;
  c.err = sy + SY_ERR;
  c.status = p.status;
  c.bar_target = 0;
  c.smem = scratch;
  c.dbg = nullptr;
  return safe ? 2 : 1;
}

// ---------------------------------------------------------------------------------------------------
// GEMM + LSTM-cell phase of the generic (weights re-read every step) persistent kernels: rnn_persist.hip's f32 / generic
// captioner recurrence and nmt_persist.hip's pivot decoder.
constexpr int NWAVE = 8;
constexpr int NTH = NWAVE * 64;
constexpr int HALF_T = 3;           // row tiles reduced per LDS pass (8 waves x 3 tiles x 4 gates x 1 KB = 96 KB)
constexpr int LDS_BYTES = NWAVE * HALF_T * 4 * 1024;

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
      fb[buf][g] = bload<false>(rb, (brow[g] * (unsigned)ldb[sg] + (unsigned)(c.lq * VEC)) * (unsigned)sizeof(T), kk);
#pragma unroll
    for (int i = 0; i < MT_MAX; ++i)
      if (i < c.MT) fa[buf][i] = bload<true>(ra, aoff[i], kk);
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
            st_x<SAFE>(h_drop + o, hd);                      // (the pivot decoder's next layer reads it in the same launch)
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

}  // namespace
