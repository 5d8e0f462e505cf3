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

}  // namespace
