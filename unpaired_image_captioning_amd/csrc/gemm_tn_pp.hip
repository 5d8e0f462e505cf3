// "TN" MFMA GEMM, second form (round 5):  C[i, j] = sum_k A[k, i] * B[k, j]  -- every weight gradient dW = dG^T X of the hot
// path (the autograd backward of the nn.Linear / nn.LSTMCell call sites P/models/AttModel.py:76-92,426-441,543 at
// P/trainer.py:173) -- on the schedule of gemm_pp.hip: 256 x 256 output tile, 64 reduction rows per K tile, ONE 8-wave
// workgroup per CU whose two 4-wave halves alternate between LDS-DMA / fragment reads and MFMA, three 16 KB staging units in
// flight across raw barriers with counted vmcnt.
//
// Why a second TN kernel.  uic_gemm_tn_kernel (gemm_tn.hip) is a 128 x 128 tile, 4 waves, two LDS stages with vmcnt(0) + barrier
// per K tile.  The weight-gradient grids of the step are small (208 tiles per 4-step chunk of the LSTM gradients): one such
// workgroup per CU, nothing hides the global -> LDS latency of a K tile (0.94 us per K tile against 0.21 us of MFMA issue:
// 14 % MFMA-busy, profiles/r04_v7_mfma_busy.txt), and the window it runs in -- beside the latency-bound BPTT chain -- is bound
// by CU time.  What counts there is CU-time per flop, not the launch's own latency: this kernel spends 64 KB of
// LDS traffic per 8.4 MFLOP (half the 128 x 128 tile's bytes per flop), keeps two K tiles of operands in flight behind the one
// being multiplied, and occupies a quarter as many CUs for the same problem, so the BPTT chain's kernels find free CUs.
//
// Geometry.  Both operands lie with the REDUCTION index as their row index; tiles are staged as they lie and the transposition
// happens in the LDS read (ds_read_b64_tr_b16, as in gemm_tn.hip).  A staging "unit" is [64 k][128 columns] bf16 = 16 KB with
// 256-byte rows, 16-byte chunk ch of row r at 256 r + 16 (ch ^ f(r)), f(r) = ((r & 3) << 2) | ((r >> 2) & 3): the transposed
// reads of one half-wave (4 k-rows x 4 chunks, two k-blocks) then cover all 64 banks once.  The LDS-DMA writes lane-linearly,
// so the XOR sits on the SOURCE side.  A K-tile buffer is four units, a0 = columns [0, 128) of the A tile, a1 = [128, 256), b0 /
// b1 the same of B: 64 KB, two buffers.  Waves 2 x 4: wave (wr, wc) owns output rows 64 wr + [0, 64) of a0 and of a1 and output
// columns 32 wc + [0, 32) of b0 and of b1 -- 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16, issued TRANSPOSED (B fragment
// first) so that a lane holds 4 consecutive COLUMNS of one output row (16-byte stores).  A K tile is four phases, one unit pair
// each: (a0,b0) (a0,b1) (a1,b1) (a1,b0), 8 MFMAs per wave and phase, fragment reads per phase 8 B + 16 A / 8 B / 16 A / none
// (two transposing 8-byte reads per operand).
//
// Schedule = gemm_pp.hip's (tile t in buffer t & 1; every phase = reads, ONE unit staged, barrier, MFMA x 8, barrier):
//   phase 1: read b0 a0 (t)  | stage a1 (t+1)        phase 5: read b0 a0 (t+1) | stage a1 (t+2)
//   phase 2: read b1         | stage b0 (t+2)        phase 6: read b1          | stage b0 (t+3)
//   phase 3: read a1         | stage a0 (t+2)        phase 7: read a1          | stage a0 (t+3)
//   phase 4:                 | stage b1 (t+2), vmcnt(6)    phase 8:            | stage b1 (t+3), vmcnt(6)
// Every wave issues two DMA instructions per unit, so vmcnt(6) leaves the three youngest units in flight exactly as there; the
// RAW / WAR distances are those of gemm_pp.hip (its header derives them, with the one-barrier stagger of waves 4-7).  b0 is
// restaged one phase after its reads: phases 1 / 5 wait for their B reads (issued first) before the phase's first barrier.
// lgkmcnt is a 4-bit counter: phases 1 / 5 issue 8 B + 7 A reads, wait lgkmcnt(6) (all B reads and one A read are back), then
// the other 9 A reads -- never more than 15 in flight.
//
// B may be up to 4 column segments living in different matrices (multiples of 128 columns: every unit lies in ONE segment); the
// epilogue writes (or adds to) up to 4 destinations directly, or stores raw f32 partial tiles into slab[z][M][N] for the
// deterministic split-K reduction (uic_splitk_reduce_multi_launch).
#include "uic_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(2))) unsigned u32x2t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4t;

#define TP_RD(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <bool SLAB>
__global__ __launch_bounds__(512) void uic_gemm_tnpp_kernel(const UicGemmTnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][a0 | a1 | b0 | b1] x 16 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware grouped tile order (as uic_gemm_pp_kernel): each XCD takes a contiguous run of tiles, 8 row tiles per column step
  int bm, bn;
  {
    const int gx = gridDim.x, gy = gridDim.y, nblk = gx * gy;
    const int lin = blockIdx.x + gx * blockIdx.y;
    const int q = nblk >> 3, r = nblk & 7, xcd = lin & 7, idx = lin >> 3;
    const int lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr int GM = 8;
    const int width = GM * gy;
    const int first = (lp / width) * GM;
    const int gsz = min(gx - first, GM);
    const int rem = lp % width;
    bm = first + rem % gsz;
    bn = rem / gsz;
  }
  const int m0 = bm * 256, n0 = bn * 256;

  // K tiles of this workgroup: all of them, or slice blockIdx.z of a split-K launch (an even number each; the launcher checks)
  int kt0 = 0, nt = p.K / 64;
  if (p.splitk > 1) {
    const int tps = nt / p.splitk;
    kt0 = blockIdx.z * tps;
    nt = tps;
  }

  // ---- staging: wave-uniform source bases per unit (SGPR pairs, advanced by one K tile after every use), per-lane 32-bit byte
  // offsets, wave-uniform LDS destinations.  Instruction i of wave w covers k-rows 4 (2 w + i) .. + 3 of a unit: lane -> row +
  // (lane >> 4), LDS slot lane & 15 of that row, which holds chunk slot ^ f(row).
  const char* baseA[2];
  const char* baseB[2];
  unsigned offA[2][2], offB[2][2];
  size_t strideB[2];
  const size_t strideA = (size_t)64 * p.lda * 2;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    // the unit's 128 columns of B: one segment (segment widths are multiples of 128); a unit past the last column re-reads the
    // last one (its products are never stored)
    int c0 = min(n0 + 128 * s, p.N - 128), sidx = 0;
    while (sidx + 1 < p.nseg && c0 >= p.seg[sidx].ncols) { c0 -= p.seg[sidx].ncols; ++sidx; }
    const int ldb = p.seg[sidx].ldb;
    strideB[s] = (size_t)64 * ldb * 2;
    baseA[s] = (const char*)p.A + (size_t)kt0 * strideA;
    baseB[s] = (const char*)p.seg[sidx].B + (size_t)kt0 * strideB[s] + (size_t)c0 * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 4 + (lane >> 4);
      const int f = ((r & 3) << 2) | ((r >> 2) & 3);
      const int ch = (lane & 15) ^ f;
      // (M % 8 != 0: the last chunk runs into the row's padding, lda >= M rounded up to 8; rows past M are never stored)
      const int colA = min(m0 + 128 * s + ch * 8, ((p.M + 7) & ~7) - 8);
      offA[s][i] = (unsigned)((r * p.lda + colA) * 2);           // (< 64 rows x lda x 2 bytes: the launcher checks lda)
      offB[s][i] = (unsigned)((r * ldb + ch * 8) * 2);
    }
  }
  const unsigned dst0 = (unsigned)(wave * 2048);    // this wave's first instruction inside a unit; the second is 1 KB up
#define TP_GLDS(BASE, OFF, DST)                                                                                           \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((BASE) + (size_t)(OFF)),              \
                                   (__attribute__((address_space(3))) void*)(smem + (DST)), 16, 0, 0)
#define TP_STAGE_A(S, BUF)                                                                                   \
  do {                                                                                                      \
    TP_GLDS(baseA[S], offA[S][0], (BUF) * 65536u + (S) * 16384u + dst0);                                    \
    TP_GLDS(baseA[S], offA[S][1], (BUF) * 65536u + (S) * 16384u + dst0 + 1024u);                            \
    baseA[S] += strideA;                                                                                    \
  } while (0)
#define TP_STAGE_B(S, BUF)                                                                                   \
  do {                                                                                                      \
    TP_GLDS(baseB[S], offB[S][0], (BUF) * 65536u + 32768u + (S) * 16384u + dst0);                           \
    TP_GLDS(baseB[S], offB[S][1], (BUF) * 65536u + 32768u + (S) * 16384u + dst0 + 1024u);                   \
    baseB[S] += strideB[S];                                                                                 \
  } while (0)
  // all but the three youngest units (always b0, a0, b1 where the schedule waits) have landed
#define TP_WAIT_UNITS asm volatile("s_waitcnt vmcnt(6)" ::: "memory")

  // ---- transposed fragment reads (gemm_tn.hip's lane map): 16-lane group g: k half = g >> 1, column block = g & 1 of the
  // 32-wide operand tile; inside the group lane 4 q + pp supplies the address of block row q, columns 4 pp .. + 3 and RECEIVES
  // column (lane & 15), rows 0 .. 3.  Read h covers k rows 4 h .. + 3 of the lane's 8.  The swizzle term depends on (q, k half, h)
  // only: the 16-row K step (4096 B) and the unit (16 KB) are immediate offsets; the second buffer (64 KB up: beyond the 16-bit
  // offset field) has address registers of its own.
  const int g = lane >> 4, khalf = g >> 1, colblk = g & 1, q = (lane & 15) >> 2, pq = lane & 3;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  unsigned adA[2][2][2], adB[2][2];      // [buffer][32-row tile of the sub-half][h], [buffer][h]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = khalf * 8 + h * 4 + q;
    const int f = (q << 2) | (khalf * 2 + h);
    const int cb = ((wc * 32 + colblk * 16) >> 3) + (pq >> 1);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        const int ca = ((wr * 64 + ti * 32 + colblk * 16) >> 3) + (pq >> 1);
        adA[b][ti][h] = lds0 + (unsigned)(b * 65536) + (unsigned)(256 * r + 16 * (ca ^ f) + 8 * (pq & 1));
      }
      adB[b][h] = lds0 + (unsigned)(b * 65536) + 32768u + (unsigned)(256 * r + 16 * (cb ^ f) + 8 * (pq & 1));
    }
  }

  f32x16 acc[4][2];      // [32-row tile: 2 S + ti][column half S']
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  // operand halves: [.][K step][h] -- (lo, hi) of one K step form the 8-element bf16 MFMA operand
  u32x2t a[2][4][2], b0[4][2], b1[4][2];

  // A reads of sub-half S from buffer BUF, tile TI, K steps [KS0, KS1)
#define TP_READ_A1(S, BUF, TI, KS)                                                       \
  do {                                                                                   \
    TP_RD(a[TI][KS][0], adA[BUF][TI][0], (S) * 16384 + (KS) * 4096);                     \
    TP_RD(a[TI][KS][1], adA[BUF][TI][1], (S) * 16384 + (KS) * 4096);                     \
  } while (0)
#define TP_READ_B(BX, S, BUF)                                                            \
  do {                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                   \
      TP_RD(BX[ks][0], adB[BUF][0], (S) * 16384 + ks * 4096);                            \
      TP_RD(BX[ks][1], adB[BUF][1], (S) * 16384 + ks * 4096);                            \
    }                                                                                    \
  } while (0)
#define TP_READ_A_ALL(S, BUF)                                                            \
  do {                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 3; ++ks) {                                   \
      TP_READ_A1(S, BUF, 0, ks); TP_READ_A1(S, BUF, 1, ks);                              \
    }                                                                                    \
    TP_READ_A1(S, BUF, 0, 3);                                                            \
    TP_RD(a[1][3][0], adA[BUF][1][0], (S) * 16384 + 3 * 4096);                           \
    asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");      /* (16 reads: never more than 15 in flight) */ \
    TP_RD(a[1][3][1], adA[BUF][1][1], (S) * 16384 + 3 * 4096);                           \
  } while (0)
  // phases 1 / 5: 8 B reads, 7 A reads, wait until at most 6 are out (every B read is back: b0 may be restaged next phase),
  // then the other 9 A reads -- at most 15 reads in flight (lgkmcnt is 4 bits wide)
#define TP_READ_B_A(BX, BUF)                                                             \
  do {                                                                                   \
    TP_READ_B(BX, 0, BUF);                                                               \
    TP_READ_A1(0, BUF, 0, 0); TP_READ_A1(0, BUF, 1, 0); TP_READ_A1(0, BUF, 0, 1);        \
    TP_RD(a[1][1][0], adA[BUF][1][0], 1 * 4096);                                         \
    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");                                   \
    TP_RD(a[1][1][1], adA[BUF][1][1], 1 * 4096);                                         \
    TP_READ_A1(0, BUF, 0, 2); TP_READ_A1(0, BUF, 1, 2);                                  \
    TP_READ_A1(0, BUF, 0, 3); TP_READ_A1(0, BUF, 1, 3);                                  \
  } while (0)
#define TP_TIE_A                                                                                                      \
  do {                                                                                                                \
    _Pragma("unroll") for (int ti = 0; ti < 2; ++ti)                                                                  \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(a[ti][ks][0]), "+v"(a[ti][ks][1]));     \
  } while (0)
#define TP_TIE_B(BX)                                                                                                  \
  do {                                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(BX[ks][0]), "+v"(BX[ks][1]));             \
  } while (0)
  // all LDS reads of the phase have returned; the ties make the MFMAs below depend on this point (hipcc would otherwise hoist a
  // register-only MFMA over an inline-asm wait), the sched_barrier keeps the machine scheduler from moving them back up
#define TP_WAIT_READS(BX)                                  \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    TP_TIE_A; TP_TIE_B(BX);                                \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
#define TP_OP(X) __builtin_bit_cast(bf16x8, (u32x4t){X[0].x, X[0].y, X[1].x, X[1].y})
#define TP_MFMA(RT0, CT, BX)                                                                                                    \
  do {                                                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                            \
      _Pragma("unroll") for (int ti = 0; ti < 2; ++ti)                                                                          \
        acc[(RT0) + ti][CT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TP_OP(BX[ks]), TP_OP(a[ti][ks]), acc[(RT0) + ti][CT], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                          \
  } while (0)
#define TP_BARRIER do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

  // ---- prologue: tile 0 whole, tile 1 less its a1 unit (phase 1 stages that one)
  TP_STAGE_B(0, 0); TP_STAGE_A(0, 0); TP_STAGE_B(1, 0); TP_STAGE_A(1, 0);
  TP_STAGE_B(0, 1); TP_STAGE_A(0, 1); TP_STAGE_B(1, 1);
  TP_WAIT_UNITS;
  TP_BARRIER;
  if (wr == 1) TP_BARRIER;                 // waves 4-7 run one barrier behind from here on

  auto body = [&](auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    // ---------------- K tile in buffer 0
    // phase 1
    TP_READ_B_A(b0, 0);
    TP_STAGE_A(1, 1);
    TP_BARRIER;
    TP_WAIT_READS(b0);
    TP_MFMA(0, 0, b0);
    TP_BARRIER;
    // phase 2
    TP_READ_B(b1, 1, 0);
    if constexpr (!LAST) TP_STAGE_B(0, 0);
    TP_BARRIER;
    TP_WAIT_READS(b1);
    TP_MFMA(0, 1, b1);
    TP_BARRIER;
    // phase 3
    TP_READ_A_ALL(1, 0);
    if constexpr (!LAST) TP_STAGE_A(0, 0);
    TP_BARRIER;
    TP_WAIT_READS(b1);
    TP_MFMA(2, 1, b1);
    TP_BARRIER;
    // phase 4
    if constexpr (!LAST) {
      TP_STAGE_B(1, 0);
      TP_WAIT_UNITS;
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TP_BARRIER;
    TP_MFMA(2, 0, b0);
    TP_BARRIER;
    // ---------------- K tile in buffer 1
    // phase 5
    TP_READ_B_A(b0, 1);
    if constexpr (!LAST) TP_STAGE_A(1, 0);
    TP_BARRIER;
    TP_WAIT_READS(b0);
    TP_MFMA(0, 0, b0);
    TP_BARRIER;
    // phase 6
    TP_READ_B(b1, 1, 1);
    if constexpr (!LAST) TP_STAGE_B(0, 1);
    TP_BARRIER;
    TP_WAIT_READS(b1);
    TP_MFMA(0, 1, b1);
    TP_BARRIER;
    // phase 7
    TP_READ_A_ALL(1, 1);
    if constexpr (!LAST) TP_STAGE_A(0, 1);
    TP_BARRIER;
    TP_WAIT_READS(b1);
    TP_MFMA(2, 1, b1);
    TP_BARRIER;
    // phase 8
    if constexpr (!LAST) {
      TP_STAGE_B(1, 1);
      TP_WAIT_UNITS;
    }
    TP_BARRIER;
    TP_MFMA(2, 0, b0);
    TP_BARRIER;
  };
  for (int it = 0; it < nt / 2 - 1; ++it) body(std::false_type{});
  body(std::true_type{});
  if (wr == 0) TP_BARRIER;                 // (every wave executes the same number of barriers)

  // ---- epilogue.  D of the transposed 32 x 32 call: lane holds output row (lane & 31) of its tile and the columns
  // 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3): register quad j = four consecutive columns
  const int r32 = lane & 31, hl = lane >> 5;
  if constexpr (SLAB) {
    float* slab = p.slab + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int row = m0 + 128 * (rt >> 1) + 64 * wr + 32 * (rt & 1) + r32;
      if (row >= p.M) continue;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = n0 + 128 * ct + 32 * wc + 8 * j + 4 * hl;
          if (col < p.N)
            *(f32x4*)(slab + (size_t)row * p.N + col) = f32x4{acc[rt][ct][4 * j], acc[rt][ct][4 * j + 1], acc[rt][ct][4 * j + 2], acc[rt][ct][4 * j + 3]};
        }
    }
  } else {
    const bool accum = p.accumulate != 0;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + 128 * ct + 32 * wc + 8 * j + 4 * hl;
        if (col >= p.N) continue;
        // the destination of this column quad (destinations start at multiples of 4 columns; a destination narrower than the
        // quad -- the bias column of a ones segment -- takes its leading columns only)
        float* C = nullptr;
        int ldc = 0, nv = 0;
        for (int d = 0; d < p.ndst; ++d)
          if (col >= p.dst[d].col0 && col < p.dst[d].col0 + p.dst[d].ncols) {
            C = p.dst[d].C + (col - p.dst[d].col0); ldc = p.dst[d].ldc;
            nv = min(4, p.dst[d].col0 + p.dst[d].ncols - col);
          }
        if (!C) continue;
        const bool vec = nv == 4 && (ldc & 3) == 0 && (((size_t)C) & 15) == 0;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = m0 + 128 * (rt >> 1) + 64 * wr + 32 * (rt & 1) + r32;
          if (row >= p.M) continue;
          float* o = C + (size_t)row * ldc;
          f32x4 v = f32x4{acc[rt][ct][4 * j], acc[rt][ct][4 * j + 1], acc[rt][ct][4 * j + 2], acc[rt][ct][4 * j + 3]};
          if (vec) {
            if (accum) v += *(const f32x4*)o;
            *(f32x4*)o = v;
          } else {
            for (int e = 0; e < nv; ++e) o[e] = accum ? o[e] + v[e] : v[e];
          }
        }
      }
  }
}

#undef TP_BARRIER
#undef TP_MFMA
#undef TP_OP
#undef TP_WAIT_READS
#undef TP_TIE_B
#undef TP_TIE_A
#undef TP_READ_B_A
#undef TP_READ_A_ALL
#undef TP_READ_B
#undef TP_READ_A1
#undef TP_WAIT_UNITS
#undef TP_STAGE_B
#undef TP_STAGE_A
#undef TP_GLDS
#undef TP_RD

}  // namespace

// whole pairs of 64-row K tiles per split-K slice, segment widths multiples of 128, 16-byte aligned rows, destinations starting
// at multiples of 4 columns
bool uic_gemm_tnpp_eligible(const UicGemmTnParams& p) {
  if (!uic_gemm_tn_eligible(p)) return false;
  const int sk = p.splitk > 1 ? p.splitk : 1;
  if (p.M < 128 || p.N < 128 || p.K % (128 * sk) != 0 || p.lda >= (1 << 24)) return false;   // (per-lane byte offsets inside a K tile are 32-bit)
  for (int i = 0; i < p.nseg; ++i)
    if (p.seg[i].ldb >= (1 << 24)) return false;
  if (sk > 1 || p.ndst == 0) return p.slab != nullptr && ((uintptr_t)p.slab & 15) == 0;
  for (int d = 0; d < p.ndst; ++d)
    if (p.dst[d].col0 % 4 != 0 || !p.dst[d].C) return false;
  return true;
}

int uic_gemm_tnpp_launch(const UicGemmTnParams& p, hipStream_t s) {
  UIC_REQUIRE(uic_gemm_tnpp_eligible(p), "gemm_tnpp: shape M=%d N=%d K=%d splitk=%d not eligible (K %% (128 splitk), segment widths %% 128, 16-byte alignment)",
              p.M, p.N, p.K, p.splitk);
  static bool configured = false;
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_tnpp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072), "hipFuncSetAttribute(gemm tnpp)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_tnpp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072), "hipFuncSetAttribute(gemm tnpp slab)"));
    configured = true;
  }
  const int sk = p.splitk > 1 ? p.splitk : 1;
  dim3 grid((p.M + 255) / 256, (p.N + 255) / 256, sk);
  if (sk > 1 || p.ndst == 0) hipLaunchKernelGGL(uic_gemm_tnpp_kernel<true>, grid, dim3(512), 131072, s, p);
  else hipLaunchKernelGGL(uic_gemm_tnpp_kernel<false>, grid, dim3(512), 131072, s, p);
  UIC_LAUNCH_CHECK("uic_gemm_tnpp_kernel");
  return UIC_OK;
}
