// Large NT GEMM, second form: 256 x 256 output tile, 128-byte K rows (64 bf16), ONE 8-wave workgroup per CU whose two 4-wave
// halves alternate between LDS-DMA / fragment reads and MFMA ("ping-pong"), three 16 KB staging units in flight across barriers
// with counted vmcnt.  C[M,N] = A[M,K] B[N,K]^T, bf16 operands, f32 accumulate -- the throughput GEMMs of the step with
// K >= 512 and M >= 2048: att_embed (P/models/AttModel.py:76-80,111-115), the batched input / logit / d x GEMMs.
//
// Why a second kernel (profiles/LOG.md, round 3): the 128 x 128 tile of uic_gemm_glds_kernel moves 32 KB through a CU's
// 64 B/clk L1 -> LDS path per 2.1 MFLOP -- exactly the ratio of the CU's MFMA peak to that path -- so no pipeline depth on that
// tile gets past ~14 % MFMA-busy; a 256 x 256 tile halves the bytes per flop.  One such workgroup fills a CU (128 KB LDS, 8
// waves x <= 256 registers), so nothing else hides its latencies: the schedule below does.
//
// Geometry.  Waves 2 (rows) x 4 (columns), each owns 128 x 64 of the tile as 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16
// (issued TRANSPOSED -- B fragment as the first operand -- so that a lane holds 4 consecutive COLUMNS of one output row: 8- / 16-
// byte stores).  A K step (one 64-deep "K tile") is four phases, one 64 x 32 quadrant of the wave's tile each: (a0,b0) (a0,b1)
// (a1,b1) (a1,b0), so the fragment reads per phase are 4 B + 8 A / 4 B / 8 A / none.  LDS: two K-tile buffers of 64 KB =
// A [256 rows][128 B] | B [256 rows][128 B], rows XOR-swizzled on the source side (chunk c of row r sits at c ^ ((r >> 1) & 7):
// conflict-free ds_read_b128 for the 16-row fragments, the LDS-DMA writes stay lane-linear).  A "unit" is what the 512 lanes
// stage with two global_load_lds_dwordx4 each: the 128 rows one phase's fragment reads cover over all waves (a0 = rows
// [0,64) + [128,192) of A, a1 the others; b0 = rows [0,32) + [64,96) + [128,160) + [192,224) of B, b1 the others).
//
// Schedule (tile t in buffer t & 1; every phase = reads, ONE unit staged, barrier, MFMA x 16, barrier):
//   phase 1: read b0 a0 (t)  | stage a1 (t+1)        phase 5: read b0 a0 (t+1) | stage a1 (t+2)
//   phase 2: read b1         | stage b0 (t+2)        phase 6: read b1          | stage b0 (t+3)
//   phase 3: read a1         | stage a0 (t+2)        phase 7: read a1          | stage a0 (t+3)
//   phase 4:                 | stage b1 (t+2), vmcnt(6)    phase 8:            | stage b1 (t+3), vmcnt(6)
// vmcnt(6) leaves the three youngest units in flight, so phase 8's wait retires all of tile t+2 (read from phase 1 on, one
// barrier later) and phase 4's all of tile t+1.  A region is restaged two phases after its last read -- b0 one phase after,
// legal because phase 1 / 5 wait lgkmcnt(8) for their four B reads BEFORE the phase's first barrier (reads are issued B first).
// Waves 4-7 run one barrier behind waves 0-3: while one half issues MFMAs the other issues its reads and DMAs on the same SIMDs.
// The RAW / WAR distances above hold with that stagger (a wait precedes the waiting wave's next barrier, which the reader has
// to pass; a reader's lgkmcnt(0) precedes its own MFMA barrier, which the restager of two phases later is behind).
//
// Row-tile height.  RT = 4 is the 256-row tile described above.  One workgroup fills a CU, so what a launch takes is set by
// the CU with the most tiles: RT = 3 (192 rows) and RT = 2 (128 rows) are the same pipeline with 3 / 2 row fragments per
// sub-half (12 / 8 MFMAs per phase, A units of 12 / 8 KB) for problems whose 256-row tile count falls badly against the 256
// CUs (att_embed, 23040 x 512: 180 tiles of 256 rows, 240 of 192; d xt, 10880 x 512: 86 tiles of 256 rows, 170 of 128).
//
// f32 A operand (AF32; att_embed, whose input -- the loader's region features -- arrives as f32): A cannot take the LDS-DMA path
// (it needs rounding), so an A unit is loaded into registers where the DMA was issued (two global_load_dwordx4 per eight-row
// piece), and COMMITTED three phases later: counted vmcnt wait, v_cvt_pk_bf16_f32 x 4, one ds_write_b128 per piece to the
// address the DMA would have written (lane-linear, source-side swizzle unchanged), lgkmcnt(0) before the phase's barrier.
// (Loading whole 128-byte lines per instruction -- floats [4c, 4c+4) and [32+4c, ..) per lane, two ds_write_b64 and two 8-byte
// copy stores per piece -- was built and measured: 87.8 us against 81.0 for this form at 23040 x 512 x 2048.)
// Loaded at phases 1 / 3 / 5 / 7, committed at 4 / 6 / 8 / 2, read at 7 / 1' / 3' / 5' as before; two register sets (a0 units,
// a1 units: 32 registers).  The column-0 workgroups also store the bf16 chunks (p.a_copy): the weight gradient's operand,
// which a separate 283 MB cast pass used to make.  vmcnt counts loads, stores and LDS-DMA together in issue order, so every
// even phase waits vmcnt(4 + nA + nS): the ops younger than the unit being committed are two B units (2 DMA each), one A unit
// (nA loads) and the previous commit's nS copy stores.
#include "uic_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4pp;

#define PP_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int RT, bool SLAB, bool AF32 = false>
__global__ __launch_bounds__(512) void uic_gemm_pp_kernel(const UicGemmParams p) {
  static_assert(!(AF32 && SLAB), "the f32-A path has no split-K form");
  static_assert(RT == 2 || RT == 3 || RT == 4, "row fragments per sub-half");
  constexpr int BM = 64 * RT;              // rows of the tile; a wave owns 32 RT of them, a sub-half is 16 RT
  constexpr unsigned ABYTES = (unsigned)BM * 128u;       // one K tile of A in LDS; B (256 rows) follows it
  constexpr unsigned BUFB = ABYTES + 32768u;             // one K-tile buffer: 64 / 56 / 48 KB -- the shorter tiles leave LDS for
                                                         // a co-resident workgroup of another stream (the BPTT chain's kernels)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 8 RT KB | B 32 KB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware grouped tile order (as uic_gemm_glds_kernel): each XCD takes a contiguous run of tiles, 8 row tiles per column step
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
  const int m0 = bm * BM, n0 = bn * 256;
  const UicGemmSeg sg = p.seg[0];

  // K tiles of this workgroup: all of them, or slice blockIdx.z of a split-K launch (an even number each; launch_pp checks)
  int kt0 = 0, nt = sg.K / 64;
  if (p.splitk > 1) {
    const int tps = nt / p.splitk;
    kt0 = blockIdx.z * tps;
    nt = tps;
  }

  // ---- staging: per-lane source pointers (advanced by one K tile after every use), wave-uniform LDS destinations.
  // An A unit is 4 RT eight-row pieces (one wave instruction each): two per wave at RT = 4, one at RT = 2; at RT = 3 waves 0-3
  // take two and waves 4-7 one (their vmcnt counts differ accordingly).  A B unit is always 16 pieces, two per wave.
  const char* srcA[2][2];
  const char* srcB[2][2];
  unsigned dstA[2][2], dstB[2][2];
  unsigned offA[2][2], offC[2][2], wrA[2][2];          // (AF32 only)
  f32x4 ra[2][2][2];                                   // (AF32 only) [unit a0 / a1][piece][16-byte half]: an A unit on its way to LDS
  const void* const baseA = sg.A;
  void* const baseC = p.a_copy;
  const bool two = RT == 4 || (RT == 3 && wr == 0);    // this wave stages two pieces per A unit
  // the bf16 image's stores are shared by the first two column tiles of a row tile: unit a0's rows by column 0, a1's by column 1
  const bool cp0 = AF32 && p.a_copy != nullptr && bn == 0;
  const bool cp1 = AF32 && p.a_copy != nullptr && bn == (gridDim.y > 1 ? 1 : 0);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int qa = RT == 4 ? wave * 2 + j : RT == 2 ? wave : (wave < 4 ? wave * 2 + j : 4 + wave);   // (RT == 3, waves 4-7: pieces 8-11)
      const int rowA0 = (qa / (2 * RT)) * (32 * RT) + s * (16 * RT) + (qa % (2 * RT)) * 8;
      const int rowA = rowA0 + (lane >> 3);
      const int qb = wave * 2 + j;
      const int rowB = (qb >> 2) * 64 + s * 32 + (qb & 3) * 8 + (lane >> 3);
      const int chA = (lane & 7) ^ ((rowA >> 1) & 7);
      const int chB = (lane & 7) ^ ((rowB >> 1) & 7);
      const int gm = min(m0 + rowA, p.M - 1);
      const int gn = min(n0 + rowB, p.N - 1);
      srcA[s][j] = (const char*)sg.A + ((size_t)gm * sg.lda + (size_t)kt0 * 64) * 2 + chA * 16;
      srcB[s][j] = (const char*)sg.B + ((size_t)gn * sg.ldb + (size_t)kt0 * 64) * 2 + chB * 16;
      dstA[s][j] = (unsigned)(rowA0 * 128);
      if constexpr (AF32) {   // byte offsets from the f32 matrix (loads) and from its bf16 image (copy stores): < 4 GB, launch_pp checks
        offA[s][j] = (unsigned)(((size_t)gm * sg.lda + (size_t)chA * 8) * 4);
        offC[s][j] = (unsigned)(((size_t)gm * p.ld_a_copy + (size_t)chA * 8) * 2);
        wrA[s][j] = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + dstA[s][j] + (unsigned)(lane * 16);
      }
      dstB[s][j] = ABYTES + (unsigned)(((qb >> 2) * 64 + s * 32 + (qb & 3) * 8) * 128);
    }
#define PP_GLDS(SRC, DST)                                                                                   \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),                   \
                                   (__attribute__((address_space(3))) void*)(smem + (DST)), 16, 0, 0)
#define PP_STAGE_A(S, BUF)                                                                                   \
  do {                                                                                                      \
    PP_GLDS(srcA[S][0], dstA[S][0] + (BUF) * BUFB); srcA[S][0] += 128;                                      \
    if (RT == 4 || (RT == 3 && wr == 0)) { PP_GLDS(srcA[S][1], dstA[S][1] + (BUF) * BUFB); srcA[S][1] += 128; }   \
  } while (0)
  // AF32: the A unit's global loads (registers), and its commit three phases later
#define PP_LOADA1(S, J)                                                                                                         \
  do {                                                                                                                          \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[S][J][0]) : "v"(offA[S][J]), "s"(baseA) : "memory");                 \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ra[S][J][1]) : "v"(offA[S][J]), "s"(baseA) : "memory");       \
    offA[S][J] += 256;                                                                                                          \
  } while (0)
#define PP_LOADA(S) do { PP_LOADA1(S, 0); if (two) PP_LOADA1(S, 1); } while (0)
#define PP_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
  // all but the EXTRA + nA (+ nS) youngest vector-memory operations are done
  // (ST: the previous commit stored its share of the image: those stores are among them)
#define PP_WAIT_A(EXTRA, ST)                                                               \
  do {                                                                                     \
    if (two) { if (ST) PP_VMCNT((EXTRA) + 6); else PP_VMCNT((EXTRA) + 4); }                \
    else     { if (ST) PP_VMCNT((EXTRA) + 3); else PP_VMCNT((EXTRA) + 2); }                \
  } while (0)
#define PP_COMMIT1(S, J, BUF, CP)                                                                                               \
  do {                                                                                                                          \
    asm volatile("" : "+v"(ra[S][J][0]), "+v"(ra[S][J][1]));                                                                     \
    u32x4pp pk;                                                                                                                 \
    pk[0] = uic_pack_bf16x2(ra[S][J][0][0], ra[S][J][0][1]); pk[1] = uic_pack_bf16x2(ra[S][J][0][2], ra[S][J][0][3]);             \
    pk[2] = uic_pack_bf16x2(ra[S][J][1][0], ra[S][J][1][1]); pk[3] = uic_pack_bf16x2(ra[S][J][1][2], ra[S][J][1][3]);             \
    *(__attribute__((address_space(3))) u32x4pp*)(size_t)(wrA[S][J] + (unsigned)(BUF) * BUFB) = pk;   /* (compiler-visible too) */  \
    /* (a compiler-visible store: as inline asm the machine does not know that pk is VMEM store data and lets the next piece's */ \
    /* conversions overwrite the registers while the store still reads them -- 32-64 wrong elements in 1 % of the launches     */ \
    /* beside a busy neighbour, found by a soak; the asm statements around it carry memory clobbers, so it stays in place and  */ \
    /* counts in vmcnt exactly where the waits expect it)                                                                      */ \
    if (CP) *(u32x4pp*)((char*)baseC + offC[S][J]) = pk;                                                                        \
    offC[S][J] += 128;                                                                                                          \
  } while (0)
  // (the wait is the caller's: PP_WAIT_A / PP_VMCNT)
#define PP_COMMIT_A(S, BUF)                                          \
  do {                                                               \
    PP_COMMIT1(S, 0, BUF, (S) ? cp1 : cp0);                          \
    if (two) PP_COMMIT1(S, 1, BUF, (S) ? cp1 : cp0);                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               \
  } while (0)
#define PP_STAGE_B(S, BUF) do { PP_GLDS(srcB[S][0], dstB[S][0] + (BUF) * BUFB); PP_GLDS(srcB[S][1], dstB[S][1] + (BUF) * BUFB); \
                                srcB[S][0] += 128; srcB[S][1] += 128; } while (0)
  // all but the three youngest units (always b0, a0, b1 where the schedule waits) have landed
#define PP_WAIT_UNITS                                                                      \
  do {                                                                                     \
    if (RT == 4 || (RT == 3 && wr == 0)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                  \
  } while (0)

  // ---- fragment reads: LDS byte address of this lane's row / 16-byte chunk per K step (32 deep), both buffers
  const int fr = lane & 15, kg = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  unsigned adA0[2], adB0[2], adA1[2], adB1[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned pc = (unsigned)(((ks * 4 + kg) ^ sw) * 16);
    adA0[ks] = lds0 + (unsigned)((wr * 32 * RT + fr) * 128) + pc;
    adB0[ks] = lds0 + ABYTES + (unsigned)((wc * 64 + fr) * 128) + pc;
    adA1[ks] = adA0[ks] + BUFB;
    adB1[ks] = adB0[ks] + BUFB;
  }

  f32x4 acc[2 * RT][4];
#pragma unroll
  for (int i = 0; i < 2 * RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4pp a[4][2], b0[2][2], b1[2][2];

#define PP_READ_A(S, AD)                                                                                    \
  do {                                                                                                      \
    PP_DSR(a[0][0], AD[0], (S) * 2048 * RT + 0);    PP_DSR(a[0][1], AD[1], (S) * 2048 * RT + 0);            \
    PP_DSR(a[1][0], AD[0], (S) * 2048 * RT + 2048); PP_DSR(a[1][1], AD[1], (S) * 2048 * RT + 2048);         \
    if constexpr (RT > 2) { PP_DSR(a[2][0], AD[0], (S) * 2048 * RT + 4096); PP_DSR(a[2][1], AD[1], (S) * 2048 * RT + 4096); } \
    if constexpr (RT > 3) { PP_DSR(a[3][0], AD[0], (S) * 2048 * RT + 6144); PP_DSR(a[3][1], AD[1], (S) * 2048 * RT + 6144); } \
  } while (0)
#define PP_READ_B(BX, S, AD)                                                                                \
  do {                                                                                                      \
    PP_DSR(BX[0][0], AD[0], (S) * 4096 + 0);    PP_DSR(BX[0][1], AD[1], (S) * 4096 + 0);                    \
    PP_DSR(BX[1][0], AD[0], (S) * 4096 + 2048); PP_DSR(BX[1][1], AD[1], (S) * 4096 + 2048);                 \
  } while (0)
#define PP_TIE_A                                                                                    \
  do {                                                                                              \
    asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]));                   \
    if constexpr (RT > 2) asm volatile("" : "+v"(a[2][0]), "+v"(a[2][1]));                           \
    if constexpr (RT > 3) asm volatile("" : "+v"(a[3][0]), "+v"(a[3][1]));                           \
  } while (0)
#define PP_TIE_B(BX) asm volatile("" : "+v"(BX[0][0]), "+v"(BX[0][1]), "+v"(BX[1][0]), "+v"(BX[1][1]))
  // all LDS reads of the phase have returned; the ties make the MFMAs below depend on this point (hipcc would otherwise hoist a
  // register-only MFMA over an inline-asm wait), the sched_barrier keeps the machine scheduler from moving them back up
#define PP_WAIT_READS(BX)                                  \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    PP_TIE_A; PP_TIE_B(BX);                                \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
#define PP_MFMA(RT0, CT0, BX)                                                                                                   \
  do {                                                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                            \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                         \
        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                        \
          acc[(RT0) + rt][(CT0) + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, BX[ct][ks]),         \
                                                                                __builtin_bit_cast(bf16x8, a[rt][ks]),          \
                                                                                acc[(RT0) + rt][(CT0) + ct], 0, 0, 0);          \
    __builtin_amdgcn_s_setprio(0);                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                          \
  } while (0)
  // the four B reads of a 4 + 2 RT read phase (issued first) have returned: their LDS region may be restaged next phase
#define PP_WAIT_B_READS                                                          \
  do {                                                                           \
    if constexpr (RT == 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");    \
    else if constexpr (RT == 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); \
    else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                      \
  } while (0)
#define PP_BARRIER do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

  // ---- prologue: tile 0 whole, tile 1 less its a1 unit (phase 1 stages that one)
  if constexpr (AF32) {
    // the same units, issued in the order an iteration's phases 5-8 would have issued them; tile 0's a0 is committed here, its
    // a1 and tile 1's a0 stay in their registers (phases 1 and 3 commit them)
    PP_LOADA(0);
    PP_STAGE_B(0, 0); PP_STAGE_B(1, 0);
    PP_LOADA(1);
    PP_STAGE_B(0, 1);
    PP_WAIT_A(6, false);                             // younger than tile 0's a0 loads: three B units + a1's loads
    PP_COMMIT_A(0, 0);
    PP_LOADA(0);
    PP_STAGE_B(1, 1);
    // tile 0's B units have landed: younger than them are a1's loads, b0 of tile 1, the image stores, a0's loads, b1 of tile 1
    if (two) { if (cp0) PP_VMCNT(14); else PP_VMCNT(12); } else { if (cp0) PP_VMCNT(9); else PP_VMCNT(8); }
  } else {
    PP_STAGE_B(0, 0); PP_STAGE_A(0, 0); PP_STAGE_B(1, 0); PP_STAGE_A(1, 0);
    PP_STAGE_B(0, 1); PP_STAGE_A(0, 1); PP_STAGE_B(1, 1);
    PP_WAIT_UNITS;
  }
  PP_BARRIER;
  if (wr == 1) PP_BARRIER;                 // waves 4-7 run one barrier behind from here on

  auto body = [&](auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    // ---------------- K tile in buffer 0
    // (AF32: every odd phase COMMITS the A unit loaded four phases earlier -- wait, round, ds_write, lgkmcnt(0) -- and then loads
    // the next unit of the same kind into the registers it freed; the commit's counted wait also retires the B units in time)
    // phase 1
    PP_READ_B(b0, 0, adB0); PP_READ_A(0, adA0);
    PP_WAIT_B_READS;
    if constexpr (AF32) { PP_WAIT_A(4, cp0); PP_COMMIT_A(1, 0); PP_LOADA(1); }      // a1 of this tile (buffer 0), read in phase 3
    else PP_STAGE_A(1, 1);
    PP_BARRIER;
    PP_WAIT_READS(b0);
    PP_MFMA(0, 0, b0);
    PP_BARRIER;
    // phase 2
    PP_READ_B(b1, 1, adB0);
    if constexpr (!LAST) PP_STAGE_B(0, 0);
    PP_BARRIER;
    PP_WAIT_READS(b1);
    PP_MFMA(0, 2, b1);
    PP_BARRIER;
    // phase 3
    PP_READ_A(1, adA0);
    if constexpr (AF32) {                                                            // a0 of the tile in buffer 1, read in phase 5
      if constexpr (LAST) PP_WAIT_A(2, cp1); else PP_WAIT_A(4, cp1);
      PP_COMMIT_A(0, 1);
      if constexpr (!LAST) PP_LOADA(0);
    } else if constexpr (!LAST) PP_STAGE_A(0, 0);
    PP_BARRIER;
    PP_WAIT_READS(b1);
    PP_MFMA(RT, 2, b1);
    PP_BARRIER;
    // phase 4
    if constexpr (!LAST) {
      PP_STAGE_B(1, 0);
      if constexpr (!AF32) PP_WAIT_UNITS;
    } else if constexpr (!AF32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER;
    PP_MFMA(RT, 0, b0);
    PP_BARRIER;
    // ---------------- K tile in buffer 1
    // phase 5
    PP_READ_B(b0, 0, adB1); PP_READ_A(0, adA1);
    PP_WAIT_B_READS;
    if constexpr (AF32) {                                                            // a1 of the tile in buffer 1, read in phase 7
      if constexpr (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else PP_WAIT_A(4, cp0);
      PP_COMMIT_A(1, 1);
      if constexpr (!LAST) PP_LOADA(1);
    } else if constexpr (!LAST) PP_STAGE_A(1, 0);
    PP_BARRIER;
    PP_WAIT_READS(b0);
    PP_MFMA(0, 0, b0);
    PP_BARRIER;
    // phase 6
    PP_READ_B(b1, 1, adB1);
    if constexpr (!LAST) PP_STAGE_B(0, 1);
    PP_BARRIER;
    PP_WAIT_READS(b1);
    PP_MFMA(0, 2, b1);
    PP_BARRIER;
    // phase 7
    PP_READ_A(1, adA1);
    if constexpr (!LAST) {
      if constexpr (AF32) { PP_WAIT_A(4, cp1); PP_COMMIT_A(0, 0); PP_LOADA(0); }    // a0 of the next tile for buffer 0, read in phase 1
      else PP_STAGE_A(0, 1);
    }
    PP_BARRIER;
    PP_WAIT_READS(b1);
    PP_MFMA(RT, 2, b1);
    PP_BARRIER;
    // phase 8
    if constexpr (!LAST) {
      PP_STAGE_B(1, 1);
      if constexpr (!AF32) PP_WAIT_UNITS;
    }
    PP_BARRIER;
    PP_MFMA(RT, 0, b0);
    PP_BARRIER;
  };
  for (int it = 0; it < nt / 2 - 1; ++it) body(std::false_type{});
  body(std::true_type{});
  if (wr == 0) PP_BARRIER;                 // (every wave executes the same number of barriers)

  // ---- epilogue: lane holds C[row = .. + fr][col = .. + kg * 4 + 0..3] per accumulator
  const int rbase = m0 + wr * 32 * RT + fr;
  const int cbase = n0 + wc * 64 + kg * 4;
  if constexpr (SLAB) {
    float* slab = p.slab + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int rt = 0; rt < 2 * RT; ++rt) {
      const int row = rbase + rt * 16;
      if (row >= p.M) continue;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int col = cbase + ct * 16;
        if (col < p.N) *(f32x4*)(slab + (size_t)row * p.N + col) = acc[rt][ct];
      }
    }
    return;
  } else {
    const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    const bool out_f32 = (p.flags & UIC_GEMM_OUT_F32) != 0;
    const bool relu = (p.flags & UIC_GEMM_RELU) != 0, drop = p.drop_p > 0.f, accum = (p.flags & UIC_GEMM_ACCUM) != 0;
    f32x4 bj[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      bj[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int col = cbase + ct * 16;
      if (col < p.N) {
        if (p.bias) bj[ct] += *(const f32x4*)(p.bias + col);
        if (p.bias2) bj[ct] += *(const f32x4*)(p.bias2 + col);
      }
    }
    auto emit = [&](auto relu_c, auto drop_c, auto f32_c) {
      constexpr bool RELU = decltype(relu_c)::value, DROP = decltype(drop_c)::value, F32 = decltype(f32_c)::value;
#pragma unroll
      for (int rt = 0; rt < 2 * RT; ++rt) {
        const int row = rbase + rt * 16;
        if (row >= p.M) continue;
        const size_t ro = (size_t)row * p.ldc;
        const unsigned dro = (unsigned)(row + p.drop_row0) * (unsigned)p.N;
        // per-row work of the rarer options: the addend's row, the region mask of pack_wrapper (AttModel.py:44-53)
        const float* ad = p.addend ? p.addend + (size_t)(row % p.add_mod) * p.ld_add : nullptr;
        bool live = true;
        if (p.row_len) { const int n = row / p.R; live = row - n * p.R < p.row_len[n]; }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int col = cbase + ct * 16;
          if (col >= p.N) continue;
          f32x4 v = acc[rt][ct] + bj[ct];
          if (ad) v += *(const f32x4*)(ad + col);
          if (p.acc_src) v += *(const f32x4*)(p.acc_src + (size_t)row * p.ld_acc_src + col);
          if (p.mask_act) {          // backward of ReLU (+ dropout scale) by the forward activation's sign
            const uint2 q = *(const uint2*)((const bf16_t*)p.mask_act + (size_t)row * p.ld_mask_act + col);
            v[0] = __uint_as_float(q.x << 16) > 0.f ? v[0] * p.mask_scale : 0.f;
            v[1] = __uint_as_float(q.x & 0xffff0000u) > 0.f ? v[1] * p.mask_scale : 0.f;
            v[2] = __uint_as_float(q.y << 16) > 0.f ? v[2] * p.mask_scale : 0.f;
            v[3] = __uint_as_float(q.y & 0xffff0000u) > 0.f ? v[3] * p.mask_scale : 0.f;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (RELU) v[j] = fmaxf(v[j], 0.f);
            if (!live) v[j] = 0.f;
            if (DROP) v[j] *= uic_drop_scale(p.seed, p.site, dro + (unsigned)(col + j), p.drop_p, inv_keep);
          }
          if (F32) {
            f32x4* o = (f32x4*)((float*)p.C + ro + col);
            if (accum) v += *o;
            *o = v;
          } else {
            uint2* o = (uint2*)((bf16_t*)p.C + ro + col);
            if (accum) {
              const uint2 q = *o;
              v[0] += __uint_as_float(q.x << 16); v[1] += __uint_as_float(q.x & 0xffff0000u);
              v[2] += __uint_as_float(q.y << 16); v[3] += __uint_as_float(q.y & 0xffff0000u);
            }
            const uint2 q2 = make_uint2(uic_pack_bf16x2(v[0], v[1]), uic_pack_bf16x2(v[2], v[3]));
            *o = q2;
            if (p.C_exp2) {
              const float e0 = __uint_as_float(q2.x << 16), e1 = __uint_as_float(q2.x & 0xffff0000u);
              const float e2 = __uint_as_float(q2.y << 16), e3 = __uint_as_float(q2.y & 0xffff0000u);
#define PP_E2(x) __builtin_amdgcn_exp2f(fminf(fmaxf((x) * 2.8853900817779268f, -UIC_E2_CLAMP), UIC_E2_CLAMP))
              *(uint2*)((bf16_t*)p.C_exp2 + ro + col) = make_uint2(uic_pack_bf16x2(PP_E2(e0), PP_E2(e1)), uic_pack_bf16x2(PP_E2(e2), PP_E2(e3)));
#undef PP_E2
            }
          }
        }
      }
    };
    using TT = std::true_type; using FF = std::false_type;
#define PP_EPI(R, D) do { if (out_f32) emit(R{}, D{}, TT{}); else emit(R{}, D{}, FF{}); } while (0)
    if (relu && drop) PP_EPI(TT, TT);
    else if (relu) PP_EPI(TT, FF);
    else if (drop) PP_EPI(FF, TT);
    else PP_EPI(FF, FF);
#undef PP_EPI
  }
}

#undef PP_BARRIER
#undef PP_WAIT_B_READS
#undef PP_WAIT_UNITS
#undef PP_MFMA
#undef PP_WAIT_READS
#undef PP_TIE_B
#undef PP_TIE_A
#undef PP_READ_B
#undef PP_READ_A
#undef PP_COMMIT_A
#undef PP_COMMIT1
#undef PP_WAIT_A
#undef PP_VMCNT
#undef PP_LOADA
#undef PP_LOADA1
#undef PP_STAGE_B
#undef PP_STAGE_A
#undef PP_GLDS
#undef PP_DSR

}  // namespace

// one K segment of whole pairs of 64-deep K tiles per slice, 16-byte aligned vector stores, no tanh / pre-dropout copy / LSTM epilogue
bool uic_gemm_pp_eligible(const UicGemmParams& p) {
  if (p.dtype != UIC_BF16 || p.nseg != 1 || p.lstm || p.C_pre || (p.flags & UIC_GEMM_TANH)) return false;
  const int K = p.seg[0].K;
  const int sk = p.splitk > 1 ? p.splitk : 1;
  if (K % (128 * sk) != 0 || K / sk < 128) return false;
  if (p.N % 4 != 0) return false;
  if (p.a_f32) {   // f32 A rounded in the kernel: 16-byte loads through 32-bit byte offsets, no split-K form
    if (sk > 1 || p.slab || p.seg[0].lda % 4 != 0 || ((uintptr_t)p.seg[0].A & 15) || (size_t)p.M * p.seg[0].lda * 4 >= ((size_t)1 << 32)) return false;
    if (p.a_copy && (p.ld_a_copy % 8 != 0 || ((uintptr_t)p.a_copy & 15) || p.ld_a_copy < K)) return false;
  } else if (p.a_copy) return false;
  if (p.slab) return ((uintptr_t)p.slab & 15) == 0;
  if (!p.C || p.ldc % 4 != 0) return false;
  const bool f32 = (p.flags & UIC_GEMM_OUT_F32) != 0;
  if (((uintptr_t)p.C & (f32 ? 15 : 7)) != 0) return false;
  if (p.C_exp2 && (f32 || ((uintptr_t)p.C_exp2 & 7) != 0)) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  if (p.bias2 && ((uintptr_t)p.bias2 & 15)) return false;
  if (p.addend && (((uintptr_t)p.addend & 15) || p.ld_add % 4 != 0)) return false;
  if (p.acc_src && (((uintptr_t)p.acc_src & 15) || p.ld_acc_src % 4 != 0)) return false;
  if (p.mask_act && (((uintptr_t)p.mask_act & 7) || p.ld_mask_act % 4 != 0)) return false;
  return true;
}

namespace {
template <int RT>
int launch_pp(const UicGemmParams& p, hipStream_t s) {
  static bool configured = false;
  constexpr int lds = 2 * (64 * RT * 128 + 32768);
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_pp_kernel<RT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(gemm pp)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_pp_kernel<RT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(gemm pp slab)"));
    if constexpr (RT < 4)
      UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_pp_kernel<RT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(gemm pp f32 A)"));
    configured = true;
  }
  dim3 grid((p.M + 64 * RT - 1) / (64 * RT), (p.N + 255) / 256, p.splitk > 1 ? p.splitk : 1);
  if (p.a_f32) {
    // (the 256-row tile has no registers left for the two A units in flight: 248 + 32 -- it would spill registers whose loads
    // are still in flight)
    if constexpr (RT < 4) hipLaunchKernelGGL((uic_gemm_pp_kernel<RT, false, true>), grid, dim3(512), lds, s, p);
    else UIC_REQUIRE(false, "gemm_pp: the f32-A path has 192- and 128-row tiles only");
  } else if (p.slab) hipLaunchKernelGGL((uic_gemm_pp_kernel<RT, true>), grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL((uic_gemm_pp_kernel<RT, false>), grid, dim3(512), lds, s, p);
  UIC_LAUNCH_CHECK("uic_gemm_pp_kernel");
  return UIC_OK;
}
}  // namespace

// rows: the tile height, 256 / 192 / 128 (0 = uic_gemm_pp_rows' choice)
int uic_gemm_pp_launch(const UicGemmParams& p, int rows, hipStream_t s) {
  UIC_REQUIRE(uic_gemm_pp_eligible(p), "gemm_pp: problem not eligible for the 256-column ping-pong kernel");
  if (rows == 0) rows = uic_gemm_pp_rows(p.M, p.N, p.a_f32 ? 192 : 256, p.seg[0].K / (p.splitk > 1 ? p.splitk : 1));
  UIC_REQUIRE(rows == 256 || rows == 192 || rows == 128, "gemm_pp: tile height %d (256 / 192 / 128)", rows);
  return rows == 256 ? launch_pp<4>(p, s) : rows == 192 ? launch_pp<3>(p, s) : launch_pp<2>(p, s);
}

// The tile height whose slowest CU has the least work: one workgroup per CU and 256 CUs, so a launch of `tiles` tiles takes
// ceil(tiles / 256) rounds of one tile's time -- its K loop (~ rows x K tiles) plus what every tile pays regardless of its height
// (the 256-column B operand, prologue, epilogue: ~ 96 rows' worth of an 8-K-tile loop, fitted to the vocabulary GEMMs with K =
// 512: 1984 x 50004 takes 185 us with 256-row tiles, 187 with 192, 225 with 128 -- the count of rounds alone picked 128).
// Ties go to the taller tile (fewer operand bytes per flop).
int uic_gemm_pp_rows(int M, int N, int tallest, int K) {
  const long cols = (N + 255) / 256;
  const long ktiles = K > 64 ? K / 64 : 1;
  int best = tallest; long best_cost = -1;
  for (int rows = tallest; rows >= 128; rows -= 64) {
    const long tiles = (long)((M + rows - 1) / rows) * cols;
    const long cost = ((tiles + 255) / 256) * ((long)rows * ktiles + 768);
    if (best_cost < 0 || cost < best_cost) { best = rows; best_cost = cost; }
  }
  return best;
}
