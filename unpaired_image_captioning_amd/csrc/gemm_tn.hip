// "TN" MFMA GEMM for gfx950 (bf16):  C[i, j] = sum_k A[k, i] * B[k, j]   -- both operands stored with the REDUCTION
// index as their row index.  This is the shape of every weight gradient of the hot path,
//     dW = dG^T X      (dG [rows, out], X [rows, in], rows = decode steps x captions or captions x regions),
// i.e. the backward of the nn.Linear / nn.LSTMCell call sites of P/models/AttModel.py:76-92,426-441,543.  With
// K-contiguous ("NT") kernels only, both operands had to be transposed through HBM first (64 transpose launches and
// 0.85 ms per training step); here the tiles are staged exactly as they lie in memory and the transposition happens in
// the LDS read: gfx950's ds_read_b64_tr_b16 hands every lane the 4 k-consecutive elements of ITS column, so two of them
// build the 8-element bf16 MFMA operand.
//
//  * 128 x 128 output tile per workgroup of 4 waves (2 x 2, 64 x 64 each = 2 x 2 MFMA 32x32x16 tiles), 64 k-rows per
//    round, double-buffered LDS (2 x 32 KB), operands staged with 16-byte global_load_lds DMA (no VGPR round trip).
//  * LDS image of a staged tile: [64 k][128 columns] rows of 256 B; 16-byte chunk ch of row r sits at
//    256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))) -- the XOR keeps the transposed reads bank-conflict free.  The
//    DMA writes LDS lane-linearly, so the XOR is applied on the SOURCE side: lane (row, slot) fetches chunk slot ^ f(row).
//  * B may be up to 4 column segments living in different matrices ([ctx | h_att | h_lang] of the LSTM weight
//    gradients): torch.cat is never materialised on this side either.
//  * Output: raw f32 partial tiles into slab[z][M][N] (split-K over blockIdx.z); uic_splitk_reduce_launch sums the
//    slices in a fixed order and scatters column ranges to their destinations (deterministic).
#include "uic_common.h"
#include <type_traits>
#include <stdlib.h>

thread_local int g_uic_tn_ring_off = 0;
thread_local int g_uic_knobs = 0;

namespace {

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;

// NS = LDS stages (32 KB each).  2: one K round in flight behind the one being multiplied, two workgroups per CU.  4: three
// rounds in flight with counted vmcnt waits, one workgroup per CU -- for grids of at most one workgroup per CU (the 4-step
// weight-gradient chunks: 192 tiles), where nothing else hides the global -> LDS latency of a round.
template <int NS>
__global__ __launch_bounds__(256) void uic_gemm_tn_kernel(const UicGemmTnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [NS][A 16 KB | B 16 KB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, r32 = lane & 31;

  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  // column segment of this tile (uniform): segments are multiples of 128 columns wide
  int sidx = 0, c0 = n0;
  while (sidx + 1 < p.nseg && c0 >= p.seg[sidx].ncols) { c0 -= p.seg[sidx].ncols; ++sidx; }
  const char* Bbase = (const char*)p.seg[sidx].B;
  const int ldb = p.seg[sidx].ldb, ncolsB = p.seg[sidx].ncols;

  // per-lane source pointers of this wave's 4 A and 4 B LDS-DMA instructions per round (4 k-rows of 256 B each)
  const char* srcA[4];
  const char* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 4 + (lane >> 4);
    const int f = ((r & 3) << 2) | ((r >> 2) & 3);
    const int ch = (lane & 15) ^ f;
    const int colA = min(m0 + ch * 8, ((p.M + 7) & ~7) - 8);   // (M % 8 != 0: the last chunk runs into the row's padding, lda >= M rounded up)
    const int colB = min(c0 + ch * 8, ncolsB - 8);
    srcA[i] = (const char*)p.A + ((size_t)r * p.lda + colA) * 2;
    srcB[i] = Bbase + ((size_t)r * ldb + colB) * 2;
  }
  const size_t strideA = (size_t)64 * p.lda * 2, strideB = (size_t)64 * ldb * 2;
  auto stage = [&](int kt, int buf) {
    char* dA = smem + buf * 32768 + wave * 4096;
    char* dB = dA + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (size_t)kt * strideA),
                                       (__attribute__((address_space(3))) void*)(dA + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + (size_t)kt * strideB),
                                       (__attribute__((address_space(3))) void*)(dB + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  int kt0 = 0, nt = p.K / 64;
  if (p.splitk > 1) {
    const int tps = (nt + p.splitk - 1) / p.splitk;
    kt0 = blockIdx.z * tps;
    nt = max(0, min(nt - kt0, tps));
  }

  // Transposed fragment reads.  16-lane group g of the wave: k half = g >> 1 (MFMA lanes 0-31 carry k 0..7, lanes 32-63
  // k 8..15 of a 16-k step), column block = g & 1 (columns 0-15 / 16-31 of the 32-wide operand tile).  Inside the group
  // lane 4q + pp supplies the address of block row q, columns 4pp..4pp+3 and RECEIVES column (lane & 15), rows 0..3.
  // Read h (0 / 1) covers k rows 4h .. 4h+3 of the lane's 8.  The swizzle term depends on (q, k half, h) only, so the
  // K step is an immediate offset (4096 B per 16 k-rows) and so is the stage buffer (32768 B).
  const int g = lane >> 4, khalf = g >> 1, colblk = g & 1, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  // (the DS offset field is 16 bits: buffers 2 and 3 of the 4-stage ring get their own base registers, 64 KB up)
  constexpr int NHI = NS / 2;
  unsigned adAx[NHI][2][2], adBx[NHI][2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = khalf * 8 + h * 4 + q;
      const int f = (q << 2) | (khalf * 2 + h);
      const int ca = ((wm * 64 + ti * 32 + colblk * 16) >> 3) + (pp >> 1);
      const int cb = ((wn * 64 + ti * 32 + colblk * 16) >> 3) + (pp >> 1);
#pragma unroll
      for (int hi = 0; hi < NHI; ++hi) {
        adAx[hi][ti][h] = lds0 + (unsigned)(hi * 65536) + (unsigned)(256 * r + 16 * (ca ^ f) + 8 * (pp & 1));
        adBx[hi][ti][h] = lds0 + (unsigned)(hi * 65536) + 16384u + (unsigned)(256 * r + 16 * (cb ^ f) + 8 * (pp & 1));
      }
    }

#define TN_RD(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define TN_ISSUE_A(S, KS, BO)                                                             \
  TN_RD(S##a0l, adAx[HI][0][0], BO + KS * 4096); TN_RD(S##a0h, adAx[HI][0][1], BO + KS * 4096);     \
  TN_RD(S##a1l, adAx[HI][1][0], BO + KS * 4096); TN_RD(S##a1h, adAx[HI][1][1], BO + KS * 4096)
#define TN_ISSUE_B(S, KS, BO)                                                             \
  TN_RD(S##b0l, adBx[HI][0][0], BO + KS * 4096); TN_RD(S##b0h, adBx[HI][0][1], BO + KS * 4096);     \
  TN_RD(S##b1l, adBx[HI][1][0], BO + KS * 4096); TN_RD(S##b1h, adBx[HI][1][1], BO + KS * 4096)
#define TN_WAIT(N, S)                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                         \
               : "+v"(S##a0l), "+v"(S##a0h), "+v"(S##a1l), "+v"(S##a1h), "+v"(S##b0l), "+v"(S##b0h), "+v"(S##b1l), "+v"(S##b1h)); \
  __builtin_amdgcn_sched_barrier(0)
#define TN_OP(lo, hi) __builtin_bit_cast(bf16x8, (u32x4v){lo.x, lo.y, hi.x, hi.y})
#define TN_MFMA4(S)                                                                                                          \
  do {                                                                                                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TN_OP(S##a0l, S##a0h), TN_OP(S##b0l, S##b0h), acc[0][0], 0, 0, 0);   \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TN_OP(S##a0l, S##a0h), TN_OP(S##b1l, S##b1h), acc[0][1], 0, 0, 0);   \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TN_OP(S##a1l, S##a1h), TN_OP(S##b0l, S##b0h), acc[1][0], 0, 0, 0);   \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TN_OP(S##a1l, S##a1h), TN_OP(S##b1l, S##b1h), acc[1][1], 0, 0, 0);   \
  } while (0)
  // two register sets (x: even K steps, y: odd); at most 12 LDS reads outstanding (lgkmcnt is a 4-bit counter)
  auto compute = [&](auto bufc) {
    constexpr int BO = (decltype(bufc)::value & 1) * 32768, HI = decltype(bufc)::value >> 1;
    u32x2 xa0l, xa0h, xa1l, xa1h, xb0l, xb0h, xb1l, xb1h, ya0l, ya0h, ya1l, ya1h, yb0l, yb0h, yb1l, yb1h;
    TN_ISSUE_A(x, 0, BO); TN_ISSUE_B(x, 0, BO);
    TN_ISSUE_A(y, 1, BO);
    TN_WAIT(4, x);
    TN_MFMA4(x);
    TN_ISSUE_B(y, 1, BO);
    TN_ISSUE_A(x, 2, BO);
    TN_WAIT(4, y);
    TN_MFMA4(y);
    TN_ISSUE_B(x, 2, BO);
    TN_ISSUE_A(y, 3, BO);
    TN_WAIT(4, x);
    TN_MFMA4(x);
    TN_ISSUE_B(y, 3, BO);
    TN_WAIT(0, y);
    TN_MFMA4(y);
  };
#undef TN_MFMA4
#undef TN_OP
#undef TN_WAIT
#undef TN_ISSUE_B
#undef TN_ISSUE_A
#undef TN_RD
  if constexpr (NS == 2) {
    if (nt > 0) stage(kt0, 0);
    for (int t = 0; t < nt; t += 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 1 < nt) stage(kt0 + t + 1, 1);
      compute(std::integral_constant<int, 0>{});
      if (t + 1 < nt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) stage(kt0 + t + 2, 0);
        compute(std::integral_constant<int, 1>{});
      }
    }
  } else {
    static_assert(NS == 4, "ring written for 4 stages");
    // Ring of 4 buffers, 3 rounds in flight.  A round's 8 LDS-DMA loads per lane are retired by a COUNTED vmcnt (16 = the two
    // younger rounds may still be in flight), then the barrier makes every wave's share visible; the round issued now goes
    // into the buffer read LAST round (all waves are past that read once they are past this barrier).
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) stage(kt0 + i, i);
    auto round = [&](int r, auto jc) {
      constexpr int J = decltype(jc)::value;
      const int younger = nt - 1 - r;        // rounds issued after r that are still allowed to be in flight: min(2, younger)
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (r + 3 < nt) stage(kt0 + r + 3, (J + 3) & 3);
      compute(std::integral_constant<int, J>{});
    };
    for (int t = 0; t < nt; t += 4) {
      round(t, std::integral_constant<int, 0>{});
      if (t + 1 < nt) round(t + 1, std::integral_constant<int, 1>{});
      if (t + 2 < nt) round(t + 2, std::integral_constant<int, 2>{});
      if (t + 3 < nt) round(t + 3, std::integral_constant<int, 3>{});
    }
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (p.splitk == 1 && p.ndst > 0) {   // straight to the destination(s)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + (wn * 2 + j) * 32 + r32;
        if (col >= p.N) continue;
        float* C = nullptr;
        int ldc = 0, cc = 0;
        for (int d = 0; d < p.ndst; ++d)
          if (col >= p.dst[d].col0 && col < p.dst[d].col0 + p.dst[d].ncols) { C = p.dst[d].C; ldc = p.dst[d].ldc; cc = col - p.dst[d].col0; }
        if (!C) continue;
        // (the accumulate test hoisted, 32-bit offsets where the destination allows, whole row tiles unchecked)
        const bool rows_ok = m0 + 128 <= p.M, small = (size_t)p.M * (size_t)ldc < ((size_t)1 << 31);
        if (small && rows_ok && !p.accumulate) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            C[(unsigned)(m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * (unsigned)ldc + (unsigned)cc] = acc[i][j][reg];
        } else if (small && rows_ok) {
          float old[16];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            old[reg] = C[(unsigned)(m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * (unsigned)ldc + (unsigned)cc];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            C[(unsigned)(m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * (unsigned)ldc + (unsigned)cc] = old[reg] + acc[i][j][reg];
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            if (row >= p.M) continue;
            float* o = C + (size_t)row * ldc + cc;
            *o = p.accumulate ? *o + acc[i][j][reg] : acc[i][j][reg];
          }
        }
      }
    return;
  }
  float* slab = p.slab + (size_t)blockIdx.z * p.M * p.N;
  if (m0 + 128 <= p.M && n0 + 128 <= p.N && (size_t)p.M * (size_t)p.N < ((size_t)1 << 31)) {   // whole tile: no tests, 32-bit offsets
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const unsigned ro = (unsigned)(m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * (unsigned)p.N + (unsigned)(n0 + wn * 64 + r32);
        slab[ro] = acc[i][0][reg];
        slab[ro + 32] = acc[i][1][reg];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + (wn * 2 + j) * 32 + r32;
      if (col >= p.N) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if (row < p.M) slab[(size_t)row * p.N + col] = acc[i][j][reg];
      }
    }
}

// blockIdx.y picks the destination (a uniform select: indexing an array of the by-value structs would put it in scratch)
__global__ void splitk_reduce_multi_kernel(const float* __restrict__ slab, int splitk, int M, int N, UicSlabDest d0, UicSlabDest d1,
                                           UicSlabDest d2, UicSlabDest d3, int accumulate) {
  const int k = blockIdx.y;
  float* const C = k == 0 ? d0.C : k == 1 ? d1.C : k == 2 ? d2.C : d3.C;
  const int ldc = k == 0 ? d0.ldc : k == 1 ? d1.ldc : k == 2 ? d2.ldc : d3.ldc;
  const int col0 = k == 0 ? d0.col0 : k == 1 ? d1.col0 : k == 2 ? d2.col0 : d3.col0;
  const int ncols = k == 0 ? d0.ncols : k == 1 ? d1.ncols : k == 2 ? d2.ncols : d3.ncols;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t total = (size_t)M * ncols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int row = (int)(i / ncols), c = (int)(i - (size_t)row * ncols);
    const float* src = slab + (size_t)row * N + col0 + c;
    float v = 0.f;
    for (int z = 0; z < splitk; ++z) v += src[(size_t)z * M * N];
    float* o = C + (size_t)row * ldc + c;
    *o = accumulate ? *o + v : v;
  }
}

// the same, four columns per lane and the row from the grid (no division, 16-byte accesses): every extent and leading dimension
// a multiple of 4 and 16-byte aligned pointers
__global__ void splitk_reduce_multi_vec_kernel(const float* __restrict__ slab, int splitk, int M, int N, UicSlabDest d0, UicSlabDest d1,
                                               UicSlabDest d2, UicSlabDest d3, int accumulate) {
  const int k = blockIdx.y;
  float* const C = k == 0 ? d0.C : k == 1 ? d1.C : k == 2 ? d2.C : d3.C;
  const int ldc = k == 0 ? d0.ldc : k == 1 ? d1.ldc : k == 2 ? d2.ldc : d3.ldc;
  const int col0 = k == 0 ? d0.col0 : k == 1 ? d1.col0 : k == 2 ? d2.col0 : d3.col0;
  const int ncols = k == 0 ? d0.ncols : k == 1 ? d1.ncols : k == 2 ? d2.ncols : d3.ncols;
  const size_t MN = (size_t)M * N;
  if (ncols % 4 || col0 % 4 || ldc % 4 || ((size_t)C & 15)) {      // a destination that cannot take 16-byte accesses (a bias column): one column per lane
    const int c1 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c1 >= ncols) return;
    for (int row = blockIdx.z; row < M; row += gridDim.z) {
      const float* src = slab + (size_t)row * N + col0 + c1;
      float v = 0.f;
      for (int z = 0; z < splitk; ++z) v += src[(size_t)z * MN];
      float* o = C + (size_t)row * ldc + c1;
      *o = accumulate ? *o + v : v;
    }
    return;
  }
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= ncols) return;
  for (int row = blockIdx.z; row < M; row += gridDim.z) {
    const float* src = slab + (size_t)row * N + col0 + c;
    float4 v = *(const float4*)src;
    for (int z = 1; z < splitk; ++z) {
      const float4 w = *(const float4*)(src + (size_t)z * MN);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    float4* o = (float4*)(C + (size_t)row * ldc + c);
    if (accumulate) { const float4 w = *o; v.x = w.x + v.x; v.y = w.y + v.y; v.z = w.z + v.z; v.w = w.w + v.w; }
    *o = v;
  }
}

}  // namespace

int uic_splitk_reduce_multi_launch(const float* slab, int splitk, int M, int N, const UicSlabDest* dst, int nd, int accumulate, hipStream_t s) {
  UIC_REQUIRE(nd >= 1 && nd <= 4, "splitk_reduce_multi: %d destinations", nd);
  UicSlabDest d[4] = {dst[0], dst[nd > 1 ? 1 : 0], dst[nd > 2 ? 2 : 0], dst[nd > 3 ? 3 : 0]};
  size_t cols = 0;
  for (int i = 0; i < nd; ++i) cols += dst[i].ncols;
  const bool vec = N % 4 == 0 && ((uintptr_t)slab & 15) == 0 && ((size_t)M * N) % 4 == 0;
  int maxt = 0;      // lanes a destination needs along its columns: ncols / 4, or ncols where it takes the one-column path
  for (int i = 0; i < nd; ++i) {
    const bool v4 = dst[i].ncols % 4 == 0 && dst[i].col0 % 4 == 0 && dst[i].ldc % 4 == 0 && ((uintptr_t)dst[i].C & 15) == 0;
    const int t = v4 ? dst[i].ncols / 4 : dst[i].ncols;
    maxt = t > maxt ? t : maxt;
  }
  if (vec && M > 0 && maxt > 0) {
    const int bt = maxt >= 256 ? 256 : ((maxt + 63) / 64) * 64;
    const unsigned gz = (unsigned)(M > 65535 ? 65535 : M);
    hipLaunchKernelGGL(splitk_reduce_multi_vec_kernel, dim3((unsigned)((maxt + bt - 1) / bt), (unsigned)nd, gz), dim3(bt), 0, s, slab, splitk, M, N,
                       d[0], d[1], d[2], d[3], accumulate);
    UIC_LAUNCH_CHECK("splitk_reduce_multi_vec");
    return UIC_OK;
  }
  size_t g = ((size_t)M * cols / nd + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3((unsigned)g, (unsigned)nd), dim3(256), 0, s, slab, splitk, M, N, d[0], d[1], d[2], d[3], accumulate);
  UIC_LAUNCH_CHECK("splitk_reduce_multi");
  return UIC_OK;
}

bool uic_gemm_tn_eligible(const UicGemmTnParams& p) {
  // (M need not be a multiple of 8: lda is, so the last 16-byte chunk of a k-row reads padding columns inside the row -- whatever
  // they hold only reaches output rows >= M, which are never stored.  The pivot NMT's generator: M = 50004 target words.)
  if (p.M < 128 || p.N < 128 || p.K < 64 || p.K % 64 != 0 || p.lda % 8 != 0 || p.lda < ((p.M + 7) & ~7)) return false;
  if (p.nseg < 1 || p.nseg > UIC_GEMM_TN_MAX_SEG || ((uintptr_t)p.A & 15)) return false;
  int n = 0;
  for (int i = 0; i < p.nseg; ++i) {
    if (p.seg[i].ncols % 128 != 0 || p.seg[i].ldb % 8 != 0 || ((uintptr_t)p.seg[i].B & 15) || !p.seg[i].B) return false;
    n += p.seg[i].ncols;
  }
  return n == p.N;
}

int uic_gemm_tn_launch(const UicGemmTnParams& p, hipStream_t s) {
  UIC_REQUIRE(uic_gemm_tn_eligible(p), "gemm_tn: shape M=%d N=%d K=%d not eligible (K %% 64, segment widths %% 128, 16-byte alignment)", p.M, p.N, p.K);
  UIC_REQUIRE(p.splitk >= 1 && (p.slab || (p.splitk == 1 && p.ndst > 0)), "gemm_tn: needs a slab (or direct destinations with splitk == 1)");
  static bool configured = false;
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_tn_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536),
                          "hipFuncSetAttribute(gemm tn)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_tn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072),
                          "hipFuncSetAttribute(gemm tn, 4 stages)"));
    configured = true;
  }
  dim3 grid((p.M + 127) / 128, (p.N + 127) / 128, p.splitk);
  const int ring_max = g_uic_tn_ring_off ? 0 : 256;     // workgroups: at most one per CU
  const long wgs = (long)grid.x * grid.y * grid.z;
  if (wgs <= ring_max && p.K / 64 / (p.splitk > 1 ? p.splitk : 1) >= 8)
    hipLaunchKernelGGL(uic_gemm_tn_kernel<4>, grid, dim3(256), 131072, s, p);
  else
    hipLaunchKernelGGL(uic_gemm_tn_kernel<2>, grid, dim3(256), 65536, s, p);
  UIC_LAUNCH_CHECK("uic_gemm_tn_kernel");
  return UIC_OK;
}
