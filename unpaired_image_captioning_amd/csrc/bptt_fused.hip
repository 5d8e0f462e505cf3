// One launch for two links of the BPTT chain (Step::bwd_step, topdown.hip): the `h2att` input-gradient GEMM
//     d h_att (+)= d att_h W_h2att                     (backward of P/models/AttModel.py:543)
// and the att_lstm cell backward that consumes it (backward of nn.LSTMCell, :434), whose other d h sources are the split-K
// partial slabs of the step's d x2 GEMM and of the previous step's d x1 GEMM.  The BPTT loop is a chain of dependent launches
// (launch + first-load latency each, ~3.5 us between two of them): the GEMM is 0.34 GFLOP and the cell math 0.3 M cells, so both
// are made of latency, and what pays is requesting EVERYTHING at once -- GEMM fragments, saved gates, cell states, partial
// slabs -- and having enough workgroups for the cell operands (64 x 64 tiles would leave them to 80 workgroups; earlier epilogue
// fusions lost for exactly that reason).
//   * 32 rows x 32 hidden units per workgroup of 4 waves (320 workgroups at 640 x 512), K = att_hid_size split over the waves
//     (wave w: K slice w), operands global -> VGPR in MFMA 16x16x32 fragment layout, partial tiles summed through 16 KB of LDS;
//   * the cell backward runs in a row-major layout (thread = one row x 4 consecutive units: 8- / 16-byte accesses), all of its
//     operands requested before the GEMM's first MFMA.
// bf16 only; other dtypes / shapes keep the two launches.
#include "uic_common.h"

namespace {

constexpr int FT = 32;                       // tile: FT rows x FT units
typedef __attribute__((ext_vector_type(4))) unsigned u32x4f;

__device__ __forceinline__ f32x4 mma16(const u32x4f& a, const u32x4f& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int KS>     // k-steps (32 K elements) per wave: A = 4 waves x KS x 32
__global__ __launch_bounds__(256) void h2att_cell_bwd_kernel(const UicH2attCellParams p) {
  __shared__ __attribute__((aligned(16))) float red[4][FT * FT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int row0 = blockIdx.x * FT, u0 = blockIdx.y * FT;
  const int H = p.H, A = p.A;
  const int crow = row0 + (tid >> 3), cu = u0 + 4 * (tid & 7);
  const bool live = crow < p.N;
  const int rr = live ? crow : p.N - 1;
  const size_t idx = (size_t)rr * H + cu;
  // ---- GEMM fragments of this wave's K slice
  const bf16_t* Ab = (const bf16_t*)p.datth + (size_t)wave * KS * 32 + lq * 8;
  const bf16_t* Bb = (const bf16_t*)p.h2attT + (size_t)wave * KS * 32 + lq * 8;
  u32x4f fa[2][KS], fb[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ar = min(row0 + 16 * i + l15, p.N - 1);
#pragma unroll
    for (int k = 0; k < KS; ++k) fa[i][k] = *(const u32x4f*)(Ab + (size_t)ar * A + k * 32);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int br = u0 + 16 * j + l15;
#pragma unroll
    for (int k = 0; k < KS; ++k) fb[j][k] = *(const u32x4f*)(Bb + (size_t)br * A + k * 32);
  }
  // ---- cell operands of this thread's (row, 4 units): behind the fragments in the request order (loads return in order, and
  // the MFMAs come first), still long before they are used
  uint2 g[4];
  {
    const bf16_t* G = (const bf16_t*)p.gates + (size_t)rr * 4 * H + cu;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = *(const uint2*)(G + q * H);
  }
  // Every request of the launch in ONE burst, none behind a run-time condition (round 6): a slab / a previous cell state that does
  // not exist is read from a valid stand-in address and dropped by a select where it would be used.  hipcc waits for a predicated
  // load at the join of its branch -- s_waitcnt vmcnt(0) right behind the first optional slab -- and left alone it also feeds the
  // fragment loads to the MFMAs a few at a time: the launch was five to six serial memory round trips (10.8 us at 0.7 % MFMA-busy).
  unsigned zero = 0;                                  // (an offset through an empty asm: hipcc must not see that the stand-in
  asm volatile("" : "+s"(zero));                      //  equals p.dc and fold the loads -- back to a load under a branch)
  const float* standin = p.dc + zero;
  const float4 c4 = *(const float4*)(p.c + idx);
  const float4 cp4r = *(const float4*)((p.c_prev ? p.c_prev : standin) + idx);
  const float4 dc4 = *(const float4*)(p.dc + idx);
  float4 sv[8];
  {
    const float* bA = p.nA ? p.slabA : standin;
    const float* bB = p.nB ? p.slabB : standin;
    const size_t sA = p.nA ? p.strideA : 0, sB = p.nB ? p.strideB : 0;
    const size_t oA = p.nA ? (size_t)rr * p.ldA + cu : idx, oB = p.nB ? (size_t)rr * p.ldB + cu : idx;
    const int mA = p.nA > 0 ? p.nA - 1 : 0, mB = p.nB > 0 ? p.nB - 1 : 0;
#pragma unroll
    for (int z = 0; z < 4; ++z) sv[z] = *(const float4*)(bA + (size_t)min(z, mA) * sA + oA);
#pragma unroll
    for (int z = 0; z < 4; ++z) sv[4 + z] = *(const float4*)(bB + (size_t)min(z, mB) * sB + oB);
  }
  __builtin_amdgcn_sched_barrier(0);                  // (everything above is in flight before the first MFMA)
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mma16(fa[i][k], fb[j][k], acc[i][j]);
  // D layout: lane holds rows 4 lq + r, column l15 of the 16 x 16 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][(16 * i + 4 * lq + r) * FT + 16 * j + l15] = acc[i][j][r];
  __syncthreads();
  // ---- the cell backward (same formulas as lstm_bwd_kernel, pointwise.hip), d h = GEMM + slab sums in a fixed order
  float dh[4];
  {
    const int o = (tid >> 3) * FT + 4 * (tid & 7);
    float4 s = *(const float4*)(&red[0][o]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 v = *(const float4*)(&red[w][o]);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4 ds = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int z = 0; z < 8; ++z) {
      const bool use = z < 4 ? z < p.nA : z - 4 < p.nB;
      ds.x += use ? sv[z].x : 0.f; ds.y += use ? sv[z].y : 0.f; ds.z += use ? sv[z].z : 0.f; ds.w += use ? sv[z].w : 0.f;
    }
    dh[0] = s.x + ds.x; dh[1] = s.y + ds.y; dh[2] = s.z + ds.z; dh[3] = s.w + ds.w;
  }
  if (!live) return;
  float gq[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    gq[q][0] = __uint_as_float(g[q].x << 16); gq[q][1] = __uint_as_float(g[q].x & 0xffff0000u);
    gq[q][2] = __uint_as_float(g[q].y << 16); gq[q][3] = __uint_as_float(g[q].y & 0xffff0000u);
  }
  const float4 cp4 = p.c_prev ? cp4r : make_float4(0.f, 0.f, 0.f, 0.f);
  const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, cpv[4] = {cp4.x, cp4.y, cp4.z, cp4.w}, dcv[4] = {dc4.x, dc4.y, dc4.z, dc4.w};
  float o4[4][4], dcn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gi = gq[0][k], gf = gq[1][k], gg = gq[2][k], go = gq[3][k];
    const float tc = uic_tanh<bf16_t>(cc[k]);
    const float d = dcv[k] + dh[k] * go * (1.f - tc * tc);
    const float d_o = dh[k] * tc;
    o4[0][k] = d * gg * gi * (1.f - gi);
    o4[1][k] = d * cpv[k] * gf * (1.f - gf);
    o4[2][k] = d * gi * (1.f - gg * gg);
    o4[3][k] = d_o * go * (1.f - go);
    dcn[k] = d * gf;
  }
  bf16_t* D = (bf16_t*)p.dgates + (size_t)crow * 4 * H + cu;
#pragma unroll
  for (int q = 0; q < 4; ++q) *(uint2*)(D + q * H) = make_uint2(uic_pack_bf16x2(o4[q][0], o4[q][1]), uic_pack_bf16x2(o4[q][2], o4[q][3]));
  *(float4*)(p.dc + idx) = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
}


}  // namespace

bool uic_h2att_cell_bwd_eligible(const UicH2attCellParams& p) {
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const int ks = p.A / 128;
  return p.dtype == UIC_BF16 && p.N >= 1 && p.H % FT == 0 && p.A % 128 == 0 && (ks == 1 || ks == 2 || ks == 4) && p.nA <= 4 && p.nB <= 4 &&
         al16(p.datth) && al16(p.h2attT) && ((uintptr_t)p.gates & 7) == 0 && ((uintptr_t)p.dgates & 7) == 0 && al16(p.c) &&
         (!p.c_prev || al16(p.c_prev)) && al16(p.dc) && (!p.nA || (al16(p.slabA) && p.ldA % 4 == 0 && p.strideA % 4 == 0)) &&
         (!p.nB || (al16(p.slabB) && p.ldB % 4 == 0 && p.strideB % 4 == 0));
}

int uic_h2att_cell_bwd_launch(const UicH2attCellParams& p, hipStream_t s) {
  UIC_REQUIRE(p.datth && p.h2attT && p.gates && p.c && p.dc && p.dgates, "h2att_cell_bwd: null pointer");
  UIC_REQUIRE(uic_h2att_cell_bwd_eligible(p), "h2att_cell_bwd: shape not eligible (bf16, H %% 32 == 0, A in {128, 256, 512})");
  const dim3 grid((unsigned)((p.N + FT - 1) / FT), (unsigned)(p.H / FT));
  switch (p.A / 128) {
    case 1: hipLaunchKernelGGL(h2att_cell_bwd_kernel<1>, grid, dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL(h2att_cell_bwd_kernel<2>, grid, dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL(h2att_cell_bwd_kernel<4>, grid, dim3(256), 0, s, p); break;
  }
  UIC_LAUNCH_CHECK("h2att_cell_bwd_kernel");
  return UIC_OK;
}
