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
  // ---- cell operands of this thread's (row, 4 units), requested first
  const int crow = row0 + (tid >> 3), cu = u0 + 4 * (tid & 7);
  const bool live = crow < p.N;
  const int rr = live ? crow : p.N - 1;
  const size_t idx = (size_t)rr * H + cu;
  uint2 g[4];
  {
    const bf16_t* G = (const bf16_t*)p.gates + (size_t)rr * 4 * H + cu;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = *(const uint2*)(G + q * H);
  }
  const float4 c4 = *(const float4*)(p.c + idx);
  const float4 cp4 = p.c_prev ? *(const float4*)(p.c_prev + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 dc4 = *(const float4*)(p.dc + idx);
  float4 sv[8];
#pragma unroll
  for (int z = 0; z < 4; ++z)
    sv[z] = z < p.nA ? *(const float4*)(p.slabA + (size_t)z * p.strideA + (size_t)rr * p.ldA + cu) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int z = 0; z < 4; ++z)
    sv[4 + z] = z < p.nB ? *(const float4*)(p.slabB + (size_t)z * p.strideB + (size_t)rr * p.ldB + cu) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    for (int z = 0; z < 8; ++z) { ds.x += sv[z].x; ds.y += sv[z].y; ds.z += sv[z].z; ds.w += sv[z].w; }
    dh[0] = s.x + ds.x; dh[1] = s.y + ds.y; dh[2] = s.z + ds.z; dh[3] = s.w + ds.w;
  }
  if (!live) return;
  float gq[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    gq[q][0] = __uint_as_float(g[q].x << 16); gq[q][1] = __uint_as_float(g[q].x & 0xffff0000u);
    gq[q][2] = __uint_as_float(g[q].y << 16); gq[q][3] = __uint_as_float(g[q].y & 0xffff0000u);
  }
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


// ---------------------------------------------------------------------------------------------------
// The other two links: d[h_lang_prev | h_att_prev] = dG1_t W1rec (640 x 1024 x 2048, the `d x1` GEMM of step t) and the
// lang_lstm cell backward of step t - 1, which consumes its h_lang half (+ the d x2 slabs of step t + the logit layer's
// d hdrop of step t - 1).  K = 4H does not fit the all-operands-up-front form above, so the fragments stream DX_D k-steps
// ahead of their MFMAs; the tile is 32 rows x 64 columns per workgroup of 4 waves (K split over the waves, 16 k-steps each):
// 320 workgroups of ~250 registers x 4 waves, two of which fit a CU beside the side stream's GEMM workgroups.  Column tiles
// of the h_att half just store their sums (one f32 buffer, read by h2att_cell_bwd_kernel of step t - 1).
constexpr int DX_TM = 32, DX_TN = 64, DX_D = 5, DX_KS = 16;     // k-steps per wave: 4H / (4 waves x 32) = 16 for H = 512

__global__ __launch_bounds__(256) void dx1_cell_bwd_kernel(const UicDx1CellParams p) {
  __shared__ __attribute__((aligned(16))) float red[4][DX_TM * DX_TN];      // 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int row0 = blockIdx.x * DX_TM, col0 = blockIdx.y * DX_TN;
  const int H = p.H, K = 4 * H;
  const bool cell = col0 < H;                      // (uniform) this tile's columns are hidden units of the lang cell
  // ---- epilogue coordinates: thread = (row, 8 consecutive columns)
  const int erow = row0 + (tid >> 3), ec = col0 + 8 * (tid & 7);
  const bool live = erow < p.N;
  const int rr = live ? erow : p.N - 1;
  // cell operands that cost few registers are requested first (gates, states); the d h sources follow inside the K loop
  u32x4f g[4];
  float4 c4[2], cp4[2], dc4[2];
  const size_t idx = (size_t)rr * H + ec;
  if (cell) {
    const bf16_t* G = (const bf16_t*)p.gates + (size_t)rr * 4 * H + ec;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = *(const u32x4f*)(G + q * H);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      c4[h] = *(const float4*)(p.c + idx + 4 * h);
      cp4[h] = p.c_prev ? *(const float4*)(p.c_prev + idx + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      dc4[h] = *(const float4*)(p.dc + idx + 4 * h);
    }
  }
  // ---- GEMM: wave w multiplies K slice w
  const bf16_t* Ab = (const bf16_t*)p.dg1 + (size_t)wave * DX_KS * 32 + lq * 8;
  const bf16_t* Bb = (const bf16_t*)p.w1recT + (size_t)wave * DX_KS * 32 + lq * 8;
  const bf16_t* ap[2];
  const bf16_t* bp[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) ap[i] = Ab + (size_t)min(row0 + 16 * i + l15, p.N - 1) * K;
#pragma unroll
  for (int j = 0; j < 4; ++j) bp[j] = Bb + (size_t)(col0 + 16 * j + l15) * K;
  u32x4f fa[DX_D][2], fb[DX_D][4];
  auto load = [&](int buf, int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[buf][i] = *(const u32x4f*)(ap[i] + ks * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[buf][j] = *(const u32x4f*)(bp[j] + ks * 32);
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < DX_D - 1; ++q) load(q, q);
  float4 d0[2], sv[4][2];                          // d hdrop of step t - 1 and the d x2 slabs of step t (h_lang columns)
#pragma unroll
  for (int ks = 0; ks < DX_KS; ++ks) {
    if (ks + DX_D - 1 < DX_KS) load((ks + DX_D - 1) % DX_D, ks + DX_D - 1);
    if (ks == DX_KS - DX_D + 1 && cell) {          // the last fragments are on their way: now the remaining cell operands
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        d0[h] = *(const float4*)(p.dh0 + (size_t)rr * p.lddh0 + ec + 4 * h);
#pragma unroll
        for (int z = 0; z < 4; ++z)
          sv[z][h] = z < p.nA ? *(const float4*)(p.slabA + (size_t)z * p.strideA + (size_t)rr * p.ldA + ec + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mma16(fa[ks % DX_D][i], fb[ks % DX_D][j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // D layout: lane holds rows 4 lq + r, column l15 of the 16 x 16 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][(16 * i + 4 * lq + r) * DX_TN + 16 * j + l15] = acc[i][j][r];
  __syncthreads();
  float v[8];
  {
    const int o = (tid >> 3) * DX_TN + 8 * (tid & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 s = *(const float4*)(&red[0][o + 4 * h]);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 x = *(const float4*)(&red[w][o + 4 * h]);
        s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
      }
      v[4 * h] = s.x; v[4 * h + 1] = s.y; v[4 * h + 2] = s.z; v[4 * h + 3] = s.w;
    }
  }
  if (!live) return;
  if (!cell) {                                     // the h_att half: for h2att_cell_bwd_kernel of step t - 1
    float* o = p.dx1_hatt + (size_t)erow * p.ld_hatt + (ec - H);
    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    return;
  }
  // ---- lang cell backward of step t - 1 (lstm_bwd_vec4_kernel's formulas; d h = dropout-scaled d hdrop + slabs + this GEMM)
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  float gq[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned w4[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { gq[q][2 * k] = __uint_as_float(w4[k] << 16); gq[q][2 * k + 1] = __uint_as_float(w4[k] & 0xffff0000u); }
  }
  float o8[4][8], dcn[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int h = k >> 2, e = k & 3;
    const float dh0 = e == 0 ? d0[h].x : e == 1 ? d0[h].y : e == 2 ? d0[h].z : d0[h].w;
    float ds = 0.f;
#pragma unroll
    for (int z = 0; z < 4; ++z) ds += e == 0 ? sv[z][h].x : e == 1 ? sv[z][h].y : e == 2 ? sv[z][h].z : sv[z][h].w;
    float dhd = dh0;
    if (p.drop_p > 0.f) dhd *= uic_drop_scale(p.seed, p.site, (unsigned)(idx + k), p.drop_p, inv_keep);
    const float dh = dhd + ds + v[k];
    const float cc = e == 0 ? c4[h].x : e == 1 ? c4[h].y : e == 2 ? c4[h].z : c4[h].w;
    const float cpv = e == 0 ? cp4[h].x : e == 1 ? cp4[h].y : e == 2 ? cp4[h].z : cp4[h].w;
    const float dcv = e == 0 ? dc4[h].x : e == 1 ? dc4[h].y : e == 2 ? dc4[h].z : dc4[h].w;
    const float gi = gq[0][k], gf = gq[1][k], gg = gq[2][k], go = gq[3][k];
    const float tc = uic_tanh<bf16_t>(cc);
    const float d = dcv + dh * go * (1.f - tc * tc);
    const float d_o = dh * tc;
    o8[0][k] = d * gg * gi * (1.f - gi);
    o8[1][k] = d * cpv * gf * (1.f - gf);
    o8[2][k] = d * gi * (1.f - gg * gg);
    o8[3][k] = d_o * go * (1.f - go);
    dcn[k] = d * gf;
  }
  bf16_t* D = (bf16_t*)p.dgates + (size_t)erow * 4 * H + ec;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *(u32x4f*)(D + q * H) = u32x4f{uic_pack_bf16x2(o8[q][0], o8[q][1]), uic_pack_bf16x2(o8[q][2], o8[q][3]),
                                   uic_pack_bf16x2(o8[q][4], o8[q][5]), uic_pack_bf16x2(o8[q][6], o8[q][7])};
  *(float4*)(p.dc + idx) = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
  *(float4*)(p.dc + idx + 4) = make_float4(dcn[4], dcn[5], dcn[6], dcn[7]);
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

bool uic_dx1_cell_bwd_eligible(const UicDx1CellParams& p) {
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  return p.dtype == UIC_BF16 && p.N >= 1 && p.H == 32 * DX_KS && p.nA <= 4 && al16(p.dg1) && al16(p.w1recT) && al16(p.gates) && al16(p.dgates) &&
         al16(p.c) && (!p.c_prev || al16(p.c_prev)) && al16(p.dc) && p.dh0 && al16(p.dh0) && p.lddh0 % 4 == 0 && al16(p.dx1_hatt) && p.ld_hatt % 4 == 0 &&
         (!p.nA || (al16(p.slabA) && p.ldA % 4 == 0 && p.strideA % 4 == 0));
}

int uic_dx1_cell_bwd_launch(const UicDx1CellParams& p, hipStream_t s) {
  UIC_REQUIRE(p.dg1 && p.w1recT && p.gates && p.c && p.dc && p.dgates && p.dx1_hatt, "dx1_cell_bwd: null pointer");
  UIC_REQUIRE(uic_dx1_cell_bwd_eligible(p), "dx1_cell_bwd: shape not eligible (bf16, H = %d)", 32 * DX_KS);
  const dim3 grid((unsigned)((p.N + DX_TM - 1) / DX_TM), (unsigned)(2 * p.H / DX_TN));
  hipLaunchKernelGGL(dx1_cell_bwd_kernel, grid, dim3(256), 0, s, p);
  UIC_LAUNCH_CHECK("dx1_cell_bwd_kernel");
  return UIC_OK;
}
