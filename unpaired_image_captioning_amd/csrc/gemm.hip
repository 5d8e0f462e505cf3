// MFMA GEMM for gfx950:  C[M,N] = sum_seg A_s[M,K_s] * B_s[N,K_s]^T  (+ epilogue)
//
// Replaces the nn.Linear / nn.LSTMCell GEMM call sites of the reference hot path
// (P/models/AttModel.py:76-92 fc_embed/att_embed/logit/ctx2att, :426-427,434,441 the two
// LSTMCells, :543 h2att).  Both operands are K-contiguous ("NT"), which is how nn.Linear
// stores its weight; transposed weight/activation copies make every backward GEMM NT too.
//
//  * bf16 operands -> v_mfma_f32_32x32x16_bf16, f32 operands -> v_mfma_f32_32x32x2_f32
//    (exact-f32 MFMA, the parity path); accumulation is always f32.
//  * A block stages a [BM x 128 B] A tile and a [BN x 128 B] B tile per K step through LDS
//    (row stride 144 B => the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots).
//    Each lane half owns 64 contiguous bytes of the 128-B K slice, so fragments are plain
//    16-B LDS reads; the K order inside a tile is permuted identically for A and B.
//  * K segments: the concatenations torch.cat([prev_h, fc, xt]) / cat([att, h_att]) of
//    TopDownCore.forward (AttModel.py:432,438) are never materialised.
//  * LSTM mode: a wave owns the 4 gates (i,f,g,o) of 32 hidden units for 32 rows in four
//    32x32 accumulators with identical lane layout, so the cell update
//    c' = s(f) c + s(i) tanh(g), h' = s(o) tanh(c') runs in the epilogue on registers.
#include "uic_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int LDS_STRIDE = 144;   // 128-B K slice + 16-B pad
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// KS > 1: KS groups of WM x WN waves each take one 128-byte K slice of a (KS * 128)-byte K step for the
// SAME output tile and are summed through LDS at the end.  The skinny per-decode-step GEMMs (M = 640)
// are latency chains of K/BK dependent load->LDS->MFMA rounds; KS = 4 makes the chain 4x shorter and
// puts 4x the bytes in flight per round.
// PF: K rounds kept in flight in registers (global -> VGPR) ahead of the round being multiplied.  With PF rounds
// outstanding the chain is one load latency plus the LDS/MFMA rounds instead of one load latency PER round.
template <typename T, int TM, int TN, int WM, int WN, int KS, bool LSTM, int PF>
__global__ __launch_bounds__(64 * WM * WN * KS) void uic_gemm_kernel(const UicGemmParams p) {
  constexpr int NT = 64 * WM * WN * KS;
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BK = KS * 128 / (int)sizeof(T);   // K elements per round
  constexpr int CPR = 8 * KS;                      // 16-byte chunks per tile row per round
  constexpr int A_CH = BM * CPR / NT;
  constexpr int B_CH = BN * CPR / NT;
  static_assert(BM * CPR % NT == 0 && BN * CPR % NT == 0, "tile/threads mismatch");
  static_assert(!LSTM || TN == 4 || TN == 5, "LSTM mode keeps the 4 (nn.LSTMCell) or 5 (maxout LSTMCore) gate chunks in the N tiles of a wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + KS * BM * LDS_STRIDE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ks_id = wave / (WM * WN);
  const int wmn = wave % (WM * WN);
  const int wm = wmn / WN;
  const int wn = wmn % WN;
  const int half = lane >> 5;
  const int r32 = lane & 31;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MB L2), so
  // give every XCD a contiguous run of tiles, M fastest: the tiles of one XCD share a few B (weight) tiles and
  // walk A once, which keeps the re-reads of both operands inside that XCD's L2.  (bijective for any grid)
  int bm, bn;
  {
    const int gx = gridDim.x, nblk = gx * gridDim.y;
    const int lin = blockIdx.x + gx * blockIdx.y;
    const int q = nblk >> 3, r = nblk & 7, xcd = lin & 7, idx = lin >> 3;
    const int lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bm = lp % gx;
    bn = lp / gx;
  }
  const int m0 = bm * BM;
  const int n0 = bn * (LSTM ? 32 * WN : BN);   // LSTM: first hidden unit of the block

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  int ntiles = 0;
  for (int s = 0; s < p.nseg; ++s) ntiles += (p.seg[s].K + BK - 1) / BK;

  uint4 rra[PF][A_CH], rrb[PF][B_CH];
  int seg = 0, k0 = 0;        // K position of the NEXT round to load

  auto load_tile = [&](uint4* ra, uint4* rb) {
    const UicGemmSeg sg = p.seg[seg];
    const char* Ab = (const char*)sg.A;
    const char* Bb = (const char*)sg.B;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int c = tid + i * NT;
      const int row = c / CPR;
      const int k = k0 + (c % CPR) * VEC;
      const int gm = m0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gm < p.M && k < sg.K) v = *(const uint4*)(Ab + ((size_t)gm * sg.lda + k) * sizeof(T));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int c = tid + i * NT;
      const int row = c / CPR;
      const int k = k0 + (c % CPR) * VEC;
      int grow;
      bool ok;
      if (LSTM) {
        const int u = n0 + (row / (TN * 32)) * 32 + (row & 31);
        grow = ((row % (TN * 32)) >> 5) * p.H + u;
        ok = u < p.H;
      } else {
        grow = n0 + row;
        ok = grow < p.N;
      }
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok && k < sg.K) v = *(const uint4*)(Bb + ((size_t)grow * sg.ldb + k) * sizeof(T));
      rb[i] = v;
    }
    k0 += BK;
    if (k0 >= sg.K) { ++seg; k0 = 0; }
  };
  auto store_tile = [&](const uint4* ra, const uint4* rb) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int c = tid + i * NT;
      const int kc = c % CPR;
      *(uint4*)(sA + ((kc >> 3) * BM + c / CPR) * LDS_STRIDE + (kc & 7) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int c = tid + i * NT;
      const int kc = c % CPR;
      *(uint4*)(sB + ((kc >> 3) * BN + c / CPR) * LDS_STRIDE + (kc & 7) * 16) = rb[i];
    }
  };

  // LSTM epilogue operands of the (row, unit) pairs this lane will finish -- the gate pre-activations made elsewhere, the
  // previous cell state, the biases: requested NOW, so that their latency passes under the K rounds instead of after them
  // (these launches are a few microseconds of dependent latencies each; the NMT step is 256 of them)
  constexpr int RPG0 = 16 / KS;
  float pf_pre[LSTM ? TM : 1][LSTM ? RPG0 : 1][LSTM ? TN : 1], pf_c[LSTM ? TM : 1][LSTM ? RPG0 : 1], pf_b[LSTM ? TN : 1];
  float pf_pre2[LSTM ? TM : 1][LSTM ? RPG0 : 1][LSTM ? TN : 1], pf_b2[LSTM ? TN : 1];
  if constexpr (LSTM) {
    // Round 6: none of these requests sits behind a condition any more.  An optional operand that is absent is read from a valid
    // stand-in word (c_out[0]) and dropped by a select; lanes / rows beyond the problem read a clamped index.  hipcc waits for a
    // load under a branch at the join of that branch: the two biases were eight serial memory round trips at the top of every
    // launch, and `pre2`, read in the epilogue under `if (p.pre2)`, four more per output row -- in launches that are a few
    // microseconds of dependent latencies in the first place.
    const int u = n0 + wn * 32 + r32;
    const int uc = u < p.H ? u : p.H - 1;
    const float* b1 = p.bias ? p.bias : p.c_out;
    const float* b2 = p.bias2 ? p.bias2 : p.c_out;
    float l1[TN], l2[TN];
#pragma unroll
    for (int g = 0; g < TN; ++g) {
      l1[g] = b1[p.bias ? g * p.H + uc : 0];
      l2[g] = b2[p.bias2 ? g * p.H + uc : 0];
    }
    const float* cpp = p.c_prev ? p.c_prev : p.c_out;
    const float* p1 = p.pre1 ? p.pre1 : p.c_out;
    const float* p2 = p.pre2 ? p.pre2 : p.c_out;
    float lc[TM][RPG0], lp1[TM][RPG0][TN], lp2[TM][RPG0][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < RPG0; ++q) {
        const int reg = ks_id * RPG0 + q;
        const int row = m0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        const int rc = row < p.M ? row : p.M - 1;
        lc[i][q] = cpp[p.c_prev ? (size_t)rc * p.H + uc : 0];
#pragma unroll
        for (int g = 0; g < TN; ++g) {
          lp1[i][q][g] = p1[p.pre1 ? (size_t)rc * p.ldpre1 + g * p.H + uc : 0];
          lp2[i][q][g] = p2[p.pre2 ? (size_t)rc * p.ldpre2 + g * p.H + uc : 0];
        }
      }
    // (the raw values stay in registers through the K rounds; the selects that drop the stand-ins are applied in the epilogue --
    // here they would be the first use of the loads and bring their wait to the top of the launch)
#pragma unroll
    for (int g = 0; g < TN; ++g) { pf_b[g] = l1[g]; pf_b2[g] = l2[g]; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < RPG0; ++q) {
        pf_c[i][q] = lc[i][q];
#pragma unroll
        for (int g = 0; g < TN; ++g) { pf_pre[i][q][g] = lp1[i][q][g]; pf_pre2[i][q][g] = lp2[i][q][g]; }
      }
  }
  // likewise the values an accumulating skinny GEMM adds to (f32 C += A B^T: the per-step `h2att` term of the BPTT loop)
  constexpr bool PFA = !LSTM && KS > 1;
  float pf_acc[PFA ? TM : 1][PFA ? TN : 1][PFA ? RPG0 : 1];
  const bool pf_accum = PFA && (p.flags & UIC_GEMM_ACCUM) && ((p.flags & UIC_GEMM_OUT_F32) || sizeof(T) == 4);
  if constexpr (PFA) {
    if (pf_accum) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + (wn * TN + j) * 32 + r32;
#pragma unroll
          for (int q = 0; q < RPG0; ++q) {
            const int reg = ks_id * RPG0 + q;
            const int row = m0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            pf_acc[i][j][q] = (row < p.M && col < p.N) ? ((const float*)p.C)[(size_t)row * p.ldc + col] : 0.f;
          }
        }
    }
  }
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < ntiles) load_tile(rra[j], rrb[j]);
  for (int it0 = 0; it0 < ntiles; it0 += PF) {
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int it = it0 + j;
    if (it >= ntiles) break;
    store_tile(rra[j], rrb[j]);
    __syncthreads();
    if (it + PF < ntiles) load_tile(rra[j], rrb[j]);
    const char* pa = sA + (ks_id * BM + wm * 32 * TM + r32) * LDS_STRIDE + half * 64;
    const char* pb = sB + (ks_id * BN + wn * 32 * TN + r32) * LDS_STRIDE + half * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(pa + i * 32 * LDS_STRIDE + ks * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(pb + j * 32 * LDS_STRIDE + ks * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].x), __uint_as_float(fb[j].x), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].y), __uint_as_float(fb[j].y), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].z), __uint_as_float(fb[j].z), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].w), __uint_as_float(fb[j].w), acc[i][j], 0, 0, 0);
          }
        }
    }
    __syncthreads();
  }
  }

  // ---- reduce-scatter over the KS wave groups: group q ends up with the complete sums of accumulator
  // registers [q*RPG, (q+1)*RPG) of every tile (= 8-row bands of the output), so ALL waves share the epilogue.
  constexpr int RPG = 16 / KS;
  if constexpr (KS > 1) {
    float* red = (float*)smem;    // [dst group][src slot][wmn][tile][RPG][64]; staging buffers are free now
    constexpr int TILE = TM * TN;
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      if (q != ks_id) {
        const int slot = ks_id < q ? ks_id : ks_id - 1;
        float* dst = red + (size_t)(((q * (KS - 1) + slot) * WM * WN + wmn) * TILE) * RPG * 64 + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int rr = 0; rr < RPG; ++rr) dst[((i * TN + j) * RPG + rr) * 64] = acc[i][j][q * RPG + rr];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      if (q == ks_id) {
#pragma unroll
        for (int slot = 0; slot < KS - 1; ++slot) {
          const float* src = red + (size_t)(((q * (KS - 1) + slot) * WM * WN + wmn) * TILE) * RPG * 64 + lane;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int rr = 0; rr < RPG; ++rr) acc[i][j][q * RPG + rr] += src[((i * TN + j) * RPG + rr) * 64];
        }
      }
    }
  }

  // ------------------------------------------------------------------ epilogue
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  auto epilogue = [&](auto r0c) {
    constexpr int R0 = decltype(r0c)::value;
    if constexpr (!LSTM) {
      const bool out_f32 = (p.flags & UIC_GEMM_OUT_F32) || sizeof(T) == 4;
      int arow[TM][RPG];                 // addend row of every output row this lane holds (row % add_mod, once per row)
      if (p.addend) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < RPG; ++q) {
            const int reg = R0 + q;
            arow[i][q] = (m0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) % p.add_mod;
          }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + (wn * TN + j) * 32 + r32;
          if (col >= p.N) continue;
          float b = 0.f;
          if (p.bias) b += p.bias[col];
          if (p.bias2) b += p.bias2[col];
#pragma unroll
          for (int reg = R0; reg < R0 + RPG; ++reg) {
            const int row = m0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            if (row >= p.M) continue;
            float v = acc[i][j][reg] + b;
            if (p.addend) v += p.addend[(size_t)arow[i][reg - R0] * p.ld_add + col];
            if (p.flags & UIC_GEMM_RELU) v = fmaxf(v, 0.f);
            if (p.flags & UIC_GEMM_TANH) v = uic_tanh<T>(v);
            if (p.row_len) {
              const int n = row / p.R;
              if (row - n * p.R >= p.row_len[n]) v = 0.f;
            }
            if (p.C_pre) ((T*)p.C_pre)[(size_t)row * p.ldc_pre + col] = uic_from_f<T>(v);
            if (p.drop_p > 0.f) v *= uic_drop_scale(p.seed, p.site, (unsigned)(row + p.drop_row0) * (unsigned)p.N + (unsigned)col, p.drop_p, inv_keep);
            const size_t o = (size_t)row * p.ldc + col;
            if (out_f32) {
              float* C = (float*)p.C;
              if constexpr (PFA) {
                if (p.flags & UIC_GEMM_ACCUM) v += pf_accum ? pf_acc[i][j][reg - R0] : C[o];   // (requested before the K rounds)
              } else {
                if (p.flags & UIC_GEMM_ACCUM) v += C[o];
              }
              C[o] = v;
            } else {
              T* C = (T*)p.C;
              if (p.flags & UIC_GEMM_ACCUM) v += uic_to_f(C[o]);
              C[o] = uic_from_f<T>(v);
            }
          }
        }
    } else {
      const int H = p.H;
      const int u = n0 + wn * 32 + r32;
      if (u < H) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int reg = R0; reg < R0 + RPG; ++reg) {
            const int row = m0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            if (row >= p.M) continue;
            float g4[TN];
#pragma unroll
            for (int g = 0; g < TN; ++g) {
              float v = acc[i][g][reg] + ((p.bias ? pf_b[g] : 0.f) + (p.bias2 ? pf_b2[g] : 0.f));
              if (p.pre1) v += pf_pre[i][reg - R0][g];             // (requested before the K rounds)
              if (p.pre2) v += pf_pre2[i][reg - R0][g];
              g4[g] = v;
            }
            const float cp = p.c_prev ? pf_c[i][reg - R0] : 0.f;
            if constexpr (TN == 4) {
              // nn.LSTMCell: chunks (i, f, g, o)
              const float gi = uic_sigmoid_t<T>(g4[0]);
              const float gf = uic_sigmoid_t<T>(g4[1]);
              const float gg = uic_tanh<T>(g4[2]);
              const float go = uic_sigmoid_t<T>(g4[3]);
              const float c = gf * cp + gi * gg;
              const float h = go * uic_tanh<T>(c);
              p.c_out[(size_t)row * H + u] = c;
              ((T*)p.h_out)[(size_t)row * p.ldh + u] = uic_from_f<T>(h);
              if (p.h_drop) {
                float hd = h;
                if (p.drop_p > 0.f) hd *= uic_drop_scale(p.seed, p.site, (unsigned)row * (unsigned)H + (unsigned)u, p.drop_p, inv_keep);
                ((T*)p.h_drop)[(size_t)row * p.ldhd + u] = uic_from_f<T>(hd);
              }
              if (p.gates_out) {
                T* G = (T*)p.gates_out + (size_t)row * 4 * H + u;      // read again only in the backward pass
                __builtin_nontemporal_store(uic_from_f<T>(gi), G);
                __builtin_nontemporal_store(uic_from_f<T>(gf), G + H);
                __builtin_nontemporal_store(uic_from_f<T>(gg), G + 2 * H);
                __builtin_nontemporal_store(uic_from_f<T>(go), G + 3 * H);
              }
            } else {
              // maxout LSTMCore (P/models/FCModel_NMT.py:32-50): chunks (in, forget, out, a, b); g = max(a, b);
              // the DROPPED next_h is both the output and the recurrent state (:47-50)
              const float gi = uic_sigmoid_t<T>(g4[0]);
              const float gf = uic_sigmoid_t<T>(g4[1]);
              const float go = uic_sigmoid_t<T>(g4[2]);
              const bool first = g4[3] >= g4[4];
              const float gg = first ? g4[3] : g4[4];
              const float c = gf * cp + gi * gg;
              float h = go * uic_tanh<T>(c);
              if (p.drop_p > 0.f) h *= uic_drop_scale(p.seed, p.site, (unsigned)row * (unsigned)H + (unsigned)u, p.drop_p, inv_keep);
              p.c_out[(size_t)row * H + u] = c;
              ((T*)p.h_out)[(size_t)row * p.ldh + u] = uic_from_f<T>(h);
              if (p.gates_out) {
                T* G = (T*)p.gates_out + (size_t)row * 5 * H + u;
                G[0] = uic_from_f<T>(gi);
                G[H] = uic_from_f<T>(gf);
                G[2 * H] = uic_from_f<T>(go);
                G[3 * H] = uic_from_f<T>(gg);
                G[4 * H] = uic_from_f<T>(first ? 1.f : 0.f);
              }
            }
          }
      }
    }
  };
  if constexpr (KS == 1) {
    epilogue(std::integral_constant<int, 0>{});
  } else {
    static_assert(KS == 4 || KS == 8, "epilogue dispatch is written for KS = 4 or 8");
    if (ks_id == 0) epilogue(std::integral_constant<int, 0>{});
    else if (ks_id == 1) epilogue(std::integral_constant<int, RPG>{});
    else if (ks_id == 2) epilogue(std::integral_constant<int, 2 * RPG>{});
    else if (ks_id == 3) epilogue(std::integral_constant<int, 3 * RPG>{});
    else if constexpr (KS == 8) {
      if (ks_id == 4) epilogue(std::integral_constant<int, 4 * RPG>{});
      else if (ks_id == 5) epilogue(std::integral_constant<int, 5 * RPG>{});
      else if (ks_id == 6) epilogue(std::integral_constant<int, 6 * RPG>{});
      else epilogue(std::integral_constant<int, 7 * RPG>{});
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Large-GEMM path: 128x128 tile, 4 waves (2x2, 64x64 each), K rounds of 128 bytes staged global -> LDS
// directly (global_load_lds_dwordx4, no VGPR round trip) into two LDS buffers, one barrier per round.
// LDS rows are the bare 128-byte K slices (an LDS-DMA wave instruction writes 1 KiB = 8 rows linearly),
// so the bank-conflict fix is an XOR swizzle applied to the per-lane SOURCE address and again on the
// fragment read: 16-byte chunk c of tile row r lives at chunk c ^ ((r >> 1) & 7), which makes the 16-lane
// groups of ds_read_b128 hit 16 distinct 16-byte slots of the 256-byte bank row.
// Requires one K segment with K a multiple of the round (64 bf16 / 32 f32 elements).
// STAGES (round 6): 2 = two LDS buffers, the next round's DMA issued at the top of a round and waited for at the top of the next
// -- one L2 / Infinity Cache latency (~0.8 us) per 64-deep K round whatever the MFMA time (0.2 us), hidden only by the CU's second
// workgroup; 3 = a ring of three buffers (96 KB: one workgroup per CU), two rounds in flight.  For launches that put at most one
// workgroup on a CU anyway -- the BPTT loop's split-K d x GEMMs: 160 / 240 workgroups of 8 rounds -- the ring is what hides the latency.
template <typename T, int STAGES = 2>
__global__ __launch_bounds__(256, 2) void uic_gemm_glds_kernel(const UicGemmParams p) {   // (2 waves per SIMD = two workgroups per CU: <= 256 registers)
  constexpr int BM = 128, BN = 128;
  constexpr int BK = 128 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16 KB | B 16 KB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, r32 = lane & 31;

  // XCD-aware + grouped tile order: each XCD takes a contiguous run of tiles; inside it tiles advance over
  // GM = 8 row-tiles before moving to the next column tile, so one XCD's L2 holds an 8-tile A band while
  // B tiles stream through once per band.
  // Split-K launches (round 6): the run is taken over the whole 3-D grid in dispatch order (x fastest, z slowest), K slice
  // slowest in the tile order -- an XCD then works on ONE K slice (two at most) of a few column tiles: at the BPTT loop's
  // 5 x 12 x 4 grid it reads 0.66 MB of A and 0.79 MB of B instead of all 2.6 MB of A and B tiles of every K slice, and its B
  // tiles are the same ones every decode step.  (gz = 1: the order above, unchanged)
  int bm, bn, bz;
  {
    const int gx = gridDim.x, gy = gridDim.y, nxy = gx * gy, nblk = nxy * gridDim.z;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int q = nblk >> 3, r = nblk & 7, xcd = lin & 7, idx = lin >> 3;
    const int lp3 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bz = lp3 / nxy;
    const int lp = lp3 - bz * nxy;
    constexpr int GM = 8;
    const int width = GM * gy;
    const int first = (lp / width) * GM;
    const int gsz = min(gx - first, GM);
    const int rem = lp % width;
    bm = first + rem % gsz;
    bn = rem / gsz;
  }
  const int m0 = bm * BM, n0 = bn * BN;

  const UicGemmSeg sg = p.seg[0];
  // per-lane source pointers of this wave's 4 A and 4 B LDS-DMA instructions per round
  const int slot = lane & 7;
  const char* srcA[4];
  const char* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int chunk = slot ^ ((row >> 1) & 7);
    const int gm = min(m0 + row, p.M - 1);
    const int gn = min(n0 + row, p.N - 1);
    srcA[i] = (const char*)sg.A + (size_t)gm * sg.lda * sizeof(T) + chunk * 16;
    srcB[i] = (const char*)sg.B + (size_t)gn * sg.ldb * sizeof(T) + chunk * 16;
  }
  auto stage = [&](int kt, int buf) {
    char* dA = smem + buf * 32768 + wave * 4096;
    char* dB = dA + 16384;
    const size_t koff = (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + koff),
                                       (__attribute__((address_space(3))) void*)(dA + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + koff),
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

  // K rounds of this workgroup: all of them, or slice blockIdx.z of a split-K launch
  int kt0 = 0, nt = sg.K / BK;
  if (p.splitk > 1) {
    const int tps = (nt + p.splitk - 1) / p.splitk;
    kt0 = bz * tps;
    nt = max(0, min(nt - kt0, tps));
  }
  // LDS byte addresses of this lane's fragment chunks (buffer 0, first 32-row sub-tile), one per K step
  const int sw = (r32 >> 1) & 7;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  unsigned adA[4], adB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned pc = (unsigned)(((half * 4 + ks) ^ sw) * 16);
    adA[ks] = lds0 + (unsigned)((wm * 64 + r32) * 128) + pc;
    adB[ks] = lds0 + 16384u + (unsigned)((wn * 64 + r32) * 128) + pc;
  }
  // The fragment reads are inline asm on purpose: hipcc cannot tell the LDS-DMA writes of the NEXT round's
  // buffer from these reads and would drain vmcnt(0) before the first ds_read of every round, serialising
  // load and MFMA.  Ordering is by hand: vmcnt(0) + barrier at the top of a round covers the RAW on this
  // round's buffer and the WAR on the other one; counted lgkmcnt waits name their destination registers.
#define UIC_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define UIC_MFMA4(A0, A1, B0, B1)                                                                         \
  do {                                                                                                   \
    if constexpr (sizeof(T) == 2) {                                                                      \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A0), __builtin_bit_cast(bf16x8, B0), acc[0][0], 0, 0, 0); \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A0), __builtin_bit_cast(bf16x8, B1), acc[0][1], 0, 0, 0); \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1), __builtin_bit_cast(bf16x8, B0), acc[1][0], 0, 0, 0); \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1), __builtin_bit_cast(bf16x8, B1), acc[1][1], 0, 0, 0); \
    } else {                                                                                             \
      const u32x4 aa[2] = {A0, A1}, bb[2] = {B0, B1};                                                    \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) {      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(aa[i].x), __uint_as_float(bb[j].x), acc[i][j], 0, 0, 0); \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(aa[i].y), __uint_as_float(bb[j].y), acc[i][j], 0, 0, 0); \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(aa[i].z), __uint_as_float(bb[j].z), acc[i][j], 0, 0, 0); \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(aa[i].w), __uint_as_float(bb[j].w), acc[i][j], 0, 0, 0); \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)
#define UIC_ISSUE(A0, A1, B0, B1, KS) \
  UIC_DSR(A0, aA[KS], 0); UIC_DSR(A1, aA[KS], 4096); UIC_DSR(B0, aB[KS], 0); UIC_DSR(B1, aB[KS], 4096)
#define UIC_WAIT(N, A0, A1, B0, B1)                                                           \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1));         \
  __builtin_amdgcn_sched_barrier(0)
  auto compute = [&](unsigned bo) {                    // bo = byte offset of the round's buffer (uniform)
    unsigned aA[4], aB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { aA[ks] = adA[ks] + bo; aB[ks] = adB[ks] + bo; }
    u32x4 a00, a01, b00, b01, a10, a11, b10, b11;     // two register sets: K step ks uses set ks & 1
    UIC_ISSUE(a00, a01, b00, b01, 0);
    UIC_ISSUE(a10, a11, b10, b11, 1);
    UIC_WAIT(4, a00, a01, b00, b01);
    UIC_MFMA4(a00, a01, b00, b01);
    UIC_ISSUE(a00, a01, b00, b01, 2);
    UIC_WAIT(4, a10, a11, b10, b11);
    UIC_MFMA4(a10, a11, b10, b11);
    UIC_ISSUE(a10, a11, b10, b11, 3);
    UIC_WAIT(4, a00, a01, b00, b01);
    UIC_MFMA4(a00, a01, b00, b01);
    UIC_WAIT(0, a10, a11, b10, b11);
    UIC_MFMA4(a10, a11, b10, b11);
  };
#undef UIC_WAIT
#undef UIC_ISSUE
#undef UIC_MFMA4
#undef UIC_DSR
  if constexpr (STAGES == 2) {
    if (nt > 0) stage(kt0, 0);
    for (int t = 0; t < nt; t += 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 1 < nt) stage(kt0 + t + 1, 1);
      compute(0u);
      if (t + 1 < nt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) stage(kt0 + t + 2, 0);
        compute(32768u);
      }
    }
  } else {
    // ring: rounds t .. t + STAGES - 2 are in flight at the top of round t.  A wave's DMA instructions complete in order, 8 per round:
    // round t's have landed once at most 8 x (newer rounds) are outstanding; the barrier behind the wait makes every wave's share
    // of round t visible and says that buffer (t - 1) % STAGES -- which the next DMA overwrites -- has been read by all.
    for (int i = 0; i < STAGES - 1 && i < nt; ++i) stage(kt0 + i, i);
    int buf = 0;
    for (int t = 0; t < nt; ++t) {
      const int newer = min(STAGES - 2, nt - 1 - t);
      if (newer >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + STAGES - 1 < nt) stage(kt0 + t + STAGES - 1, buf == 0 ? STAGES - 1 : buf - 1);
      compute((unsigned)buf * 32768u);
      buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
  }

  if (p.slab) {   // raw partial tile -> slab[z]
    float* slab = p.slab + (size_t)bz * p.M * p.N;
    if (m0 + BM <= p.M && n0 + BN <= p.N && (size_t)p.M * (size_t)p.N < ((size_t)1 << 31)) {   // whole tile: no tests, 32-bit offsets
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
    return;
  }

  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const bool out_f32 = (p.flags & UIC_GEMM_OUT_F32) || sizeof(T) == 4;
  // Fast epilogue (everything but tanh): ReLU / dropout / output type / whole-tile are decided ONCE (16 specialised bodies), the
  // rarer options (addend, region mask) cost a uniform test per row, offsets are 32-bit, whole tiles skip the bounds tests.  Measured: the general
  // epilogue below, which tests its eight options per element, was ~14 us per wave of tiles -- 43 of the 62 us of a logit chunk.
  if (!(p.flags & UIC_GEMM_TANH) && (size_t)p.M * (size_t)p.ldc < ((size_t)1 << 31)) {
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
    const bool relu = (p.flags & UIC_GEMM_RELU) != 0, drop = p.drop_p > 0.f, accum = (p.flags & UIC_GEMM_ACCUM) != 0;
    float bj[2];
    unsigned cj[2];
    bool cok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + (wn * 2 + j) * 32 + r32;
      cj[j] = (unsigned)col; cok[j] = col < p.N; bj[j] = 0.f;
      if (cok[j]) { if (p.bias) bj[j] += p.bias[col]; if (p.bias2) bj[j] += p.bias2[col]; }
    }
    auto body = [&](auto relu_c, auto drop_c, auto f32_c, auto full_c) {
      constexpr bool RELU = decltype(relu_c)::value, DROP = decltype(drop_c)::value, F32 = decltype(f32_c)::value, FULL = decltype(full_c)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
          if (!FULL && row >= p.M) continue;
          const unsigned ro = (unsigned)row * (unsigned)p.ldc;
          const unsigned dro = (unsigned)(row + p.drop_row0) * (unsigned)p.N;
          // per-row work of the rarer options (uniform tests): the addend's row, the region mask of pack_wrapper
          const float* ad = p.addend ? p.addend + (size_t)(row % p.add_mod) * p.ld_add : nullptr;
          bool live = true;
          if (p.row_len) { const int n = row / p.R; live = row - n * p.R < p.row_len[n]; }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (!FULL && !cok[j]) continue;
            float v = acc[i][j][reg] + bj[j];
            if (ad) v += ad[cj[j]];
            if (RELU) v = fmaxf(v, 0.f);
            if (!live) v = 0.f;
            if (DROP) v *= uic_drop_scale(p.seed, p.site, dro + cj[j], p.drop_p, inv_keep);
            if (F32) {
              float* o = (float*)p.C + (ro + cj[j]);
              if (accum) v += *o;
              *o = v;
            } else {
              T* o = (T*)p.C + (ro + cj[j]);
              if (accum) v += uic_to_f(*o);
              *o = uic_from_f<T>(v);
            }
          }
        }
    };
    using TT = std::true_type; using FF = std::false_type;
#define UIC_EPI(R, D)                                                                              \
    do {                                                                                           \
      if (out_f32) { if (full) body(R{}, D{}, TT{}, TT{}); else body(R{}, D{}, TT{}, FF{}); }      \
      else { if (full) body(R{}, D{}, FF{}, TT{}); else body(R{}, D{}, FF{}, FF{}); }              \
    } while (0)
    if (relu && drop) UIC_EPI(TT, TT);
    else if (relu) UIC_EPI(TT, FF);
    else if (drop) UIC_EPI(FF, TT);
    else UIC_EPI(FF, FF);
#undef UIC_EPI
    return;
  }
  int arow[2][16];                       // addend row of every output row this lane holds (row % add_mod, once per row)
  if (p.addend) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) arow[i][reg] = (m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) % p.add_mod;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + (wn * 2 + j) * 32 + r32;
      if (col >= p.N) continue;
      float b = 0.f;
      if (p.bias) b += p.bias[col];
      if (p.bias2) b += p.bias2[col];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + (wm * 2 + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if (row >= p.M) continue;
        float v = acc[i][j][reg] + b;
        if (p.addend) v += p.addend[(size_t)arow[i][reg] * p.ld_add + col];
        if (p.flags & UIC_GEMM_RELU) v = fmaxf(v, 0.f);
        if (p.flags & UIC_GEMM_TANH) v = uic_tanh<T>(v);
        if (p.row_len) {
          const int n = row / p.R;
          if (row - n * p.R >= p.row_len[n]) v = 0.f;
        }
        if (p.drop_p > 0.f) v *= uic_drop_scale(p.seed, p.site, (unsigned)(row + p.drop_row0) * (unsigned)p.N + (unsigned)col, p.drop_p, inv_keep);
        const size_t o = (size_t)row * p.ldc + col;
        if (out_f32) {
          float* C = (float*)p.C;
          if (p.flags & UIC_GEMM_ACCUM) v += C[o];
          C[o] = v;
        } else {
          T* C = (T*)p.C;
          if (p.flags & UIC_GEMM_ACCUM) v += uic_to_f(C[o]);
          C[o] = uic_from_f<T>(v);
        }
      }
    }
}

template <typename T>
int launch_glds(const UicGemmParams& p, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_glds_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536),
                          "hipFuncSetAttribute(gemm glds)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_glds_kernel<T, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304),
                          "hipFuncSetAttribute(gemm glds ring)"));
    configured = true;
  }
  dim3 grid((p.M + 127) / 128, (p.N + 127) / 128, p.splitk > 1 ? p.splitk : 1);
  // the three-buffer ring where the launch is well under one workgroup per CU and the K loop has rounds to overlap.  Measured in the
  // training step (rocprofv3, median): the first logit chunk's d h (5 x 4 x 8 workgroups, 18 rounds) 22.7 -> 17.6 us, the BPTT loop's
  // d x1 (5 x 8 x 4, 8 rounds) 10.1 -> 9.8; its d x2 (5 x 12 x 4 = 240 workgroups) 13.2 -> 15.6: at 96 KB a CU takes ONE of them where
  // it took two 64-KB ones, and beside the side streams' 128-KB workgroups the launch then waits for 240 free CUs instead of 120 --
  // so the ring stops at 192 workgroups.  Alone on the chip it makes a K round cost 0.3 us instead of 0.8 (tools/gemm_headroom.py).
  const int rounds = p.seg[0].K / (128 / (int)sizeof(T)) / (p.splitk > 1 ? p.splitk : 1);
#ifdef UIC_GLDS_NO_RING           // (A/B builds: tools/build_variant.sh)
  const bool ring = false;
  (void)rounds;
#else
  const long wgs = (long)grid.x * grid.y * grid.z;
  const bool ring = ((wgs <= 192 && rounds >= 3) || (wgs <= 256 && rounds >= 32)) && !(p.flags & UIC_GEMM_NO_RING);   // (long K loops: also at one workgroup per CU)
#endif
  if (ring) hipLaunchKernelGGL((uic_gemm_glds_kernel<T, 3>), grid, dim3(256), 98304, s, p);
  else hipLaunchKernelGGL((uic_gemm_glds_kernel<T, 2>), grid, dim3(256), 65536, s, p);
  UIC_LAUNCH_CHECK("uic_gemm_glds_kernel");
  return UIC_OK;
}

template <typename T, int TM, int TN, int WM, int WN, int KS, bool LSTM, int PF = 1>
int launch_cfg(const UicGemmParams& p, hipStream_t s) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int stage_bytes = KS * (BM + BN) * LDS_STRIDE;
  constexpr int red_bytes = KS * (KS - 1) * WM * WN * TM * TN * (16 / KS) * 64 * 4;
  constexpr int lds = stage_bytes > red_bytes ? stage_bytes : red_bytes;
  static bool configured = false;
  if (!configured) {
    if (lds > 64 * 1024)
      UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)uic_gemm_kernel<T, TM, TN, WM, WN, KS, LSTM, PF>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(gemm)"));
    configured = true;
  }
  dim3 grid((p.M + BM - 1) / BM, LSTM ? (p.H + 32 * WN - 1) / (32 * WN) : (p.N + BN - 1) / BN);
  hipLaunchKernelGGL((uic_gemm_kernel<T, TM, TN, WM, WN, KS, LSTM, PF>), grid, dim3(64 * WM * WN * KS), lds, s, p);
  UIC_LAUNCH_CHECK("uic_gemm_kernel");
  return UIC_OK;
}

// Where the ping-pong kernel (gemm_pp.hip) is the faster of the two large-GEMM kernels.  It holds one workgroup per CU, so it needs
// enough tiles of its best height (uic_gemm_pp_rows) to cover most of the 256 CUs; measured (profiles/r04_*_gemm_headroom.txt,
// us, 128 x 128 kernel -> ping-pong): att_embed 23040 x 512 x 2048 66 -> 47.5 (192 rows), ctx2att 23040 x 512 x 512 29 -> 23,
// Gx 10880 x 2048 x 1024 70 -> 57, d xt 10880 x 512 x 2048 42 -> 32 (128 rows), logit chunk 2560 x 9488 x 512 50 -> 49; the
// BPTT loop's 640-row GEMMs (18-36 tiles) lose 2-3x and stay on the 128 x 128 / skinny kernels.
inline bool uic_gemm_pp_wins(const UicGemmParams& p) {
  if (!uic_gemm_pp_eligible(p) || p.seg[0].K < 512) return false;
  const int rows = uic_gemm_pp_rows(p.M, p.N, 256, p.seg[0].K / (p.splitk > 1 ? p.splitk : 1));
  const long tiles = (long)((p.M + rows - 1) / rows) * ((p.N + 255) / 256) * (p.splitk > 1 ? p.splitk : 1);
  return tiles >= 160;
}

template <typename T>
int launch_typed(const UicGemmParams& p, hipStream_t s) {
  // skinny problems (the per-decode-step GEMMs, M = rows of one step): 64-row tiles with a 4-way in-block K split
  // PF (register prefetch depth) stays 1: measured on MI355X, PF = 2 / 4 do not shorten these launches (their time is
  // launch + epilogue overhead plus ~0.9 us per K round, not exposed load latency) and PF = 4 slows the dX GEMMs.
  if (p.lstm == 2) return launch_cfg<T, 1, 5, 2, 1, 4, true>(p, s);
  if (p.lstm) return launch_cfg<T, 1, 4, 2, 1, 4, true>(p, s);
  const long blocks128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const bool glds_ok = p.nseg == 1 && p.seg[0].K % (128 / (int)sizeof(T)) == 0;
  if (p.C_exp2) {        // second output e^{2 C}: the ping-pong kernel's epilogue, or one element-wise pass behind any other kernel
    UIC_REQUIRE(sizeof(T) == 2 && !(p.flags & UIC_GEMM_OUT_F32) && p.C && p.ldc == p.N && ((size_t)p.M * p.N) % 8 == 0 && !p.slab,
                "gemm: C_exp2 needs a contiguous bf16 output");
    if (!(uic_gemm_pp_eligible(p) && ((p.flags & (UIC_GEMM_FORCE_256 | UIC_GEMM_FORCE_192 | UIC_GEMM_FORCE_PP128)) || p.a_f32 || p.acc_src || p.mask_act ||
                                      (!(p.flags & UIC_GEMM_FORCE_128) && uic_gemm_pp_wins(p))))) {
      UicGemmParams q = p;
      q.C_exp2 = nullptr;
      UIC_TRY(launch_typed<T>(q, s));
      return uic_exp2x2_launch(p.C, p.C_exp2, (size_t)p.M * p.N, s);
    }
  }
  if (p.a_f32) {
    UIC_REQUIRE(sizeof(T) == 2 && uic_gemm_pp_eligible(p), "gemm: an f32 A operand needs the ping-pong kernel (bf16, one K segment of whole 128-element rounds, < 4 GB)");
    return uic_gemm_pp_launch(p, 0, s);
  }
  if (p.acc_src || p.mask_act) {   // the fused backward-of-ReLU epilogue exists in the ping-pong kernel only (callers test uic_gemm_pp_eligible first)
    UIC_REQUIRE(sizeof(T) == 2 && !p.slab && uic_gemm_pp_eligible(p), "gemm: acc_src / mask_act need the ping-pong kernel (bf16, one K segment of whole 128-element rounds)");
    return uic_gemm_pp_launch(p, 0, s);
  }
  if (p.flags & UIC_GEMM_FORCE_256) return uic_gemm_pp_launch(p, 256, s);
  if (p.flags & UIC_GEMM_FORCE_192) return uic_gemm_pp_launch(p, 192, s);
  if (p.flags & UIC_GEMM_FORCE_PP128) return uic_gemm_pp_launch(p, 128, s);
  if (!(p.flags & UIC_GEMM_FORCE_128) && uic_gemm_pp_wins(p)) return uic_gemm_pp_launch(p, 0, s);
  if (p.slab) {
    UIC_REQUIRE(glds_ok && !p.lstm, "gemm: slab output needs one K segment that is a multiple of 128 bytes");
    return launch_glds<T>(p, s);
  }
  if (blocks128 >= 200 && glds_ok && !p.C_pre) return launch_glds<T>(p, s);
  if (blocks128 >= 200) return launch_cfg<T, 1 + 1, 2, 2, 2, 1, false>(p, s);
  // one row tile (M <= 64: the pivot NMT's per-step GEMMs at batch 64) with a long reduction: 8-way in-block K split on a
  // 64 x 32 tile -- the chain of dependent K rounds is what such a launch takes, and twice as many workgroups share the columns
  if (p.M <= 64) {
    int ktot = 0;
    for (int i = 0; i < p.nseg; ++i) ktot += p.seg[i].K;
    if (ktot * (int)sizeof(T) >= 2048) return launch_cfg<T, 1, 1, 2, 1, 8, false>(p, s);   // (K = 512 launches: no gain measured)
  }
  return launch_cfg<T, 1, 1, 2, 2, 4, false>(p, s);
}

}  // namespace

namespace {
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int splitk, int M, int N, int col0, int ncols,
                                     float* __restrict__ C, int ldc, int accumulate) {
  const size_t total = (size_t)M * ncols;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int row = (int)(i / ncols), c = (int)(i - (size_t)row * ncols);
    const float* src = slab + (size_t)row * N + col0 + c;
    float v = 0.f;
    for (int z = 0; z < splitk; ++z) v += src[(size_t)z * M * N];
    if (accumulate) v += C[(size_t)row * ldc + c];
    C[(size_t)row * ldc + c] = v;
  }
}
// four columns per lane, the row from the grid (see splitk_reduce_multi_vec_kernel in gemm_tn.hip)
__global__ void splitk_reduce_vec_kernel(const float* __restrict__ slab, int splitk, int M, int N, int col0, int ncols,
                                         float* __restrict__ C, int ldc, int accumulate) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= ncols) return;
  const size_t MN = (size_t)M * N;
  for (int row = blockIdx.y; row < M; row += gridDim.y) {
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
// the same sum with the output rows placed by a list: row r < map_rows of the slab goes to row map[r] of C (entries outside
// [0, map_limit) -- a list's -1 padding -- and rows >= map_rows are dropped).  The live-position logit layer's d hdrop (topdown.hip).
__global__ void splitk_reduce_rows_kernel(const float* __restrict__ slab, int splitk, int M, int N, const int* __restrict__ map, int map_rows,
                                          int map_limit, float* __restrict__ C, int ldc) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= N) return;
  const size_t MN = (size_t)M * N;
  for (int row = blockIdx.y; row < map_rows; row += gridDim.y) {
    const int orow = map[row];
    const float* src = slab + (size_t)row * N + c;
    float4 v = *(const float4*)src;
    for (int z = 1; z < splitk; ++z) {
      const float4 w = *(const float4*)(src + (size_t)z * MN);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if ((unsigned)orow < (unsigned)map_limit) *(float4*)(C + (size_t)orow * ldc + c) = v;
  }
}
}  // namespace

bool uic_splitk_reduce_rows_ok(const float* slab, int M, int N, const float* C, int ldc) {
  return N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)slab & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((size_t)M * N) % 4 == 0;
}
int uic_splitk_reduce_rows_launch(const float* slab, int splitk, int M, int N, const int* map, int map_rows, int map_limit, float* C, int ldc,
                                  hipStream_t s) {
  UIC_REQUIRE(map && map_rows <= M && uic_splitk_reduce_rows_ok(slab, M, N, C, ldc), "splitk_reduce_rows: bad arguments");
  if (map_rows <= 0) return UIC_OK;
  const int bt = N / 4 >= 256 ? 256 : ((N / 4 + 63) / 64) * 64;
  hipLaunchKernelGGL(splitk_reduce_rows_kernel, dim3((unsigned)((N / 4 + bt - 1) / bt), (unsigned)(map_rows > 65535 ? 65535 : map_rows)), dim3(bt), 0, s,
                     slab, splitk, M, N, map, map_rows, map_limit, C, ldc);
  UIC_LAUNCH_CHECK("splitk_reduce_rows");
  return UIC_OK;
}

int uic_splitk_reduce_launch(const float* slab, int splitk, int M, int N, int col0, int ncols, float* C, int ldc, hipStream_t s,
                             int accumulate) {
  if (M == 0 || ncols == 0) return UIC_OK;
  if (N % 4 == 0 && ncols % 4 == 0 && col0 % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)slab & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((size_t)M * N) % 4 == 0) {
    const int bt = ncols / 4 >= 256 ? 256 : ((ncols / 4 + 63) / 64) * 64;
    hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3((unsigned)((ncols / 4 + bt - 1) / bt), (unsigned)(M > 65535 ? 65535 : M)), dim3(bt), 0, s,
                       slab, splitk, M, N, col0, ncols, C, ldc, accumulate);
    UIC_LAUNCH_CHECK("splitk_reduce_vec");
    return UIC_OK;
  }
  size_t g = ((size_t)M * ncols + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, slab, splitk, M, N, col0, ncols, C, ldc, accumulate);
  UIC_LAUNCH_CHECK("splitk_reduce");
  return UIC_OK;
}

bool uic_gemm_glds_eligible(int dtype, int K) { return K > 0 && K % (dtype == UIC_BF16 ? 64 : 32) == 0; }

int uic_gemm_launch(const UicGemmParams& p, hipStream_t s) {
  UIC_REQUIRE(p.dtype == UIC_F32 || p.dtype == UIC_BF16, "gemm: bad dtype %d", p.dtype);
  UIC_REQUIRE(p.M >= 0 && p.N >= 0, "gemm: negative size");
  UIC_REQUIRE(p.nseg >= 1 && p.nseg <= UIC_GEMM_MAX_SEG, "gemm: nseg %d out of range", p.nseg);
  const int vec = p.dtype == UIC_BF16 ? 8 : 4;
  for (int i = 0; i < p.nseg; ++i) {
    const UicGemmSeg& g = p.seg[i];
    UIC_REQUIRE(g.A && g.B, "gemm: null operand in segment %d", i);
    // (an f32 A operand of the bf16 ping-pong kernel -- uic_linear_f32a -- is read in 16-byte pieces of FOUR floats: lda % 4)
    const int veca = p.a_f32 ? 4 : vec;
    UIC_REQUIRE(g.K > 0 && g.K % vec == 0 && g.lda % veca == 0 && g.ldb % vec == 0,
                "gemm: segment %d K=%d lda=%d ldb=%d must be multiples of %d elements (lda: %d)", i, g.K, g.lda, g.ldb, vec, veca);
    UIC_REQUIRE(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0, "gemm: segment %d operands must be 16-byte aligned", i);
  }
  if (p.lstm) {
    UIC_REQUIRE(p.H > 0 && p.N == (p.lstm == 2 ? 5 : 4) * p.H, "gemm(lstm): N=%d must equal %d*H (H=%d)", p.N, p.lstm == 2 ? 5 : 4, p.H);
    UIC_REQUIRE(p.c_out && p.h_out, "gemm(lstm): c_out and h_out are required");
  } else {
    UIC_REQUIRE(p.C != nullptr || p.slab != nullptr, "gemm: null C");
  }
  if (p.M == 0 || p.N == 0) return UIC_OK;
  return p.dtype == UIC_BF16 ? launch_typed<bf16_t>(p, s) : launch_typed<float>(p, s);
}
