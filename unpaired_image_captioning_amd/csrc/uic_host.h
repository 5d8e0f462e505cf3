// Host-side helpers shared by the model-level sequencers (topdown.hip, fcmodel.hip).
#pragma once
#include "uic_common.h"
#include <string.h>

namespace {

// bump allocator over a caller-provided arena (base == nullptr: size query)
struct Bump {
  char* base;
  size_t off;
  void* take(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

inline size_t rup8(size_t x) { return (x + 7) & ~(size_t)7; }
// padded vocabulary width (leading dimension of logits / dlogits): a multiple of 64 for real vocabularies so the
// K = V1 backward GEMM runs on the 128-byte-round LDS-DMA path, a multiple of 8 for toy sizes
inline size_t vpad(size_t v1) { return v1 >= 1024 ? (v1 + 63) & ~(size_t)63 : rup8(v1); }

inline const char* off(const void* p, size_t elems, int dtype) { return (const char*)p + elems * uic_dtype_size(dtype); }
inline char* offw(void* p, size_t elems, int dtype) { return (char*)p + elems * uic_dtype_size(dtype); }

inline UicGemmParams gemm_base(int dtype, int M, int N) {
  UicGemmParams g;
  memset(&g, 0, sizeof(g));
  g.dtype = dtype; g.M = M; g.N = N;
  return g;
}
inline void add_seg(UicGemmParams& g, const void* A, int lda, const void* B, int ldb, int K) {
  UicGemmSeg& s = g.seg[g.nseg++];
  s.A = A; s.B = B; s.K = K; s.lda = lda; s.ldb = ldb;
}

// Weight gradient(s) C_i = left[lrows, K] * right[cols_i, K]^T for one or several destinations that share `left`
// (right operands stacked row-wise in `right`).  Long-K, few-tile problems run split-K over workgroups on the
// LDS-DMA GEMM with deterministic slab reduction; anything else falls back to one direct GEMM per destination.
struct WDest { float* C; int ldc; int col0; int ncols; };
// rows: optional (one destination, no accumulate): the split-K path stores output row r < rows->n at row rows->map[r] of rows->C
// instead of dst (entries outside [0, rows->limit) dropped) and sets rows->used; the direct path ignores it and writes dst
struct WRows { const int* map; int n; int limit; float* C; int ldc; bool used; };
inline int wgrad_multi(float* slab, size_t slab_bytes, int dt, const void* left, int lrows, const void* right, int rrows, int K,
                const WDest* dst, int nd, hipStream_t s, bool accumulate = false, WRows* rows = nullptr) {
  if (rows) rows->used = false;
  const long blocks = (long)((lrows + 127) / 128) * ((rrows + 127) / 128);
  if (uic_gemm_glds_eligible(dt, K) && lrows >= 128 && rrows >= 128) {
    const int nt = K / (dt == UIC_BF16 ? 64 : 32);
    int sk = (int)((384 + blocks - 1) / blocks);
    if (sk > 8) sk = 8;
    if (sk > nt / 4) sk = nt / 4 > 0 ? nt / 4 : 1;
    // alone on the chip with a long reduction (the NMT generator's d out: 64 tiles x 782 rounds): one workgroup per CU, so that the
    // launch takes the 128 x 128 kernel's three-buffer ring (a K round then costs 0.3 us instead of 0.8: 87 -> 65 us)
    if (g_uic_tn_ring_off == 0 && blocks <= 128 && nt >= 128) {
      int one_per_cu = (int)(256 / blocks);
      if (one_per_cu > 8) one_per_cu = 8;
      if (one_per_cu >= 2 && nt / one_per_cu >= 32) sk = one_per_cu;
    }
    while (sk > 1 && (size_t)sk * lrows * rrows * 4 > slab_bytes) --sk;
    if ((size_t)sk * lrows * rrows * 4 <= slab_bytes && (sk > 1 || nd > 1 || accumulate)) {
      UicGemmParams g = gemm_base(dt, lrows, rrows);
      add_seg(g, left, K, right, K, K);
      g.splitk = sk; g.slab = slab;
      UIC_TRY(uic_gemm_launch(g, s));
#ifndef UIC_NO_REDUCE_ROWS        // (A/B builds: the separate scatter launch)
      if (rows && nd == 1 && !accumulate && dst[0].col0 == 0 && dst[0].ncols == rrows && rows->n <= lrows &&
          uic_splitk_reduce_rows_ok(slab, lrows, rrows, rows->C, rows->ldc)) {
        rows->used = true;
        return uic_splitk_reduce_rows_launch(slab, sk, lrows, rrows, rows->map, rows->n, rows->limit, rows->C, rows->ldc, s);
      }
#endif
      for (int i = 0; i < nd; ++i)
        UIC_TRY(uic_splitk_reduce_launch(slab, sk, lrows, rrows, dst[i].col0, dst[i].ncols, dst[i].C, dst[i].ldc, s, accumulate ? 1 : 0));
      return UIC_OK;
    }
  }
  for (int i = 0; i < nd; ++i) {
    UicGemmParams g = gemm_base(dt, lrows, dst[i].ncols);
    add_seg(g, left, K, (const char*)right + (size_t)dst[i].col0 * K * uic_dtype_size(dt), K, K);
    g.C = dst[i].C; g.ldc = dst[i].ldc; g.flags = UIC_GEMM_OUT_F32 | (accumulate ? UIC_GEMM_ACCUM : 0);
    UIC_TRY(uic_gemm_launch(g, s));
  }
  return UIC_OK;
}

// The same weight gradients WITHOUT transposed copies (bf16, gfx950 transposing LDS reads, gemm_tn.hip / gemm_tn_pp.hip):
// C_i = A[K, lrows]^T * [B_0 | B_1 | ...][K, cols].  Returns UIC_OK with *done = false when the shape is not eligible (the
// caller then transposes and uses wgrad_multi).
//
// Two kernels.  The 256 x 256 ping-pong form takes every problem with whole pairs of K tiles and >= 1024 reduction rows: it
// needs a quarter of the 128 x 128 kernel's workgroups for the same flops and half its LDS traffic.  How many workgroups a
// launch should have depends on where it runs (g_uic_tn_ring_off, set by the fused training step around the BPTT loop):
//   * beside the BPTT chain (CU-time-bound window): as FEW workgroups as the problem has 256 x 256 tiles -- the chain's kernels
//     need the other CUs --, split over K only until ~48 CUs work on it;
//   * alone on the chip (the step's tail): split over K until one workgroup per CU, slices of >= 4 K tiles.
// how: UIC_TN_FORCE_* | UIC_TN_SPLITK(n) for measurements and tests (0: the dispatcher's choice).
// K slices of a 256 x 256 launch.  Cost model fitted to tools/tn_bench.py (profiles/r05_*_tn_bench.txt): one workgroup per CU,
// so ceil(workgroups / 256) rounds of (K tiles per slice x 1.05 us + 8 us), plus the slab traffic of a split launch (partials
// written, read back, destination written) at 4 TB/s.  Slices hold a whole, even number of K tiles.
inline double tnpp_cost_us(long tiles, int nt, int sk, size_t out_bytes) {
  const long wgs = tiles * sk;
  const double body = (double)((wgs + 255) / 256) * ((double)(nt / sk) * 1.05 + 8.0);
  const double slab = sk > 1 ? ((double)out_bytes * (2.0 * sk + 1.0)) / 4.0e6 + 3.0 : (double)out_bytes / 4.0e6;
  return body + slab;
}
inline int tnpp_splitk(long tiles, int nt, size_t out_bytes, size_t slab_bytes, bool beside) {
  int best = 1;
  if (beside) {
    // as few workgroups as the problem has tiles; split only a problem of a handful of tiles, until ~24 CUs work on it
    if (tiles >= 24) return 1;
    for (int sk = 2; sk <= 64; ++sk) {
      if (nt % (2 * sk) != 0 || nt / sk < 4) continue;
      if ((size_t)sk * out_bytes > slab_bytes) break;
      best = sk;
      if (tiles * sk >= 24) break;
    }
    return best;
  }
  double best_cost = tnpp_cost_us(tiles, nt, 1, out_bytes);
  for (int sk = 2; sk <= 64; ++sk) {
    if (nt % (2 * sk) != 0 || nt / sk < 4) continue;
    if ((size_t)sk * out_bytes > slab_bytes || tiles * sk > 1024) break;
    const double c = tnpp_cost_us(tiles, nt, sk, out_bytes);
    if (c < best_cost) { best_cost = c; best = sk; }
  }
  return best;
}
inline int wgrad_tn(float* slab, size_t slab_bytes, int dt, const void* A, int lda, int lrows, const UicGemmTnSeg* segs, int nseg,
                    int K, const WDest* dst, int nd, hipStream_t s, bool accumulate, bool* done, int how = 0) {
  *done = false;
  if (dt != UIC_BF16 || nseg > UIC_GEMM_TN_MAX_SEG) return UIC_OK;
#ifdef UIC_TNPP_OFF       // (A/B builds: tools/build_variant.sh)
  if (!(how & UIC_TN_FORCE_256)) how |= UIC_TN_FORCE_128;
#endif
  UicGemmTnParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.M = lrows; p.K = K; p.nseg = nseg;
  int rrows = 0;
  for (int i = 0; i < nseg; ++i) { p.seg[i] = segs[i]; rrows += segs[i].ncols; }
  p.N = rrows;
  if (!uic_gemm_tn_eligible(p)) return UIC_OK;
  const int nt = K / 64;
  const int sk_forced = (how >> 16) & 0xff;
  const size_t out_bytes = (size_t)lrows * rrows * 4;
  auto set_dst = [&]() {
    p.ndst = nd; p.accumulate = accumulate ? 1 : 0;
    for (int i = 0; i < nd; ++i) { p.dst[i].C = dst[i].C; p.dst[i].ldc = dst[i].ldc; p.dst[i].col0 = dst[i].col0; p.dst[i].ncols = dst[i].ncols; }
  };
  auto reduce = [&](int sk) -> int {
    UicSlabDest sd[4];
    for (int i0 = 0; i0 < nd; i0 += 4) {
      const int n = nd - i0 < 4 ? nd - i0 : 4;
      for (int i = 0; i < n; ++i) sd[i] = UicSlabDest{dst[i0 + i].C, dst[i0 + i].ldc, dst[i0 + i].col0, dst[i0 + i].ncols};
      UIC_TRY(uic_splitk_reduce_multi_launch(slab, sk, lrows, rrows, sd, n, accumulate ? 1 : 0, s));
    }
    return UIC_OK;
  };
  // ---- the 256 x 256 ping-pong kernel
  if (!(how & UIC_TN_FORCE_128)) {
    const long t256 = (long)((lrows + 255) / 256) * ((rrows + 255) / 256);
    int sk = sk_forced ? sk_forced : tnpp_splitk(t256, nt, out_bytes, slab_bytes, g_uic_tn_ring_off != 0);
    UicGemmTnParams q = p;
    q.splitk = sk; q.slab = slab;
    // beside the BPTT chain every problem with a real K loop; alone on the chip the 128 x 128 kernel's many small workgroups
    // finish a small problem sooner (tools/tn_bench.py: LSTM chunk 31 vs 45 us, ctx2att 36 vs 45; logit 181 vs 146, att_embed 79 vs 70)
    const bool beside = g_uic_tn_ring_off != 0;
    bool wanted = (how & UIC_TN_FORCE_256) || (lrows >= 256 && (beside ? K >= 1024 : 2.0 * lrows * rrows * K >= 3.0e10));
    if (beside && !(how & UIC_TN_FORCE_256)) {
      const bool logit = K > 8192;
      if ((g_uic_knobs & UIC_KNOB_CHUNK_TN128) && !logit) wanted = false;
      if ((g_uic_knobs & UIC_KNOB_LOGIT_TN128) && logit) wanted = false;
      if ((g_uic_knobs & UIC_KNOB_CHUNK_SK2) && !logit && !sk_forced && sk == 1 && nt % 4 == 0) { sk = 2; q.splitk = 2; }
    }
    const bool direct = sk == 1 && nd <= UIC_GEMM_TN_MAX_SEG;
    if (direct) { q.ndst = nd; q.accumulate = accumulate ? 1 : 0; for (int i = 0; i < nd; ++i) { q.dst[i].C = dst[i].C; q.dst[i].ldc = dst[i].ldc; q.dst[i].col0 = dst[i].col0; q.dst[i].ncols = dst[i].ncols; } }
    const bool fits = direct || (size_t)sk * out_bytes <= slab_bytes;
    if (wanted && fits && uic_gemm_tnpp_eligible(q)) {
      UIC_TRY(uic_gemm_tnpp_launch(q, s));
      if (!direct) UIC_TRY(reduce(sk));
      *done = true;
      return UIC_OK;
    }
    UIC_REQUIRE(!(how & UIC_TN_FORCE_256), "wgrad_tn: M=%d N=%d K=%d splitk=%d is not a problem of the 256 x 256 kernel (K %% (128 splitk), slab of %zu bytes)",
                lrows, rrows, K, sk, slab_bytes);
  }
  // ---- the 128 x 128 kernel
  const long blocks = (long)((lrows + 127) / 128) * ((rrows + 127) / 128);
  int sk = blocks >= 160 ? 1 : (int)((384 + blocks - 1) / blocks);   // >= 160 tiles fill the 256 CUs well enough: no slab pass
  if (sk > 8) sk = 8;
  if (sk > nt / 4) sk = nt / 4 > 0 ? nt / 4 : 1;
  if (sk_forced) sk = sk_forced;
  while (sk > 1 && (size_t)sk * out_bytes > slab_bytes) --sk;
  if (sk > 1 && (size_t)sk * out_bytes > slab_bytes) return UIC_OK;
  p.splitk = sk; p.slab = slab;
  if (sk == 1 && nd <= UIC_GEMM_TN_MAX_SEG) {
    set_dst();
    UIC_TRY(uic_gemm_tn_launch(p, s));
  } else {
    if ((size_t)sk * out_bytes > slab_bytes) return UIC_OK;
    UIC_TRY(uic_gemm_tn_launch(p, s));
    UIC_TRY(reduce(sk));
  }
  *done = true;
  return UIC_OK;
}

// One group of weight gradients sharing the left operand:  C_i = A[rows, lrows]^T * [B_0 | B_1 | ...][rows, cols].
// bf16 on eligible shapes: gemm_tn.hip reads both operands as they lie (transposing LDS reads).  Otherwise (f32 parity path,
// odd sizes): transposed copies into tA [lrows, rows^8] / tB [cols, rows^8] and the NT kernels (wgrad_multi).
inline int wgrad_group(float* slab, size_t slab_bytes, int dt, const void* A, int lda, int lrows, const UicGemmTnSeg* segs, int nseg,
                       int rows, const WDest* dst, int nd, hipStream_t s, bool accumulate, void* tA, void* tB) {
  bool done = false;
  UIC_TRY(wgrad_tn(slab, slab_bytes, dt, A, lda, lrows, segs, nseg, rows, dst, nd, s, accumulate, &done));
  if (done) return UIC_OK;
  const int Kp = (int)rup8(rows);
  UIC_TRY(uic_transpose_launch(dt, A, rows, lrows, lda, tA, Kp, s));
  int col = 0;
  for (int i = 0; i < nseg; ++i) {
    UIC_TRY(uic_transpose_launch(dt, segs[i].B, rows, segs[i].ncols, segs[i].ldb, offw(tB, (size_t)col * Kp, dt), Kp, s));
    col += segs[i].ncols;
  }
  return wgrad_multi(slab, slab_bytes, dt, tA, lrows, tB, col, Kp, dst, nd, s, accumulate);
}

}  // namespace
