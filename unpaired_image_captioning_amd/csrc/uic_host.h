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
inline int wgrad_multi(float* slab, size_t slab_bytes, int dt, const void* left, int lrows, const void* right, int rrows, int K,
                const WDest* dst, int nd, hipStream_t s, bool accumulate = false) {
  const long blocks = (long)((lrows + 127) / 128) * ((rrows + 127) / 128);
  if (uic_gemm_glds_eligible(dt, K) && lrows >= 128 && rrows >= 128) {
    const int nt = K / (dt == UIC_BF16 ? 64 : 32);
    int sk = (int)((384 + blocks - 1) / blocks);
    if (sk > 8) sk = 8;
    if (sk > nt / 4) sk = nt / 4 > 0 ? nt / 4 : 1;
    while (sk > 1 && (size_t)sk * lrows * rrows * 4 > slab_bytes) --sk;
    if ((size_t)sk * lrows * rrows * 4 <= slab_bytes && (sk > 1 || nd > 1 || accumulate)) {
      UicGemmParams g = gemm_base(dt, lrows, rrows);
      add_seg(g, left, K, right, K, K);
      g.splitk = sk; g.slab = slab;
      UIC_TRY(uic_gemm_launch(g, s));
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

// The same weight gradients WITHOUT transposed copies (bf16, gfx950 transposing LDS reads, gemm_tn.hip):
// C_i = A[K, lrows]^T * [B_0 | B_1 | ...][K, cols].  Returns UIC_OK with *done = false when the shape is not eligible (the
// caller then transposes and uses wgrad_multi).
inline int wgrad_tn(float* slab, size_t slab_bytes, int dt, const void* A, int lda, int lrows, const UicGemmTnSeg* segs, int nseg,
                    int K, const WDest* dst, int nd, hipStream_t s, bool accumulate, bool* done) {
  *done = false;
  if (dt != UIC_BF16 || nseg > UIC_GEMM_TN_MAX_SEG) return UIC_OK;
  UicGemmTnParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.M = lrows; p.K = K; p.nseg = nseg;
  int rrows = 0;
  for (int i = 0; i < nseg; ++i) { p.seg[i] = segs[i]; rrows += segs[i].ncols; }
  p.N = rrows;
  if (!uic_gemm_tn_eligible(p)) return UIC_OK;
  const long blocks = (long)((lrows + 127) / 128) * ((rrows + 127) / 128);
  const int nt = K / 64;
  int sk = blocks >= 160 ? 1 : (int)((384 + blocks - 1) / blocks);   // >= 160 tiles fill the 256 CUs well enough: no slab pass
  if (sk > 8) sk = 8;
  if (sk > nt / 4) sk = nt / 4 > 0 ? nt / 4 : 1;
  while (sk > 1 && (size_t)sk * lrows * rrows * 4 > slab_bytes) --sk;
  if (sk > 1 && (size_t)sk * lrows * rrows * 4 > slab_bytes) return UIC_OK;
  p.splitk = sk; p.slab = slab;
  if (sk == 1 && nd <= UIC_GEMM_TN_MAX_SEG) {
    p.ndst = nd; p.accumulate = accumulate ? 1 : 0;
    for (int i = 0; i < nd; ++i) { p.dst[i].C = dst[i].C; p.dst[i].ldc = dst[i].ldc; p.dst[i].col0 = dst[i].col0; p.dst[i].ncols = dst[i].ncols; }
    UIC_TRY(uic_gemm_tn_launch(p, s));
  } else {
    if ((size_t)sk * lrows * rrows * 4 > slab_bytes) return UIC_OK;
    UIC_TRY(uic_gemm_tn_launch(p, s));
    UicSlabDest sd[4];
    for (int i0 = 0; i0 < nd; i0 += 4) {
      const int n = nd - i0 < 4 ? nd - i0 : 4;
      for (int i = 0; i < n; ++i) sd[i] = UicSlabDest{dst[i0 + i].C, dst[i0 + i].ldc, dst[i0 + i].col0, dst[i0 + i].ncols};
      UIC_TRY(uic_splitk_reduce_multi_launch(slab, sk, lrows, rrows, sd, n, accumulate ? 1 : 0, s));
    }
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
