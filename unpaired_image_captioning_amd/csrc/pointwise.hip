// HBM-bound helper kernels of the TopDown training / decoding path (gfx950).
// Every kernel cites the reference lines whose arithmetic it carries.
#include "uic_common.h"
#include <type_traits>
#include <string.h>
#include <stdlib.h>
#include "../../include/uic_hip.h"

namespace {

constexpr int NT = 256;

inline int grid_for(size_t n, int per_block) {
  size_t g = (n + per_block - 1) / per_block;
  if (g > 65536) g = 65536;   // grid-stride beyond that
  if (g == 0) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------ casts / fills
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      // the f32 source (master parameters, the loader's features) is not read again this step: streamed
      const float4 v = make_float4(__builtin_nontemporal_load(src + i), __builtin_nontemporal_load(src + i + 1),
                                   __builtin_nontemporal_load(src + i + 2), __builtin_nontemporal_load(src + i + 3));
      if constexpr (sizeof(T) == 2) {
        *(uint2*)(dst + i) = make_uint2(uic_pack_bf16x2(v.x, v.y), uic_pack_bf16x2(v.z, v.w));
      } else {
        *(float4*)(dst + i) = v;
      }
    } else {
      for (size_t j = i; j < n; ++j) dst[j] = uic_from_f<T>(src[j]);
    }
  }
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = uic_to_f(src[i]);
}

// ------------------------------------------------------------------ transpose
template <typename T>
__global__ __launch_bounds__(NT) void transpose_kernel(const T* __restrict__ src, int rows, int cols, int lds,
                                                       T* __restrict__ dst, int ldd) {
  __shared__ T tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    T v = uic_from_f<T>(0.f);
    if (r < rows && c < cols) v = src[(size_t)r * lds + c];
    tile[ty + 4 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < cols && r < ldd) dst[(size_t)c * ldd + r] = tile[tx][ty + 4 * i];
  }
}

// several transposes in one launch (blockIdx.z picks the job; every field by a uniform select -- indexing the by-value
// table would put it in private memory): the weight refresh's transposes are launch-bound, not byte-bound
struct TransposeMulti { UicTransposeJob j[UIC_TRANSPOSE_MULTI]; int start[UIC_TRANSPOSE_MULTI + 1]; };   // start[k]: first 64 x 64 tile of job k
#define UIC_TSEL2(f) (k == 0 ? c.f[0] : k == 1 ? c.f[1] : k == 2 ? c.f[2] : k == 3 ? c.f[3] : k == 4 ? c.f[4] : k == 5 ? c.f[5] : \
                      k == 6 ? c.f[6] : k == 7 ? c.f[7] : k == 8 ? c.f[8] : k == 9 ? c.f[9] : k == 10 ? c.f[10] : c.f[11])
#define UIC_TSEL(f) (k == 0 ? c.j[0].f : k == 1 ? c.j[1].f : k == 2 ? c.j[2].f : k == 3 ? c.j[3].f : k == 4 ? c.j[4].f : k == 5 ? c.j[5].f : \
                     k == 6 ? c.j[6].f : k == 7 ? c.j[7].f : k == 8 ? c.j[8].f : k == 9 ? c.j[9].f : k == 10 ? c.j[10].f : c.j[11].f)
template <typename T>
__global__ __launch_bounds__(NT) void transpose_multi_kernel(const TransposeMulti c) {
  __shared__ T tile[64][66];
  // the job of this tile: the grid is the concatenation of the jobs' tile lists (one block per 64 x 64 tile, none idle)
  int k = 0;
#pragma unroll
  for (int q = 1; q < UIC_TRANSPOSE_MULTI; ++q) k += (int)blockIdx.x >= c.start[q] ? 1 : 0;
  const T* __restrict__ src = (const T*)UIC_TSEL(src);
  T* __restrict__ dst = (T*)UIC_TSEL(dst);
  const int rows = UIC_TSEL(rows), cols = UIC_TSEL(cols), lds = UIC_TSEL(lds), ldd = UIC_TSEL(ldd);
  const int t = blockIdx.x - UIC_TSEL2(start), tx_n = (ldd + 63) / 64;
  const int r0 = (t % tx_n) * 64, c0 = (t / tx_n) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, cc = c0 + tx;
    T v = uic_from_f<T>(0.f);
    if (r < rows && cc < cols) v = src[(size_t)r * lds + cc];
    tile[ty + 4 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int cc = c0 + ty + 4 * i, r = r0 + tx;
    if (cc < cols && r < ldd) dst[(size_t)cc * ldd + r] = tile[tx][ty + 4 * i];
  }
}
#undef UIC_TSEL
#undef UIC_TSEL2

// ------------------------------------------------------------------ column sums (bias gradients)
// stage 1: a block sums `rows_per_block` rows of a 64-column strip; its 4 waves take rows r, r+4, ...
template <typename T>
__global__ __launch_bounds__(NT) void colsum_stage1(const T* __restrict__ src, int rows, int cols, int lds,
                                                    int rows_per_block, float* __restrict__ part) {
  __shared__ float s_part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int rb = blockIdx.y;
  const int r_lo = rb * rows_per_block;
  const int r_hi = min(rows, r_lo + rows_per_block);
  float s = 0.f;
  if (c < cols) {
#pragma unroll 4
    for (int r = r_lo + wave; r < r_hi; r += 4) s += uic_to_f(src[(size_t)r * lds + c]);
  }
  s_part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < cols) part[(size_t)rb * cols + c] = s_part[0][lane] + s_part[1][lane] + s_part[2][lane] + s_part[3][lane];
}
// stage 2: 64 columns per block; the 4 waves split the row-block partials (fixed order -> deterministic)
__global__ __launch_bounds__(NT) void colsum_stage2(const float* __restrict__ part, int nrb, int cols, float* __restrict__ out) {
  __shared__ float s_part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < cols) {
#pragma unroll 4
    for (int rb = wave; rb < nrb; rb += 4) s += part[(size_t)rb * cols + c];
  }
  s_part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < cols) out[c] = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
}

template <typename T>
__global__ void sum_steps_kernel(const T* __restrict__ src, int TS, size_t step, T* __restrict__ dst) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < step; i += stride) {
    float s = 0.f;
    for (int t = 0; t < TS; ++t) s += uic_to_f(src[(size_t)t * step + i]);
    dst[i] = uic_from_f<T>(s);
  }
}
// 16-byte chunks (step_elems a multiple of the vector width, 16-byte aligned pointers): one load instruction per step
template <typename T>
__global__ void sum_steps_vec_kernel(const T* __restrict__ src, int TS, size_t step, T* __restrict__ dst) {
  constexpr int VEC = uic_vec<T>::N;
  const size_t nch = step / VEC;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nch; c += stride) {
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int t = 0; t < TS; ++t) {
      const uint4 v = *(const uint4*)(src + (size_t)t * step + c * VEC);
      float f[VEC];
      uic_unpack<T>(v, f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += f[j];
    }
    *(uint4*)(dst + c * VEC) = uic_pack<T>(acc);
  }
}

// ------------------------------------------------------------------ word embedding
// self.embed = Embedding + ReLU + Dropout (P/models/AttModel.py:73-75,160), all T steps at once:
// out[(t*N+n), :] = dropout(relu(table[tokens[n, t]]))
// (the table comes as f32 masters or in the operand dtype: bf16 runs of the captioner gather from the bf16 copy of the table that
// the weight refresh keeps beside the other operand copies -- the copy a data-parallel rank receives from the all-gather)
__device__ __forceinline__ void uic_table_load4(const float* p, float* f) { const float4 v = *(const float4*)p; f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
__device__ __forceinline__ void uic_table_load4(const bf16_t* p, float* f) {
  const uint2 v = *(const uint2*)p;
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u); f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
__device__ __forceinline__ void uic_table_load8(const float* p, float* f) { uic_table_load4(p, f); uic_table_load4(p + 4, f + 4); }
__device__ __forceinline__ void uic_table_load8(const bf16_t* p, float* f) {
  const uint4 v = *(const uint4*)p;
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(w[j] << 16); f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
}
template <typename T, typename TT>
__global__ void embed_fwd_kernel(const TT* __restrict__ table, int V1, int E, const int64_t* __restrict__ tokens,
                                 int ldtok, int N, int TS, float drop_p, unsigned seed, unsigned site, size_t idx_base, int relu, T* __restrict__ out) {
  const int e4 = E / 4;
  const size_t total = (size_t)TS * N * e4;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / e4;
    const int c = (int)(i - row * e4) * 4;
    const int t = (int)(row / N), n = (int)(row - (size_t)t * N);
    long tok = tokens[(size_t)n * ldtok + t];
    if (tok < 0 || tok >= V1) tok = 0;
    float f[4];
    uic_table_load4(table + (size_t)tok * E + c, f);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = fmaxf(f[j], 0.f);
    }
    if (drop_p > 0.f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] *= uic_drop_scale(seed, site, (unsigned)(idx_base + row * E + c + j), drop_p, inv_keep);
    }
    T* o = out + row * E + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = uic_from_f<T>(f[j]);
  }
}
// The same for bf16 outputs with E % 8 == 0: one WAVE per embedded row (the token is read once per wave, not once per lane;
// no 64-bit divisions), a lane takes 8 consecutive elements -- two 16-byte loads, ONE 16-byte store (the element-wise form
// above stores four 2-byte values per lane and ran at 0.7 TB/s: 48 us for the step's 10880 rows).  Same values, same dropout
// decisions (the hash is per element index).
template <typename TT>
__global__ __launch_bounds__(256) void embed_fwd_rows_bf16_kernel(const TT* __restrict__ table, int V1, int E, const int64_t* __restrict__ tokens,
                                                                  int ldtok, int N, int rows, float drop_p, unsigned seed, unsigned site, size_t idx_base,
                                                                  int relu, bf16_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int e8 = E >> 3;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const int t = row / N, n = row - t * N;
    long tok = tokens[(size_t)n * ldtok + t];
    if (tok < 0 || tok >= V1) tok = 0;
    const TT* src = table + (size_t)tok * E;
    bf16_t* dst = out + (size_t)row * E;
    for (int c = lane; c < e8; c += 64) {
      float f[8];
      uic_table_load8(src + c * 8, f);
      if (relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
      }
      if (drop_p > 0.f) {
        const unsigned base = (unsigned)(idx_base + (size_t)row * E + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= uic_drop_scale(seed, site, base + (unsigned)j, drop_p, inv_keep);
      }
      *(uint4*)(dst + c * 8) = uic_pack<bf16_t>(f);
    }
  }
}
template <typename T>
__global__ void relu_mask_bwd_kernel(const float* __restrict__ g, const T* __restrict__ act, float scale,
                                     T* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = uic_from_f<T>(uic_to_f(act[i]) > 0.f ? g[i] * scale : 0.f);
}

// The same gradient with the T*N positions first bucketed by token -- a STABLE counting sort (positions of one token stay in
// position order), so that every run of the repository is bit for bit the same: histogram per block of EMB_BLK positions ->
// per-token prefix over the blocks -> exclusive scan over the tokens -> fill (slot = bucket start + entries of earlier blocks +
// rank inside the block, counted in LDS).  Equal tokens become neighbours, so embed_gather_kernel can sum them in registers, and
// every table row has ONE owner that stores it (no floating-point atomics anywhere).
// Wave-aggregated counting: a token shared by many lanes of a wave (the padding token at late decode steps, frequent words)
// costs ONE atomic per wave instead of one per lane -- plain per-lane atomics on cnt[0] serialise ~1000 deep.  Two aggregation
// rounds (the tokens of the first two still-active lanes) catch the hot tokens; the remaining lanes use plain atomics.
// (integer adds whose old value nobody reads: the totals do not depend on arrival order)
__device__ __forceinline__ void wave_token_add(int* base, int tok, bool active) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const unsigned long long todo = __ballot(active);
    if (todo == 0) break;
    const int leader = __ffsll((long long)todo) - 1;
    const int t0 = __shfl(tok, leader, 64);
    const unsigned long long m = __ballot(active && tok == t0);
    if (lane == leader) atomicAdd(base + t0, __popcll(m));
    if (active && tok == t0) active = false;
  }
  if (active) atomicAdd(base + tok, 1);
}
constexpr int EMB_BLK = 1024;                     // positions per histogram block (one workgroup)
// split > 0: the bucket key is (t >= split) * V1 + token -- the positions of decode steps [0, split) are entries [0, split N) of the
// list, those of [split, T) the rest, and each half can be gathered on its own (the later steps' half while BPTT still runs)
__device__ __forceinline__ int embed_key(const int64_t* __restrict__ tokens, int ldtok, int N, int V1, int split, int i) {
  const int t = i / N, n = i - t * N;
  long tok = tokens[(size_t)n * ldtok + t];
  if (tok < 0 || tok >= V1) tok = 0;
  if (split && t >= split) tok += V1;
  return (int)tok;
}
// cntb[b, key] = how many of block b's positions carry `key`
__global__ __launch_bounds__(EMB_BLK) void embed_hist_kernel(const int64_t* __restrict__ tokens, int ldtok, int N, int TS, int V1, int split,
                                                             int nkeys, int* __restrict__ cntb) {
  const int i = blockIdx.x * EMB_BLK + threadIdx.x;          // whole waves enter the aggregation together
  const bool ok = i < TS * N;
  const int key = ok ? embed_key(tokens, ldtok, N, V1, split, i) : 0;
  wave_token_add(cntb + (size_t)blockIdx.x * nkeys, key, ok);
}
// per key: cntb[b, key] <- entries of blocks < b (exclusive prefix over the blocks), cnt[key] <- the key's total
__global__ void embed_block_prefix_kernel(int* __restrict__ cntb, int nkeys, int nblk, int* __restrict__ cnt) {
  const int key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= nkeys) return;
  int run = 0;
  for (int b = 0; b < nblk; ++b) {
    const int v = cntb[(size_t)b * nkeys + key];
    cntb[(size_t)b * nkeys + key] = run;
    run += v;
  }
  cnt[key] = run;
}
// single workgroup: off[v] = exclusive prefix of cnt, off[V1] = total.  Rows of 1024 consecutive keys (coalesced loads), each
// scanned by wave shuffles + one LDS hop, with the running total carried from row to row (a thread that walked its own run of
// ~50 consecutive keys -- strided loads, one at a time -- took 72 us on the pivot NMT's 100 008 keys).
__global__ __launch_bounds__(1024) void embed_scan_kernel(const int* __restrict__ cnt, int V1, int* __restrict__ off) {
  // four consecutive keys per thread and trip (one 16-byte load): 4096 keys per trip, 13 trips for the pivot NMT's 50 005 keys
  // (one key per thread: 49 trips of shuffle scan + LDS hop + barrier = 40 us)
  __shared__ int s_wave[2][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int carry = 0;
  auto load4 = [&](int v0, int (&x)[4]) {
    if (v0 + 3 < V1) {
      const int4 q = *(const int4*)(cnt + v0);          // (cnt is 16-byte aligned and v0 a multiple of 4)
      x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = v0 + k < V1 ? cnt[v0 + k] : 0;
    }
  };
  int nxt[4];
  load4((int)threadIdx.x * 4, nxt);
  for (int base = 0, it = 0; base < V1; base += 4096, ++it) {
    const int v0 = base + (int)threadIdx.x * 4;
    const int x[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
    if (base + 4096 < V1) load4(v0 + 4096, nxt);          // (the next trip's load passes behind this trip's scan)
    const int t4 = (x[0] + x[1]) + (x[2] + x[3]);
    int incl = t4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    int* sw = s_wave[it & 1];
    if (lane == 63) sw[wv] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int t = sw[k];
      if (k < wv) woff += t;
      tot += t;
    }
    int run = carry + woff + incl - t4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (v0 + k < V1) off[v0 + k] = run;
      run += x[k];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) off[V1] = carry;
}
// perm[off[key] + (entries of earlier blocks) + (earlier entries of this block with the same key)] = position
__global__ __launch_bounds__(EMB_BLK) void embed_fill_kernel(const int64_t* __restrict__ tokens, int ldtok, int N, int TS, int V1, int split,
                                                             int nkeys, const int* __restrict__ off, const int* __restrict__ cntb,
                                                             int* __restrict__ perm) {
  __shared__ __attribute__((aligned(16))) int s_key[EMB_BLK];
  const int i = blockIdx.x * EMB_BLK + threadIdx.x;
  const bool ok = i < TS * N;
  const int key = ok ? embed_key(tokens, ldtok, N, V1, split, i) : -1;
  s_key[threadIdx.x] = key;
  __syncthreads();
  if (!ok) return;
  int rank = 0;
  const int full = threadIdx.x >> 2;
  for (int q = 0; q < full; ++q) {                 // (LDS broadcast reads: every lane of a wave asks for the same 16 bytes)
    const int4 v = ((const int4*)s_key)[q];
    rank += (v.x == key) + (v.y == key) + (v.z == key) + (v.w == key);
  }
  for (int j = full * 4; j < (int)threadIdx.x; ++j) rank += s_key[j] == key;
  perm[off[key] + cntb[(size_t)blockIdx.x * nkeys + key] + rank] = i;
}
// One workgroup per CH consecutive entries of the token-bucketed position list: it loads its CH rows up front and sums runs of
// equal tokens in registers.  A bucket that lies wholly inside the workgroup's entries (most tokens occur once or twice) is
// stored straight into the table.  A bucket that straddles workgroups (a hot token -- the padding token 0 owns thousands of
// positions) leaves one partial row per workgroup in `part` ([2, nchunks, E]: plane 0 = the bucket began in an earlier
// workgroup, plane 1 = it begins here and goes on), and embed_gather_finish_kernel lets the workgroup in which the bucket begins
// add them up in list order: one owner per table row, a fixed summation order.
constexpr int EMB_CH = 16;
template <typename T>
__global__ __launch_bounds__(128) void embed_gather_kernel(const float* __restrict__ dxt, const T* __restrict__ xt, const int64_t* __restrict__ tokens,
                                                           int ldtok, int N, int V1, const int* __restrict__ off, const int* __restrict__ perm, int total,
                                                           int E, float inv_keep, long skip_token, float* __restrict__ dtable, int base, int keybase,
                                                           int accum, float* __restrict__ part, int chunk0, size_t plane) {
  // entries [base, total) of the list; keybase = this chunk's first bucket key; accum: the table already holds earlier chunks' sums
  const int start = base + blockIdx.x * EMB_CH;
  const int cnt = min(EMB_CH, total - start);
  __shared__ int s_pos[EMB_CH];
  __shared__ int s_tok[EMB_CH];
  __shared__ int s_b0[EMB_CH], s_b1[EMB_CH];          // the bucket bounds of entry j's token (round 6: fetched here, by the thread that
                                                      // has the token, instead of one dependent load pair per run in the loop below)
  if (threadIdx.x < cnt) {
    const int pos = perm[start + threadIdx.x];
    const int t = pos / N, n = pos - t * N;
    long tok = tokens[(size_t)n * ldtok + t];
    if (tok < 0 || tok >= V1) tok = 0;
    s_pos[threadIdx.x] = pos;
    s_tok[threadIdx.x] = (int)tok;
    s_b0[threadIdx.x] = off[keybase + (int)tok];
    s_b1[threadIdx.x] = off[keybase + (int)tok + 1];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < E / 4; c += blockDim.x) {
    // All 2 x EMB_CH loads of the lane requested before the first is used (round 6): entries behind `cnt` re-read the last valid
    // row and are dropped by a select, the ReLU mask comes from a stand-in (the gradient itself) when there is no xt -- behind
    // `if (j < cnt)` / `if (xt)` hipcc waited for every load at its branch's join: 32 dependent HBM round trips per workgroup.
    float4 g[EMB_CH];
    typename std::conditional<sizeof(T) == 2, uint2, float4>::type a[EMB_CH];
    const char* const xsrc = xt ? (const char*)xt : (const char*)dxt;
    const size_t xscale = xt ? sizeof(T) : sizeof(float);
#pragma unroll
    for (int j = 0; j < EMB_CH; ++j) {
      const size_t o = (size_t)s_pos[j < cnt ? j : cnt - 1] * E + c * 4;
      g[j] = *(const float4*)(dxt + o);
      if constexpr (sizeof(T) == 2) a[j] = *(const uint2*)(xsrc + o * xscale);        // 4 bf16 in one 8-byte load
      else a[j] = *(const float4*)(xsrc + o * xscale);
    }
#pragma unroll
    for (int j = 0; j < EMB_CH; ++j) {
      float a4[4];
      if constexpr (sizeof(T) == 2) {
        a4[0] = __uint_as_float(a[j].x << 16); a4[1] = __uint_as_float(a[j].x & 0xffff0000u);
        a4[2] = __uint_as_float(a[j].y << 16); a4[3] = __uint_as_float(a[j].y & 0xffff0000u);
      } else {
        a4[0] = a[j].x; a4[1] = a[j].y; a4[2] = a[j].z; a4[3] = a[j].w;
      }
      const bool live = j < cnt, relu = xt != nullptr;
      g[j].x = live && !(relu && !(a4[0] > 0.f)) ? g[j].x : 0.f;
      g[j].y = live && !(relu && !(a4[1] > 0.f)) ? g[j].y : 0.f;
      g[j].z = live && !(relu && !(a4[2] > 0.f)) ? g[j].z : 0.f;
      g[j].w = live && !(relu && !(a4[3] > 0.f)) ? g[j].w : 0.f;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < EMB_CH; ++j) {
      if (j < cnt) {
        acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; acc.w += g[j].w;
        const bool last = j + 1 == cnt || s_tok[j + 1] != s_tok[j];       // uniform over the workgroup
        if (last) {
          if (s_tok[j] != skip_token) {
            const int b0 = s_b0[j], b1 = s_b1[j];
            if (b0 >= start && b1 <= start + cnt) {                        // the whole bucket: this workgroup owns the table row
              float* o = dtable + (size_t)s_tok[j] * E + c * 4;
              float4 v = make_float4(acc.x * inv_keep, acc.y * inv_keep, acc.z * inv_keep, acc.w * inv_keep);
              if (accum) { const float4 q = *(const float4*)o; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
              *(float4*)o = v;
            } else {
              *(float4*)(part + (b0 >= start ? plane : 0) + (size_t)(chunk0 + blockIdx.x) * E + c * 4) = acc;
            }
          }
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
}
// the owner of a straddling bucket (the workgroup whose entries hold the bucket's first position) adds the partial rows and
// stores the table row.  A hot bucket spans hundreds of workgroups (the padding token: ~190 partial rows), so the sum is split
// over EMB_FG groups of 128 threads -- group g takes the partial rows k = g, g + EMB_FG, ... in that order, the groups' sums are
// added in group order through LDS: a fixed summation tree, whatever the timing (one thread walking all rows took 33 us).
constexpr int EMB_FG = 8;
__global__ __launch_bounds__(128 * EMB_FG) void embed_gather_finish_kernel(const int64_t* __restrict__ tokens, int ldtok, int N, int V1,
                                                                           const int* __restrict__ off, const int* __restrict__ perm, int total, int E,
                                                                           float inv_keep, long skip_token, float* __restrict__ dtable, int base, int keybase,
                                                                           int accum, const float* __restrict__ part, int chunk0, size_t plane) {
  __shared__ float4 s_sum[EMB_FG][128];
  const int start = base + blockIdx.x * EMB_CH;
  const int cnt = min(EMB_CH, total - start);
  const int pos = perm[start + cnt - 1];                      // the workgroup's last entry
  const int t = pos / N, n = pos - t * N;
  long tok = tokens[(size_t)n * ldtok + t];
  if (tok < 0 || tok >= V1) tok = 0;
  if (tok == skip_token) return;
  const int b0 = off[keybase + tok], b1 = off[keybase + tok + 1];
  if (b0 < start || b1 <= start + cnt) return;                // began earlier (somebody else's), or ends here (stored by the gather)
  const int c_last = (b1 - 1 - base) / EMB_CH;                // the workgroup that holds the bucket's last entry
  const int g = threadIdx.x >> 7, lt = threadIdx.x & 127;
  for (int c0 = 0; c0 < E / 4; c0 += 128) {
    const int c = c0 + lt;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < E / 4) {
      if (g == 0) acc = *(const float4*)(part + plane + (size_t)(chunk0 + blockIdx.x) * E + c * 4);   // the owner's own tail run
#pragma unroll 8
      for (int k = blockIdx.x + 1 + g; k <= c_last; k += EMB_FG) {
        const float4 q = *(const float4*)(part + (size_t)(chunk0 + k) * E + c * 4);
        acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
      }
    }
    s_sum[g][lt] = acc;
    __syncthreads();
    if (g == 0 && c < E / 4) {
#pragma unroll
      for (int j = 1; j < EMB_FG; ++j) { const float4 q = s_sum[j][lt]; acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
      float* o = dtable + (size_t)tok * E + c * 4;
      float4 v = make_float4(acc.x * inv_keep, acc.y * inv_keep, acc.z * inv_keep, acc.w * inv_keep);
      if (accum) { const float4 q = *(const float4*)o; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
      *(float4*)o = v;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ seq_per_img > 1: per-image features -> caption rows
// (the loader's S-fold replication, P/misc/dataloader/dataloader.py:270-277, done on the device instead of the host)
template <typename T>
__global__ void expand_rows_kernel(const float* __restrict__ src, T* __restrict__ dst, int S, size_t row, size_t total) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += stride) {   // i indexes src; row % 4 == 0
    const size_t img = i / row, e = i - img * row;
    const float4 v = *(const float4*)(src + i);
    for (int j = 0; j < S; ++j) {
      T* o = dst + (img * S + j) * row + e;
      if constexpr (sizeof(T) == 2) *(uint2*)o = make_uint2(uic_pack_bf16x2(v.x, v.y), uic_pack_bf16x2(v.z, v.w));
      else *(float4*)o = v;
    }
  }
}
template <typename T>
__global__ void expand_rows_scalar_kernel(const float* __restrict__ src, T* __restrict__ dst, int S, size_t row, size_t total) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t img = i / row, e = i - img * row;
    for (int j = 0; j < S; ++j) dst[(img * S + j) * row + e] = uic_from_f<T>(src[i]);
  }
}
template <typename T>
__global__ void expand_drop_kernel(const float* __restrict__ y, T* __restrict__ out, int S, size_t row, size_t total,
                                   float drop_p, unsigned seed, unsigned site) {
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += stride) {
    const size_t img = i / row, e = i - img * row;
    const float4 v = *(const float4*)(y + i);
    const float f[4] = {v.x, v.y, v.z, v.w};
    for (int j = 0; j < S; ++j) {
      const size_t o = (img * S + j) * row + e;
      float g[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        g[q] = drop_p > 0.f ? f[q] * uic_drop_scale(seed, site, (unsigned)(o + q), drop_p, inv_keep) : f[q];
      if constexpr (sizeof(T) == 2) *(uint2*)(out + o) = make_uint2(uic_pack_bf16x2(g[0], g[1]), uic_pack_bf16x2(g[2], g[3]));
      else *(float4*)(out + o) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
}
template <typename T>
__global__ void relu_mask_bwd_fold_kernel(const float* __restrict__ g, const T* __restrict__ act, float scale,
                                          T* __restrict__ dst, int S, size_t row, size_t total) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t img = i / row, e = i - img * row;
    float acc = 0.f;
    for (int j = 0; j < S; ++j) {
      const size_t o = (img * S + j) * row + e;
      acc += uic_to_f(act[o]) > 0.f ? g[o] * scale : 0.f;
    }
    dst[i] = uic_from_f<T>(acc);
  }
}

// ------------------------------------------------------------------ LSTM cell backward (pointwise part)
// Backward of nn.LSTMCell's gate math (P/models/AttModel.py:434,441): gates are stored activated.
template <typename T>
__global__ void lstm_bwd_kernel(const UicLstmBwdParams p) {
  const int H = p.H;
  const size_t total = (size_t)p.M * H;
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int m = (int)(idx / H), u = (int)(idx - (size_t)m * H);
    float dh = 0.f;
    if (p.dh0) {
      float v = p.dh0[(size_t)m * p.lddh0 + u];
      if (p.drop_p > 0.f) v *= uic_drop_scale(p.seed, p.site, (unsigned)idx, p.drop_p, inv_keep);
      dh += v;
    }
    if (p.dh1) dh += p.dh1[(size_t)m * p.lddh1 + u];
    if (p.dh2) dh += p.dh2[(size_t)m * p.lddh2 + u];
    for (int z = 0; z < p.nA; ++z) dh += p.slabA[(size_t)z * p.strideA + (size_t)m * p.ldA + u];
    for (int z = 0; z < p.nB; ++z) dh += p.slabB[(size_t)z * p.strideB + (size_t)m * p.ldB + u];
    const T* G = (const T*)p.gates + (size_t)m * 4 * H + u;
    const float gi = uic_to_f(G[0]), gf = uic_to_f(G[H]), gg = uic_to_f(G[2 * H]), go = uic_to_f(G[3 * H]);
    const float c = p.c[idx];
    const float cp = p.c_prev ? p.c_prev[idx] : 0.f;
    const float tc = uic_tanh<T>(c);                 // (the forward's own tanh: hardware exp on the bf16 path, libm on f32)
    const float dc = p.dc[idx] + dh * go * (1.f - tc * tc);
    const float d_o = dh * tc;
    T* D = (T*)p.dgates + (size_t)m * 4 * H + u;
    D[0] = uic_from_f<T>(dc * gg * gi * (1.f - gi));
    D[H] = uic_from_f<T>(dc * cp * gf * (1.f - gf));
    D[2 * H] = uic_from_f<T>(dc * gi * (1.f - gg * gg));
    D[3 * H] = uic_from_f<T>(d_o * go * (1.f - go));
    p.dc[idx] = dc * gf;
  }
}

// The same for H % 4 == 0 and 16-byte aligned rows: FOUR hidden units per lane, one 16-byte (f32) / 8-byte (bf16) access per
// tensor and gate instead of four scalar ones -- the kernel sits in the BPTT chain 34 times per step and is made of memory
// latency and instruction issue, not of bytes (captioner step 3.775 -> 3.757 ms over six alternations).  Same formulas per element.
// SIMPLE: dh0 and c_prev present, dh1 / dh2 absent -- the two calls of the split BPTT chain (topdown.hip); then no load has an
// optional address at all, only the slab stand-ins below.
template <typename T, bool SIMPLE>
__global__ __launch_bounds__(NT) void lstm_bwd_vec4_kernel(const UicLstmBwdParams p) {
  const int H = p.H, H4 = H >> 2;
  const int m = blockIdx.y, u4 = blockIdx.x * blockDim.x + threadIdx.x;      // (row from the grid: no division per lane)
  if (u4 >= H4) return;
  const int u = u4 * 4;
  const size_t idx = (size_t)m * H + u;
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  // Every load first, and NONE behind a run-time condition (round 6): an optional source that is absent is read from a valid
  // stand-in address (the d c row) and dropped by a select below.  hipcc waits for a predicated load at the join of its branch --
  // s_waitcnt vmcnt(0) right behind the first optional slab --, which made this link of the BPTT chain two to three serial memory
  // round trips instead of one.  The loads are raw buffer loads: a plain float4 load hipcc is free to split (it made one of them
  // a dwordx3 plus a dword under the dropout branch, with a full drain behind it) or to fold into another one of the same address.
  auto ld4 = [](const float* base, size_t elem) -> float4 {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
        __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000), (unsigned)(elem * 4), 0, 0));
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  const bool has0 = SIMPLE || p.dh0 != nullptr, has1 = !SIMPLE && p.dh1 != nullptr, has2 = !SIMPLE && p.dh2 != nullptr;
  const bool hascp = SIMPLE || p.c_prev != nullptr;
  const float4 c4 = ld4(p.c, idx);
  const float4 dc4 = ld4(p.dc, idx);
  const float4 d0r = has0 ? ld4(p.dh0, (size_t)m * p.lddh0 + u) : dc4;
  float4 d1r = make_float4(0.f, 0.f, 0.f, 0.f), d2r = d1r;
  if constexpr (!SIMPLE) {
    d1r = ld4(p.dh1 ? p.dh1 : p.dc, p.dh1 ? (size_t)m * p.lddh1 + u : idx);
    d2r = ld4(p.dh2 ? p.dh2 : p.dc, p.dh2 ? (size_t)m * p.lddh2 + u : idx);
  }
  const float4 cpr = ld4(hascp ? p.c_prev : p.c, idx);
  float4 sv[8];
  const int nA = p.nA < 4 ? p.nA : 4, nB = p.nB < 4 ? p.nB : 4;
  {
    const float* bA = nA > 0 ? p.slabA : p.dc;
    const float* bB = nB > 0 ? p.slabB : p.dc;
    const size_t oA = nA > 0 ? (size_t)m * p.ldA + u : idx, oB = nB > 0 ? (size_t)m * p.ldB + u : idx;
    const size_t sA = nA > 0 ? p.strideA : 0, sB = nB > 0 ? p.strideB : 0;
    const int mA = nA > 0 ? nA - 1 : 0, mB = nB > 0 ? nB - 1 : 0;
#pragma unroll
    for (int z = 0; z < 4; ++z) sv[z] = ld4(bA, (size_t)min(z, mA) * sA + oA);
#pragma unroll
    for (int z = 0; z < 4; ++z) sv[4 + z] = ld4(bB, (size_t)min(z, mB) * sB + oB);
  }
  float g[4][4];
  const T* G = (const T*)p.gates + (size_t)m * 4 * H + u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (sizeof(T) == 2) {
      const uint2 w = *(const uint2*)(G + q * H);
      g[q][0] = __uint_as_float(w.x << 16); g[q][1] = __uint_as_float(w.x & 0xffff0000u);
      g[q][2] = __uint_as_float(w.y << 16); g[q][3] = __uint_as_float(w.y & 0xffff0000u);
    } else {
      const float4 w = *(const float4*)(G + q * H);
      g[q][0] = w.x; g[q][1] = w.y; g[q][2] = w.z; g[q][3] = w.w;
    }
  }
  float4 ds = make_float4(0.f, 0.f, 0.f, 0.f);       // split-K partial slabs of two more sources, summed in a fixed order
#pragma unroll
  for (int z = 0; z < 8; ++z) {
    const bool use = z < 4 ? z < nA : z - 4 < nB;
    ds.x += use ? sv[z].x : 0.f; ds.y += use ? sv[z].y : 0.f; ds.z += use ? sv[z].z : 0.f; ds.w += use ? sv[z].w : 0.f;
  }
  const float4 cp = hascp ? cpr : make_float4(0.f, 0.f, 0.f, 0.f);
  const float dh0[4] = {d0r.x, d0r.y, d0r.z, d0r.w}, dh1[4] = {d1r.x, d1r.y, d1r.z, d1r.w}, dh2[4] = {d2r.x, d2r.y, d2r.z, d2r.w}, dsl[4] = {ds.x, ds.y, ds.z, ds.w};
  const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, cpv[4] = {cp.x, cp.y, cp.z, cp.w}, dcv[4] = {dc4.x, dc4.y, dc4.z, dc4.w};
  float o[4][4], dcn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float dh = 0.f;
    if (has0) {
      float v = dh0[k];
      if (p.drop_p > 0.f) v *= uic_drop_scale(p.seed, p.site, (unsigned)(idx + k), p.drop_p, inv_keep);
      dh += v;
    }
    if (has1) dh += dh1[k];
    if (has2) dh += dh2[k];
    dh += dsl[k];
    const float gi = g[0][k], gf = g[1][k], gg = g[2][k], go = g[3][k];
    const float tc = uic_tanh<T>(cc[k]);
    const float dc = dcv[k] + dh * go * (1.f - tc * tc);
    const float d_o = dh * tc;
    o[0][k] = dc * gg * gi * (1.f - gi);
    o[1][k] = dc * cpv[k] * gf * (1.f - gf);
    o[2][k] = dc * gi * (1.f - gg * gg);
    o[3][k] = d_o * go * (1.f - go);
    dcn[k] = dc * gf;
  }
  T* D = (T*)p.dgates + (size_t)m * 4 * H + u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (sizeof(T) == 2) *(uint2*)(D + q * H) = make_uint2(uic_pack_bf16x2(o[q][0], o[q][1]), uic_pack_bf16x2(o[q][2], o[q][3]));
    else *(float4*)(D + q * H) = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
  }
  *(float4*)(p.dc + idx) = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
}

// Backward of the maxout LSTMCore gate math (P/models/FCModel_NMT.py:32-50).  gates = (in, forget, out, g, first)
// as stored by the forward GEMM epilogue; the dropout mask applies to the SUM of the incoming dh because the dropped
// next_h is both the output and the recurrent state.
template <typename T>
__global__ void maxout_lstm_bwd_kernel(const UicLstmBwdParams p) {
  const int H = p.H;
  const size_t total = (size_t)p.M * H;
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int m = (int)(idx / H), u = (int)(idx - (size_t)m * H);
    float dh = 0.f;
    if (p.dh0) dh += p.dh0[(size_t)m * p.lddh0 + u];
    if (p.dh1) dh += p.dh1[(size_t)m * p.lddh1 + u];
    if (p.drop_p > 0.f) dh *= uic_drop_scale(p.seed, p.site, (unsigned)idx, p.drop_p, inv_keep);
    const T* G = (const T*)p.gates + (size_t)m * 5 * H + u;
    const float gi = uic_to_f(G[0]), gf = uic_to_f(G[H]), go = uic_to_f(G[2 * H]), gg = uic_to_f(G[3 * H]);
    const bool first = uic_to_f(G[4 * H]) > 0.5f;
    const float c = p.c[idx];
    const float cp = p.c_prev ? p.c_prev[idx] : 0.f;
    const float tc = tanhf(c);
    const float dc = p.dc[idx] + dh * go * (1.f - tc * tc);
    const float dg = dc * gi;
    T* D = (T*)p.dgates + (size_t)m * 5 * H + u;
    D[0] = uic_from_f<T>(dc * gg * gi * (1.f - gi));
    D[H] = uic_from_f<T>(dc * cp * gf * (1.f - gf));
    D[2 * H] = uic_from_f<T>(dh * tc * go * (1.f - go));
    D[3 * H] = uic_from_f<T>(first ? dg : 0.f);
    D[4 * H] = uic_from_f<T>(first ? 0.f : dg);
    p.dc[idx] = dc * gf;
  }
}

// FCModel_NMT._sample breaks BEFORE writing the step at which every row has finished (:203-206): zero that
// column and everything after it.
__global__ void sample_fixup_kernel(int N, int L, int ld, const int* n_unfinished, int64_t* seq, float* seq_logp) {
  int first = L;
  for (int t = 0; t < L; ++t) {
    int live = 0;
    for (int k = 0; k < UIC_NUNF_STRIPES; ++k) live += n_unfinished[t * UIC_NUNF_STRIPES + k];
    if (live == 0) { first = t; break; }
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * ld; i += gridDim.x * blockDim.x) {
    const int t = i % ld;
    if (t >= first) { seq[i] = 0; seq_logp[i] = 0.f; }
  }
}

// ------------------------------------------------------------------ log-softmax + LanguageModelCriterion
__device__ __forceinline__ float block_reduce_max(float v, float* s_buf) {
  v = uic_wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_buf[wave] = v;
  __syncthreads();
  float r = s_buf[0];
  for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, s_buf[i]);
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* s_buf) {
  v = uic_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_buf[wave] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < NT / 64; ++i) r += s_buf[i];
  return r;
}

// arg-max of a row (lowest index on ties) for the accuracy counters of NMT_loss.score; `src` may be LDS or global
__device__ __forceinline__ int block_argmax(const float* src, int V1, int* s_bi, float* s_bv) {
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = threadIdx.x; v < V1; v += NT) {
    const float x = src[v];
    if (x > bv || (x == bv && v < bi)) { bv = x; bi = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_bv[threadIdx.x >> 6] = bv; s_bi[threadIdx.x >> 6] = bi; }
  __syncthreads();
  bv = s_bv[0]; bi = s_bi[0];
  for (int w2 = 1; w2 < NT / 64; ++w2)
    if (s_bv[w2] > bv || (s_bv[w2] == bv && s_bi[w2] < bi)) { bv = s_bv[w2]; bi = s_bi[w2]; }
  return bi;
}

// One block per (t, n) row of logits: log_softmax (AttModel.py:163) fused with the masked NLL
// and its gradient (criterion.py:143-150): d logits = (softmax - onehot) * mask / sum(mask).
template <typename T>
__global__ __launch_bounds__(NT) void xe_kernel(const UicXeParams p, const float* __restrict__ logits, T* __restrict__ dlogits) {
  __shared__ float s_buf[NT / 64];
  const int m = blockIdx.x;
  const int mr = p.row_map ? p.row_map[m] : m;       // (row_map: the rows are a compacted list of (step, row) positions;
  const bool pad = p.row_map && (unsigned)mr >= (unsigned)p.row_map_limit;   //  -1 (or out of range) = a padding row: zero gradient, no loss entry)
  const int mo = pad ? 0 : mr;
  const int t = mo / p.N, n = mo - t * p.N;
  const float* row = logits + (size_t)m * p.ldv;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < p.V1; v += NT) mx = fmaxf(mx, row[v]);
  mx = block_reduce_max(mx, s_buf);
  float sum = 0.f;
  for (int v = threadIdx.x; v < p.V1; v += NT) sum += expf(row[v] - mx);
  sum = block_reduce_sum(sum, s_buf);
  const float lse = mx + logf(sum);
  long y = 0;
  float mk = 0.f;
  if (p.target) {
    y = pad ? 0 : p.target[(size_t)n * p.ldtarget + p.target_col0 + t];
    mk = p.mask && !pad ? p.mask[(size_t)n * p.ldmask + p.mask_col0 + t] : 0.f;
    if (p.score_stats) {
      __shared__ float s_bv[NT / 64];
      __shared__ int s_bi[NT / 64];
      const int am = block_argmax(row, p.V1, s_bi, s_bv);
      if (threadIdx.x == 0 && y != 0) {
        atomicAdd(&p.score_stats[1], 1);
        if (am == (int)y) atomicAdd(&p.score_stats[0], 1);
      }
    }
    if (y < 0 || y >= p.V1) y = 0;
    if (threadIdx.x == 0 && !pad) p.row_loss[m] = -(row[y] - lse) * mk;
  }
  if (p.logprobs) {
    float* lp = p.logprobs + (size_t)n * p.lp_row_stride + (size_t)t * p.lp_step_stride;
    for (int v = threadIdx.x; v < p.V1; v += NT) lp[v] = row[v] - lse;
  }
  if (p.write_grad) {
    const float sc = p.grad_scale ? p.grad_scale[(size_t)n * p.ldscale + p.scale_col0 + t] : mk * p.inv_den[0];
    T* d = dlogits + (size_t)m * p.ldv;
    for (int v = threadIdx.x; v < p.ldv; v += NT) {
      float g = 0.f;
      if (v < p.V1) g = (expf(row[v] - lse) - (v == y ? 1.f : 0.f)) * sc;
      d[v] = uic_from_f<T>(g);
    }
  }
}

// Rows too long for LDS (the 50 004-word NMT generator: 200 KB per row): TWO passes over the row instead of three --
// pass 1 keeps a running (max, sum of exp) per thread (rescaled when the max moves) together with the arg-max, pass 2
// writes the gradient -- with 16-byte loads.  The arg-max feeds the accuracy counters of NMT_loss.score
// (criterion.py:175-184), which otherwise cost a third pass of their own.
template <typename T>
__global__ __launch_bounds__(NT) void xe_big_kernel(const UicXeParams p, const float* __restrict__ logits, T* __restrict__ dlogits) {
  __shared__ float s_m[NT / 64], s_s[NT / 64], s_bv[NT / 64];
  __shared__ int s_bi[NT / 64];
  const int m = blockIdx.x;
  const int mr = p.row_map ? p.row_map[m] : m;       // (row_map: the rows are a compacted list of (step, row) positions;
  const bool pad = p.row_map && (unsigned)mr >= (unsigned)p.row_map_limit;   //  -1 (or out of range) = a padding row: zero gradient, no loss entry)
  const int mo = pad ? 0 : mr;
  const int t = mo / p.N, n = mo - t * p.N;
  const float* row = logits + (size_t)m * p.ldv;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -INFINITY, sum = 0.f, bv = -INFINITY;
  int bi = 0x7fffffff;
  auto take = [&](float x, int v) {
    if (x > bv || (x == bv && v < bi)) { bv = x; bi = v; }
    if (x > mx) { sum = sum * __expf(mx - x) + 1.f; mx = x; }
    else if (mx != -INFINITY) sum += __expf(x - mx);          // (x = mx = -inf contributes nothing)
  };
  for (int v = threadIdx.x * 4; v < p.V1; v += NT * 4) {
    const float4 x = *(const float4*)(row + v);            // ldv is a multiple of 4 and >= V1: in bounds
    take(x.x, v);
    if (v + 1 < p.V1) take(x.y, v + 1);
    if (v + 2 < p.V1) take(x.z, v + 2);
    if (v + 3 < p.V1) take(x.w, v + 3);
  }
  auto merge = [&](float om, float os, float ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    const float nm = fmaxf(mx, om);
    if (nm != -INFINITY) sum = sum * __expf(mx - nm) + os * __expf(om - nm);
    mx = nm;
  };
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    merge(__shfl_xor(mx, o, 64), __shfl_xor(sum, o, 64), __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
  if (lane == 0) { s_m[wave] = mx; s_s[wave] = sum; s_bv[wave] = bv; s_bi[wave] = bi; }
  __syncthreads();
  mx = s_m[0]; sum = s_s[0]; bv = s_bv[0]; bi = s_bi[0];
#pragma unroll
  for (int w2 = 1; w2 < NT / 64; ++w2) merge(s_m[w2], s_s[w2], s_bv[w2], s_bi[w2]);
  const float lse = mx + logf(sum);
  long y = 0;
  float mk = 0.f;
  if (p.target) {
    y = pad ? 0 : p.target[(size_t)n * p.ldtarget + p.target_col0 + t];
    mk = p.mask && !pad ? p.mask[(size_t)n * p.ldmask + p.mask_col0 + t] : 0.f;
    if (threadIdx.x == 0 && p.score_stats && y != 0) {
      atomicAdd(&p.score_stats[1], 1);
      if (bi == (int)y) atomicAdd(&p.score_stats[0], 1);
    }
    if (y < 0 || y >= p.V1) y = 0;
    if (threadIdx.x == 0 && !pad) p.row_loss[m] = -(row[y] - lse) * mk;
  }
  if (p.logprobs) {
    float* lp = p.logprobs + (size_t)n * p.lp_row_stride + (size_t)t * p.lp_step_stride;
    for (int v = threadIdx.x; v < p.V1; v += NT) lp[v] = row[v] - lse;
  }
  if (p.write_grad) {
    const float sc = p.grad_scale ? p.grad_scale[(size_t)n * p.ldscale + p.scale_col0 + t] : mk * p.inv_den[0];
    T* d = dlogits + (size_t)m * p.ldv;
    for (int v = threadIdx.x * 4; v < p.ldv; v += NT * 4) {
      const float4 x = *(const float4*)(row + v);
      const float xs[4] = {x.x, x.y, x.z, x.w};
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vv = v + j;
        g[j] = vv < p.V1 ? (__expf(xs[j] - lse) - (vv == y ? 1.f : 0.f)) * sc : 0.f;
      }
      if constexpr (sizeof(T) == 2) {
        *(uint2*)(d + v) = make_uint2(uic_pack_bf16x2(g[0], g[1]), uic_pack_bf16x2(g[2], g[3]));
      } else {
        *(float4*)(d + v) = make_float4(g[0], g[1], g[2], g[3]);
      }
    }
  }
}

// Same, with the logits row staged once in LDS (rows up to 64 KB): one HBM read of the 413 MB logits
// tensor instead of three.
template <typename T>
__global__ __launch_bounds__(NT) void xe_lds_kernel(const UicXeParams p, const float* __restrict__ logits, T* __restrict__ dlogits) {
  extern __shared__ __attribute__((aligned(16))) float s_row[];
  __shared__ float s_buf[NT / 64];
  const int m = blockIdx.x;
  const int mr = p.row_map ? p.row_map[m] : m;       // (row_map: the rows are a compacted list of (step, row) positions;
  const bool pad = p.row_map && (unsigned)mr >= (unsigned)p.row_map_limit;   //  -1 (or out of range) = a padding row: zero gradient, no loss entry)
  const int mo = pad ? 0 : mr;
  const int t = mo / p.N, n = mo - t * p.N;
  const float* row = logits + (size_t)m * p.ldv;
  float mx = -INFINITY;
  for (int v = threadIdx.x * 4; v < p.ldv; v += NT * 4) {
    const float4 x = *(const float4*)(row + v);
    *(float4*)(s_row + v) = x;
    if (v < p.V1) mx = fmaxf(mx, x.x);
    if (v + 1 < p.V1) mx = fmaxf(mx, x.y);
    if (v + 2 < p.V1) mx = fmaxf(mx, x.z);
    if (v + 3 < p.V1) mx = fmaxf(mx, x.w);
  }
  mx = block_reduce_max(mx, s_buf);
  float sum = 0.f;
  for (int v = threadIdx.x; v < p.V1; v += NT) sum += __expf(s_row[v] - mx);
  sum = block_reduce_sum(sum, s_buf);
  const float lse = mx + logf(sum);
  long y = 0;
  float mk = 0.f;
  if (p.target) {
    y = pad ? 0 : p.target[(size_t)n * p.ldtarget + p.target_col0 + t];
    mk = p.mask && !pad ? p.mask[(size_t)n * p.ldmask + p.mask_col0 + t] : 0.f;
    if (p.score_stats) {
      __shared__ float s_bv[NT / 64];
      __shared__ int s_bi[NT / 64];
      const int am = block_argmax(s_row, p.V1, s_bi, s_bv);
      if (threadIdx.x == 0 && y != 0) {
        atomicAdd(&p.score_stats[1], 1);
        if (am == (int)y) atomicAdd(&p.score_stats[0], 1);
      }
    }
    if (y < 0 || y >= p.V1) y = 0;
    if (threadIdx.x == 0 && !pad) p.row_loss[m] = -(s_row[y] - lse) * mk;
  }
  if (p.logprobs) {
    float* lp = p.logprobs + (size_t)n * p.lp_row_stride + (size_t)t * p.lp_step_stride;
    for (int v = threadIdx.x; v < p.V1; v += NT) lp[v] = s_row[v] - lse;
  }
  if (p.write_grad) {
    const float sc = p.grad_scale ? p.grad_scale[(size_t)n * p.ldscale + p.scale_col0 + t] : mk * p.inv_den[0];
    T* d = dlogits + (size_t)m * p.ldv;
    for (int v = threadIdx.x * 4; v < p.ldv; v += NT * 4) {
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vv = v + j;
        g[j] = vv < p.V1 ? (__expf(s_row[vv] - lse) - (vv == y ? 1.f : 0.f)) * sc : 0.f;
      }
      if constexpr (sizeof(T) == 2) {
        *(uint2*)(d + v) = make_uint2(uic_pack_bf16x2(g[0], g[1]), uic_pack_bf16x2(g[2], g[3]));
      } else {
        *(float4*)(d + v) = make_float4(g[0], g[1], g[2], g[3]);
      }
    }
  }
}

// Training-path variant (loss + d logits, no log-prob output): the row lives in REGISTERS (up to XE_RCH float4 per thread), one
// HBM read, one exp per element (the e^{x - max} of the normaliser pass is reused for the gradient), no LDS staging -- 8 instead
// of 4 workgroups per CU keep more loads in flight on the HBM-bound pass over the logits.
constexpr int XE_RCH = 10;     // rows up to 10 * NT * 4 = 10 240 columns
template <typename T>
__global__ __launch_bounds__(NT) void xe_reg_kernel(const UicXeParams p, const float* __restrict__ logits, T* __restrict__ dlogits) {
  __shared__ float s_buf[NT / 64];
  __shared__ float s_y;
  const int m = blockIdx.x;
  const int mr = p.row_map ? p.row_map[m] : m;       // (row_map: the rows are a compacted list of (step, row) positions;
  const bool pad = p.row_map && (unsigned)mr >= (unsigned)p.row_map_limit;   //  -1 (or out of range) = a padding row: zero gradient, no loss entry)
  const int mo = pad ? 0 : mr;
  const int t = mo / p.N, n = mo - t * p.N;
  const float* row = logits + (size_t)m * p.ldv;
  long y = pad ? 0 : p.target[(size_t)n * p.ldtarget + p.target_col0 + t];
  const float mk = p.mask && !pad ? p.mask[(size_t)n * p.ldmask + p.mask_col0 + t] : 0.f;
  if (y < 0 || y >= p.V1) y = 0;
  if (mk == 0.f && !p.grad_scale) {
    // a position behind its caption's end (a quarter of the benchmark's, a third of COCO's): loss 0 x (.), d logits = 0 x softmax - 0 --
    // exact zeros whatever the logits are, so the row is not read (round 6; uniform over the workgroup)
    if (threadIdx.x == 0 && !pad) p.row_loss[m] = 0.f;
    T* d = dlogits + (size_t)m * p.ldv;
    for (int v = threadIdx.x * 4; v < p.ldv; v += NT * 4) {
      if constexpr (sizeof(T) == 2) *(uint2*)(d + v) = make_uint2(0u, 0u);
      else *(float4*)(d + v) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  float4 x[XE_RCH];
  float mx = -INFINITY;
  // Every chunk of the row requested before the first is used, from a clamped address (round 6): behind `if (v < p.ldv)` hipcc
  // waited for each load at its branch's join -- a thread's ten loads were ten HBM round trips in a row.
#pragma unroll
  for (int i = 0; i < XE_RCH; ++i) {
    const int v = (threadIdx.x + i * NT) * 4;
    x[i] = *(const float4*)(row + (v < p.ldv ? v : p.ldv - 4));
  }
#pragma unroll
  for (int i = 0; i < XE_RCH; ++i) {
    const int v = (threadIdx.x + i * NT) * 4;
    const float4 r = x[i];
    const bool in = v < p.ldv;
    x[i].x = in && v < p.V1 ? r.x : -INFINITY; x[i].y = in && v + 1 < p.V1 ? r.y : -INFINITY;
    x[i].z = in && v + 2 < p.V1 ? r.z : -INFINITY; x[i].w = in && v + 3 < p.V1 ? r.w : -INFINITY;
    if (in && (long)v <= y && y < (long)v + 4) s_y = y == v ? r.x : y == v + 1 ? r.y : y == v + 2 ? r.z : r.w;
    mx = fmaxf(mx, fmaxf(fmaxf(x[i].x, x[i].y), fmaxf(x[i].z, x[i].w)));
  }
  mx = block_reduce_max(mx, s_buf);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < XE_RCH; ++i) {        // padded / out-of-row entries hold -inf -> e = 0
    x[i].x = __expf(x[i].x - mx); x[i].y = __expf(x[i].y - mx); x[i].z = __expf(x[i].z - mx); x[i].w = __expf(x[i].w - mx);
    sum += (x[i].x + x[i].y) + (x[i].z + x[i].w);
  }
  sum = block_reduce_sum(sum, s_buf);       // (its barriers also publish s_y)
  const float lse = mx + logf(sum);
  if (threadIdx.x == 0 && !pad) p.row_loss[m] = -(s_y - lse) * mk;
  const float sc = p.grad_scale ? p.grad_scale[(size_t)n * p.ldscale + p.scale_col0 + t] : mk * p.inv_den[0];
  const float k = sc / sum;
  T* d = dlogits + (size_t)m * p.ldv;
#pragma unroll
  for (int i = 0; i < XE_RCH; ++i) {
    const int v = (threadIdx.x + i * NT) * 4;
    if (v >= p.ldv) continue;
    float g[4] = {x[i].x * k, x[i].y * k, x[i].z * k, x[i].w * k};
    const long j = y - (long)v;           // (a dynamic index into g[] would put it in scratch)
    g[0] -= j == 0 ? sc : 0.f; g[1] -= j == 1 ? sc : 0.f; g[2] -= j == 2 ? sc : 0.f; g[3] -= j == 3 ? sc : 0.f;
    if constexpr (sizeof(T) == 2) {
      *(uint2*)(d + v) = make_uint2(uic_pack_bf16x2(g[0], g[1]), uic_pack_bf16x2(g[2], g[3]));
    } else {
      *(float4*)(d + v) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
}

// The same for rows too long for 256 threads' registers -- the pivot NMT's generator, 50 004 words = 200 KB per row: a workgroup
// of 1024 threads holds the row (13 float4 per thread), so the criterion reads the 397 MB of logits ONCE (xe_big_kernel: twice)
// and keeps the arg-max for the accuracy counters of NMT_loss.score (criterion.py:175-184) on the way.
constexpr int XE_WTH = 1024, XE_WCH = 13;     // rows up to 13 * 1024 * 4 = 53 248 columns (512 threads x 26 float4 at two
                                              // workgroups per CU: 128 registers per lane, 307 spilled -- 427 us)
template <typename T>
__global__ __launch_bounds__(XE_WTH) void xe_reg_wide_kernel(const UicXeParams p, const float* __restrict__ logits, T* __restrict__ dlogits) {
  __shared__ float s_f[XE_WTH / 64], s_bv[XE_WTH / 64];
  __shared__ int s_bi[XE_WTH / 64];
  __shared__ float s_y;
  const int m = blockIdx.x;
  const int mr = p.row_map ? p.row_map[m] : m;       // (row_map: the rows are a compacted list of (step, row) positions;
  const bool pad = p.row_map && (unsigned)mr >= (unsigned)p.row_map_limit;   //  -1 (or out of range) = a padding row: zero gradient, no loss entry)
  const int mo = pad ? 0 : mr;
  const int t = mo / p.N, n = mo - t * p.N;
  const float* row = logits + (size_t)m * p.ldv;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long y0 = pad ? 0 : p.target[(size_t)n * p.ldtarget + p.target_col0 + t];
  const float mk = p.mask && !pad ? p.mask[(size_t)n * p.ldmask + p.mask_col0 + t] : 0.f;
  const long y = y0 < 0 || y0 >= p.V1 ? 0 : y0;
  if (mk == 0.f && !p.grad_scale && !(p.score_stats && y0 != 0)) {      // (a padded target position: see xe_reg_kernel)
    if (threadIdx.x == 0 && !pad) p.row_loss[m] = 0.f;
    T* d = dlogits + (size_t)m * p.ldv;
    for (int v = threadIdx.x * 4; v < p.ldv; v += XE_WTH * 4) {
      if constexpr (sizeof(T) == 2) *(uint2*)(d + v) = make_uint2(0u, 0u);
      else *(float4*)(d + v) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  float4 x[XE_WCH];
  float mx = -INFINITY, bv = -INFINITY;
  int bi = 0x7fffffff;
  // (all thirteen chunks requested before the first is used, from clamped addresses: see xe_reg_kernel)
#pragma unroll
  for (int i = 0; i < XE_WCH; ++i) {
    const int v = (threadIdx.x + i * XE_WTH) * 4;
    x[i] = *(const float4*)(row + (v < p.ldv ? v : p.ldv - 4));
  }
#pragma unroll
  for (int i = 0; i < XE_WCH; ++i) {
    const int v = (threadIdx.x + i * XE_WTH) * 4;
    {
      const float4 r = x[i];
      const bool in = v < p.ldv;
      x[i].x = in && v < p.V1 ? r.x : -INFINITY; x[i].y = in && v + 1 < p.V1 ? r.y : -INFINITY;
      x[i].z = in && v + 2 < p.V1 ? r.z : -INFINITY; x[i].w = in && v + 3 < p.V1 ? r.w : -INFINITY;
      if (in && (long)v <= y && y < (long)v + 4) s_y = y == v ? r.x : y == v + 1 ? r.y : y == v + 2 ? r.z : r.w;
    }
    // (ascending index inside the thread: a later equal value does not replace the arg-max)
    if (x[i].x > bv) { bv = x[i].x; bi = v; }
    if (x[i].y > bv) { bv = x[i].y; bi = v + 1; }
    if (x[i].z > bv) { bv = x[i].z; bi = v + 2; }
    if (x[i].w > bv) { bv = x[i].w; bi = v + 3; }
  }
  mx = bv;
  // workgroup maximum and its lowest index
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_bv[wave] = bv; s_bi[wave] = bi; }
  __syncthreads();
  bv = s_bv[0]; bi = s_bi[0];
#pragma unroll
  for (int w2 = 1; w2 < XE_WTH / 64; ++w2)
    if (s_bv[w2] > bv || (s_bv[w2] == bv && s_bi[w2] < bi)) { bv = s_bv[w2]; bi = s_bi[w2]; }
  mx = bv;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < XE_WCH; ++i) {        // padded / out-of-row entries hold -inf -> e = 0
    x[i].x = __expf(x[i].x - mx); x[i].y = __expf(x[i].y - mx); x[i].z = __expf(x[i].z - mx); x[i].w = __expf(x[i].w - mx);
    sum += (x[i].x + x[i].y) + (x[i].z + x[i].w);
  }
  sum = uic_wave_sum(sum);
  if (lane == 0) s_f[wave] = sum;
  __syncthreads();                          // (also publishes s_y)
  sum = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < XE_WTH / 64; ++w2) sum += s_f[w2];
  const float lse = mx + logf(sum);
  if (threadIdx.x == 0) {
    p.row_loss[m] = -(s_y - lse) * mk;
    if (p.score_stats && y0 != 0) {
      atomicAdd(&p.score_stats[1], 1);
      if (bi == (int)y0) atomicAdd(&p.score_stats[0], 1);
    }
  }
  const float sc = p.grad_scale ? p.grad_scale[(size_t)n * p.ldscale + p.scale_col0 + t] : mk * p.inv_den[0];
  const float k = sc / sum;
  T* d = dlogits + (size_t)m * p.ldv;
#pragma unroll
  for (int i = 0; i < XE_WCH; ++i) {
    const int v = (threadIdx.x + i * XE_WTH) * 4;
    if (v >= p.ldv) continue;
    float g[4] = {x[i].x * k, x[i].y * k, x[i].z * k, x[i].w * k};
    const long j = y - (long)v;
    g[0] -= j == 0 ? sc : 0.f; g[1] -= j == 1 ? sc : 0.f; g[2] -= j == 2 ? sc : 0.f; g[3] -= j == 3 ? sc : 0.f;
    if constexpr (sizeof(T) == 2) {
      *(uint2*)(d + v) = make_uint2(uic_pack_bf16x2(g[0], g[1]), uic_pack_bf16x2(g[2], g[3]));
    } else {
      *(float4*)(d + v) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
}

// API-compat backward: upstream grad g wrt log-probs [n][t][v]; d logits = g - softmax * sum_v g
template <typename T>
__global__ __launch_bounds__(NT) void logsoftmax_bwd_kernel(T* __restrict__ dlogits, int V1, int ldv, int N, const float* __restrict__ g,
                                                            size_t g_step, size_t g_row, const float* __restrict__ logprobs) {
  __shared__ float s_buf[NT / 64];
  const int m = blockIdx.x;
  const int t = m / N, n = m - t * N;
  const float* gr = g + (size_t)n * g_row + (size_t)t * g_step;
  const float* lp = logprobs + (size_t)n * g_row + (size_t)t * g_step;
  float sum = 0.f;
  for (int v = threadIdx.x; v < V1; v += NT) sum += gr[v];
  sum = block_reduce_sum(sum, s_buf);
  T* d = dlogits + (size_t)m * ldv;
  for (int v = threadIdx.x; v < ldv; v += NT) {
    float x = 0.f;
    if (v < V1) x = gr[v] - expf(lp[v]) * sum;
    d[v] = uic_from_f<T>(x);
  }
}

constexpr int MS_NT = 1024, MS_U = 16;
__global__ __launch_bounds__(MS_NT) void masked_sum_kernel(const float* __restrict__ mask, int ldmask, int col0, int N, int TS,
                                                           float* out_sum, float* out_inv) {
  __shared__ float s_buf[MS_NT / 64];
  // One workgroup (the sum is one number and must not depend on a grid), but ONE round trip: 1024 threads x 16 unconditional
  // loads from clamped addresses cover the reference's 640 x 17 mask in a single pass (round 6; 256 threads x 8 loads at a time
  // took six dependent passes -- 12-19 us at the head of the step's side stream)
  float s = 0.f;
  const int total = N * TS;
  for (int i0 = threadIdx.x; i0 < total; i0 += MS_NT * MS_U) {
    float v[MS_U];
#pragma unroll
    for (int u = 0; u < MS_U; ++u) {
      int i = i0 + u * MS_NT;
      i = i < total ? i : total - 1;
      const int n = i / TS, t = i - n * TS;
      v[u] = mask[(size_t)n * ldmask + col0 + t];
    }
#pragma unroll
    for (int u = 0; u < MS_U; ++u) s += i0 + u * MS_NT < total ? v[u] : 0.f;
  }
  s = uic_wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < MS_NT / 64; ++k) tot += s_buf[k];
    if (out_sum) out_sum[0] = tot;
    if (out_inv) out_inv[0] = 1.f / tot;
  }
}

__global__ __launch_bounds__(NT) void reduce_sum_kernel(const float* __restrict__ x, size_t n, const float* scale, float* out) {
  __shared__ float s_buf[NT / 64];
  float s = 0.f;
  for (size_t i0 = threadIdx.x; i0 < n; i0 += (size_t)NT * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + (size_t)u * NT < n ? x[i0 + (size_t)u * NT] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  s = block_reduce_sum(s, s_buf);
  if (threadIdx.x == 0) out[0] = scale ? s * scale[0] : s;
}

struct UicZero4 { uint4* p[4]; size_t n[4]; };     // n in 16-byte units
__global__ void zero4_kernel(const UicZero4 z) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < z.n[k]; i += stride) z.p[k][i] = zero;
}

// up to UIC_ZERO_LIST buffers in ONE launch (blockIdx.y picks the buffer): a step of the pivot NMT cleared 18 small buffers with
// 18 memsets of ~6 us each on its only stream
struct UicZeroList { uint4* p[UIC_ZERO_LIST]; size_t n[UIC_ZERO_LIST]; };     // n in 16-byte units
__global__ void zero_list_kernel(const UicZeroList z) {
  uint4* p = nullptr;
  size_t n = 0;
#pragma unroll
  for (int k = 0; k < UIC_ZERO_LIST; ++k)
    if (k == (int)blockIdx.y) { p = z.p[k]; n = z.n[k]; }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = zero;
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam, P/misc/optimizer.py:70)
// (16 bytes per lane and four independent partial sums in flight: the scalar form read the pivot NMT's 360 MB of gradients at
// 2.7 TB/s, one dependent add per load)
__global__ __launch_bounds__(NT) void sqnorm_part_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
  __shared__ float s_buf[NT / 64];
  const size_t n4 = ((uintptr_t)g & 15) == 0 ? n / 4 : 0;
  const size_t stride = (size_t)gridDim.x * NT;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  const f32x4* g4 = (const f32x4*)g;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 a = __builtin_nontemporal_load(g4 + i), b = __builtin_nontemporal_load(g4 + i + stride);
    const f32x4 c = __builtin_nontemporal_load(g4 + i + 2 * stride), d = __builtin_nontemporal_load(g4 + i + 3 * stride);
    s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
    s1 += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
    s2 += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
    s3 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  }
  for (; i < n4; i += stride) {
    const f32x4 a = g4[i];
    s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
  }
  for (size_t j = n4 * 4 + (size_t)blockIdx.x * NT + threadIdx.x; j < n; j += stride) s1 += g[j] * g[j];
  float s = block_reduce_sum((s0 + s1) + (s2 + s3), s_buf);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// ONE definition of the element update for both Adam kernels, with floating-point contraction off: the whole-arena kernel and the
// ranges kernel of the sharded exchange must produce the same bits from the same inputs (tests/test_gpu_dp2.py compares the two
// exchanges bit for bit), whatever fused multiply-adds the optimizer would pick for each loop shape.
__device__ __forceinline__ float uic_adam_update(float p, float& m, float& v, float g, float beta1, float beta2, float step_size,
                                                 float inv_sqrt_bc2, float eps) {
#pragma clang fp contract(off)
  m = beta1 * m + (1.f - beta1) * g;
  v = beta2 * v + (1.f - beta2) * g * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  return p - step_size * (m / denom);
}
__device__ __forceinline__ float uic_adam_grad_scale(const UicAdamParams& a) {
#pragma clang fp contract(off)
  float gs = a.grad_scale;
  if (a.sqnorm) {
    const float coef = a.max_norm / (fabsf(a.grad_scale) * sqrtf(a.sqnorm[0]) + 1e-6f);
    if (coef < 1.f) gs *= coef;
  }
  return gs;
}
__global__ void adam_kernel(const UicAdamParams a) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float step_size = a.lr / a.bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(a.bc2);
  if (a.guard && a.guard[0] != 0) return;          // (uniform over the launch: the step's gradients are invalid, keep the weights)
  const float gs = uic_adam_grad_scale(a);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    // gradient and moments are touched once per step, by this kernel only: streamed past the caches (the parameters are
    // re-read right after by the operand-copy refresh and stay cacheable)
    const float g = __builtin_nontemporal_load(a.g + i) * gs;
    float m = __builtin_nontemporal_load(a.m + i), v = __builtin_nontemporal_load(a.v + i);
    const float pn = uic_adam_update(a.p[i], m, v, g, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps);
    __builtin_nontemporal_store(m, a.m + i);
    __builtin_nontemporal_store(v, a.v + i);
    a.p[i] = pn;
  }
}

// The same update on up to UIC_ADAM_RANGES index ranges of the arena in ONE launch -- the ranges a data-parallel rank owns after
// the reduce-scatter (one per gradient piece) plus the replicated tail -- optionally leaving the updated parameters of those
// ranges in the operand dtype in `w_out` (same element index), which is what the all-gather then distributes.
template <typename WT>
__global__ void adam_ranges_kernel(const UicAdamParams a, const UicAdamRanges r, WT* __restrict__ w_out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float step_size = a.lr / a.bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(a.bc2);
  if (a.guard && a.guard[0] != 0) return;
  const float gs = uic_adam_grad_scale(a);
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < r.total; j += stride) {
    int k = 0;
    while (k + 1 < r.count && j >= r.start[k + 1]) ++k;       // start[k]: position of range k in the concatenation
    const size_t i = r.lo[k] + (j - r.start[k]);
    const float g = __builtin_nontemporal_load(a.g + i) * gs;
    float m = __builtin_nontemporal_load(a.m + i), v = __builtin_nontemporal_load(a.v + i);
    const float pn = uic_adam_update(a.p[i], m, v, g, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps);
    __builtin_nontemporal_store(m, a.m + i);
    __builtin_nontemporal_store(v, a.v + i);
    a.p[i] = pn;
    if (w_out) w_out[i] = uic_from_f<WT>(pn);
  }
}

// ------------------------------------------------------------------ scheduled sampling (AttModel.py:130-143)
// One block per caption row: rows whose uniform falls under ss_prob take an inverse-CDF draw from
// softmax(previous step's logits) (= exp(outputs[:, i-1]), torch.multinomial's distribution); the others keep seq[:, i].
__global__ __launch_bounds__(NT) void ss_sample_kernel(const float* __restrict__ logits, int V1, int ldv,
                                                       const int64_t* __restrict__ labels, int ld_labels, int t, float ss_prob,
                                                       unsigned seed, int64_t* __restrict__ used, int ld_used) {
  __shared__ float s_buf[NT / 64];
  __shared__ float s_val[NT];
  const int n = blockIdx.x;
  if (!(uic_uniform(seed, UIC_SITE_SS_MASK0 + (unsigned)t, (unsigned)n) < ss_prob)) {
    if (threadIdx.x == 0) used[(size_t)n * ld_used + t] = labels[(size_t)n * ld_labels + t];
    return;
  }
  const float* row = logits + (size_t)n * ldv;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V1; v += NT) mx = fmaxf(mx, row[v]);
  mx = block_reduce_max(mx, s_buf);
  const int per = (V1 + NT - 1) / NT;
  const int v_lo = threadIdx.x * per, v_hi = min(V1, v_lo + per);
  float part = 0.f;
  for (int v = v_lo; v < v_hi; ++v) part += expf(row[v] - mx);
  s_val[threadIdx.x] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < NT; ++i) tot += s_val[i];
    const float target = uic_uniform(seed, UIC_SITE_SS_DRAW0 + (unsigned)t, (unsigned)n) * tot;
    float cum = 0.f;
    int seg = NT - 1;
    for (int i = 0; i < NT; ++i) {
      if (cum + s_val[i] > target) { seg = i; break; }
      cum += s_val[i];
    }
    int pick = -1;
    const int lo = seg * per, hi = min(V1, lo + per);
    for (int v = lo; v < hi; ++v) {
      const float pr = expf(row[v] - mx);
      if (pr > 0.f) pick = v;
      cum += pr;
      if (cum > target && pr > 0.f) break;
    }
    if (pick < 0) pick = 0;
    used[(size_t)n * ld_used + t] = pick;
  }
}
__global__ void copy_tokens_kernel(const int64_t* __restrict__ src, int ld_src, int N, int T, int64_t* __restrict__ dst, int ld_dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * T) return;
  const int n = i / T, t = i - n * T;
  dst[(size_t)n * ld_dst + t] = src[(size_t)n * ld_src + t];
}

// ------------------------------------------------------------------ one decode step of AttModel._sample
// (P/models/AttModel.py:216-251): log_softmax, optional decoding constraint, greedy max (lowest index on
// ties) or multinomial draw, finished-row bookkeeping.  The reference's host-side early break
// (`unfinished.sum() == 0`) becomes a device counter per step, so no host sync is needed.
// STAGED: the logits row is copied into LDS once (16-byte loads) and every pass below reads it from there; rows too long
// for 64 KB of LDS are read from memory by each pass.  Same arithmetic in the same order either way.
// FAST (bf16 models): hardware exp; the f32 parity path keeps libm's
template <bool STAGED, bool FAST>
__global__ __launch_bounds__(NT) void sample_step_kernel(const UicSampleParams p) {
  auto ex = [](float x) { return FAST ? __expf(x) : expf(x); };
  __shared__ float s_buf[NT / 64];
  __shared__ float s_val[NT];
  __shared__ int s_idx[NT];
  extern __shared__ __attribute__((aligned(16))) float s_row[];
  const int n = blockIdx.x;
  const int t = p.t;
  const float* grow = (const float*)p.logits + (size_t)n * p.ldv;
  const float* row = grow;
  if constexpr (STAGED) {
    const int n4 = p.V1 >> 2;
    for (int i = threadIdx.x; i < n4; i += NT) *(float4*)(s_row + 4 * i) = *(const float4*)(grow + 4 * i);
    for (int v = (n4 << 2) + threadIdx.x; v < p.V1; v += NT) s_row[v] = grow[v];
    __syncthreads();
    row = s_row;
  }
  bool dead = false;                                                       // every row had finished: the reference broke out
  if (!p.fc_mode && t > 0) {
    int live = threadIdx.x < UIC_NUNF_STRIPES ? p.n_unfinished[(t - 1) * UIC_NUNF_STRIPES + threadIdx.x] : 0;
    dead = __syncthreads_or(live) == 0;
  }
  long banned = -1;
  const int ldo = p.ld_out > 0 ? p.ld_out : p.L;
  if (p.decoding_constraint && t > 0) banned = p.seq[(size_t)n * ldo + t - 1];

  float mx = -INFINITY;
  for (int v = threadIdx.x; v < p.V1; v += NT) mx = fmaxf(mx, row[v]);
  mx = block_reduce_max(mx, s_buf);
  float sum = 0.f;
  for (int v = threadIdx.x; v < p.V1; v += NT) sum += ex(row[v] - mx);
  sum = block_reduce_sum(sum, s_buf);
  const float lse = mx + logf(sum);
  if (p.logprobs_out)
    for (int v = threadIdx.x; v < p.V1; v += NT) p.logprobs_out[(size_t)n * p.V1 + v] = row[v] - lse;

  int choice = 0;
  if (p.sample_max) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < p.V1; v += NT) {
      const float x = (v == banned) ? -INFINITY : row[v];
      if (x > bv || (x == bv && v < bi)) { bv = x; bi = v; }
    }
    // wave-level arg-max by shuffles, then one LDS exchange between the four waves (value descending, lowest index on ties)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = bv; s_idx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    bv = s_val[0]; bi = s_idx[0];
#pragma unroll
    for (int w2 = 1; w2 < NT / 64; ++w2)
      if (s_val[w2] > bv || (s_val[w2] == bv && s_idx[w2] < bi)) { bv = s_val[w2]; bi = s_idx[w2]; }
    choice = bi;
  } else if (p.forced) {
    choice = (int)p.forced[(size_t)n * p.L + t];
  } else {
    // inverse-CDF draw from softmax(logprobs / temperature) with the banned token removed.  Thread i owns the contiguous
    // vocabulary segment [i * per, (i + 1) * per); an inclusive scan of the segment masses (wave shuffles + one LDS exchange
    // between the waves) finds the segment the uniform number falls into, and only that segment's owner walks its <= per
    // entries.  (The first version summed and searched the NT segment masses on thread 0: two serial loops of NT LDS reads,
    // ~14 us of the kernel's 32.)
    const float invT = 1.f / p.temperature;
    float part = 0.f;
    const int per = (p.V1 + NT - 1) / NT;
    const int v_lo = threadIdx.x * per, v_hi = min(p.V1, v_lo + per);
    for (int v = v_lo; v < v_hi; ++v) part += (v == banned) ? 0.f : ex((row[v] - lse) * invT);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float incl = part;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    __syncthreads();                                   // s_buf was last read by block_reduce_sum
    if (lane == 63) s_buf[wv] = incl;
    __syncthreads();
    float woff = 0.f, tot = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NT / 64; ++w2) {
      if (w2 < wv) woff += s_buf[w2];
      tot += s_buf[w2];
    }
    incl += woff;
    s_val[threadIdx.x] = incl;
    if (threadIdx.x == 0) s_idx[0] = -1;
    __syncthreads();
    unsigned x = (unsigned)n * 0x9E3779B1u ^ (p.seed + (unsigned)t * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float target = (float)(x >> 8) * (1.0f / 16777216.0f) * tot;
    const float below = threadIdx.x > 0 ? s_val[threadIdx.x - 1] : 0.f;
    // the owner: the first segment whose inclusive mass exceeds the target (the last one if rounding leaves none)
    const bool owner = (incl > target && !(below > target)) || (threadIdx.x == NT - 1 && !(incl > target));
    if (owner) {
      float cum = below;
      int pick = -1;
      for (int v = v_lo; v < v_hi; ++v) {
        const float pr = (v == banned) ? 0.f : ex((row[v] - lse) * invT);
        if (pr > 0.f) pick = v;
        cum += pr;
        if (cum > target && pr > 0.f) break;
      }
      s_idx[0] = pick;
    }
    __syncthreads();
    if (s_idx[0] < 0) {
      // rounding left the target at or beyond the total mass and the last segment is empty: the last token that has mass (rare)
      if (threadIdx.x == 0) {
        int pick = 0;
        for (int v = p.V1 - 1; v >= 0; --v)
          if (v != banned && ex((row[v] - lse) * invT) > 0.f) { pick = v; break; }
        s_idx[0] = pick;
      }
      __syncthreads();
    }
    choice = s_idx[0];
  }
  if (threadIdx.x == 0) {
    if (dead) {
      p.seq[(size_t)n * ldo + t] = 0;
      p.seq_logp[(size_t)n * ldo + t] = 0.f;
      p.it[n] = 0;
    } else {
      const float lp = row[choice] - lse;
      int unf = choice > 0;
      if (t > 0) unf = unf && p.unfinished[n];
      p.unfinished[n] = unf;
      const long tok = unf ? choice : 0;
      p.it[n] = p.fc_mode ? (long)choice : tok;      // FCModel_NMT feeds the RAW sampled token to the next step (:199)
      p.seq[(size_t)n * ldo + t] = tok;
      p.seq_logp[(size_t)n * ldo + t] = lp;
      if (unf) atomicAdd(&p.n_unfinished[t * UIC_NUNF_STRIPES + (n % UIC_NUNF_STRIPES)], 1);
    }
  }
  if (p.xt_out) {   // the next step's input embedding of this row (embed_fwd_kernel's arithmetic)
    __syncthreads();                                   // (s_idx[0] was read by everybody above)
    if (threadIdx.x == 0) s_idx[0] = (int)p.it[n];
    __syncthreads();
    long tok = s_idx[0];
    if (tok < 0 || tok >= p.embed_V1) tok = 0;
    const int E = p.embed_E;
    const float inv_keep = p.embed_drop_p > 0.f ? 1.f / (1.f - p.embed_drop_p) : 1.f;
    for (int c = threadIdx.x * 4; c < E; c += NT * 4) {
      float f[4];
      if (p.embed_table_dtype == UIC_BF16) uic_table_load4((const bf16_t*)p.embed_table + (size_t)tok * E + c, f);
      else uic_table_load4((const float*)p.embed_table + (size_t)tok * E + c, f);
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = fmaxf(f[j], 0.f);
      if (p.embed_drop_p > 0.f) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          f[j] *= uic_drop_scale(p.seed, p.embed_site, (unsigned)(p.embed_idx_base + (size_t)n * E + c + j), p.embed_drop_p, inv_keep);
      }
      if (p.dtype == UIC_BF16) {
        bf16_t* o = (bf16_t*)p.xt_out + (size_t)n * E + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)f[j];
      } else {
        *(float4*)((float*)p.xt_out + (size_t)n * E + c) = make_float4(f[0], f[1], f[2], f[3]);
      }
    }
  }
}

__global__ void dropout_mask_kernel(float* out, size_t n, float pdrop, unsigned seed, unsigned site, size_t base) {
  const float inv_keep = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = pdrop > 0.f ? uic_drop_scale(seed, site, (unsigned)(base + i), pdrop, inv_keep) : 1.f;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
  do {                                         \
    if ((dtype) == UIC_BF16) { CALL_BF16; } else { CALL_F32; } \
  } while (0)

int uic_cast_f32_launch(int dtype, const float* src, void* dst, size_t n, hipStream_t s) {
  if (n == 0) return UIC_OK;
  const int g = grid_for(n, NT * 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, src, (bf16_t*)dst, n),
             hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(g), dim3(NT), 0, s, src, (float*)dst, n));
  UIC_LAUNCH_CHECK("cast_from_f32");
  return UIC_OK;
}
// up to UIC_CAST_MULTI tensors in ONE launch (blockIdx.y picks the tensor): the weight refresh of a training step is a handful
// of small casts whose launches, not their bytes, are what the step waits for
namespace {
struct CastMulti { const float* src[UIC_CAST_MULTI]; void* dst[UIC_CAST_MULTI]; size_t n[UIC_CAST_MULTI]; };
template <typename T>
__global__ void cast_multi_kernel(const CastMulti c) {
  const int k = blockIdx.y;
  const float* src = k == 0 ? c.src[0] : k == 1 ? c.src[1] : k == 2 ? c.src[2] : k == 3 ? c.src[3] : k == 4 ? c.src[4] : c.src[5];
  T* dst = (T*)(k == 0 ? c.dst[0] : k == 1 ? c.dst[1] : k == 2 ? c.dst[2] : k == 3 ? c.dst[3] : k == 4 ? c.dst[4] : c.dst[5]);
  const size_t n = k == 0 ? c.n[0] : k == 1 ? c.n[1] : k == 2 ? c.n[2] : k == 3 ? c.n[3] : k == 4 ? c.n[4] : c.n[5];
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = make_float4(__builtin_nontemporal_load(src + i), __builtin_nontemporal_load(src + i + 1),
                                   __builtin_nontemporal_load(src + i + 2), __builtin_nontemporal_load(src + i + 3));
      if constexpr (sizeof(T) == 2) *(uint2*)(dst + i) = make_uint2(uic_pack_bf16x2(v.x, v.y), uic_pack_bf16x2(v.z, v.w));
      else *(float4*)(dst + i) = v;
    } else {
      for (size_t j = i; j < n; ++j) dst[j] = uic_from_f<T>(src[j]);
    }
  }
}
}  // namespace
namespace {
typedef __attribute__((ext_vector_type(4))) unsigned u32x4m;
struct CopyMulti { const u32x4m* src[UIC_CAST_MULTI]; u32x4m* dst[UIC_CAST_MULTI]; size_t n16[UIC_CAST_MULTI]; };
__global__ __launch_bounds__(NT) void copy_multi_kernel(const CopyMulti c) {
  const int k = blockIdx.y;
  const u32x4m* src = c.src[k];
  u32x4m* dst = c.dst[k];
  const size_t n = c.n16[k], stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}
}  // namespace
// up to UIC_CAST_MULTI device-to-device copies in ONE launch (sizes in bytes, multiples of 16; 16-byte aligned pointers)
int uic_copy_multi_launch(int count, const void* const* src, void* const* dst, const size_t* bytes, hipStream_t s) {
  UIC_REQUIRE(count >= 0 && count <= UIC_CAST_MULTI, "copy_multi: %d tensors (max %d)", count, UIC_CAST_MULTI);
  CopyMulti c;
  memset(&c, 0, sizeof(c));
  size_t most = 0;
  int m = 0;
  for (int i = 0; i < count; ++i) {
    if (bytes[i] == 0) continue;
    UIC_REQUIRE(src[i] && dst[i] && (((uintptr_t)src[i] | (uintptr_t)dst[i] | bytes[i]) & 15) == 0, "copy_multi: tensor %d must be 16-byte aligned", i);
    c.src[m] = (const u32x4m*)src[i]; c.dst[m] = (u32x4m*)dst[i]; c.n16[m] = bytes[i] / 16;
    if (c.n16[m] > most) most = c.n16[m];
    ++m;
  }
  if (m == 0) return UIC_OK;
  hipLaunchKernelGGL(copy_multi_kernel, dim3(grid_for(most, NT * 2), m), dim3(NT), 0, s, c);
  UIC_LAUNCH_CHECK("copy_multi");
  return UIC_OK;
}
int uic_cast_f32_multi_launch(int dtype, int count, const float* const* src, void* const* dst, const size_t* n, hipStream_t s) {
  UIC_REQUIRE(count >= 0 && count <= UIC_CAST_MULTI, "cast_multi: %d tensors (max %d)", count, UIC_CAST_MULTI);
  CastMulti c;
  memset(&c, 0, sizeof(c));
  size_t most = 0;
  int m = 0;
  for (int i = 0; i < count; ++i) {
    if (n[i] == 0) continue;
    UIC_REQUIRE(src[i] && dst[i] && (((uintptr_t)src[i] | (uintptr_t)dst[i]) & 15) == 0, "cast_multi: tensor %d must be 16-byte aligned", i);
    c.src[m] = src[i]; c.dst[m] = dst[i]; c.n[m] = n[i];
    if (n[i] > most) most = n[i];
    ++m;
  }
  if (m == 0) return UIC_OK;
  const int g = grid_for(most, NT * 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_multi_kernel<bf16_t>, dim3(g, m), dim3(NT), 0, s, c),
             hipLaunchKernelGGL(cast_multi_kernel<float>, dim3(g, m), dim3(NT), 0, s, c));
  UIC_LAUNCH_CHECK("cast_multi");
  return UIC_OK;
}
int uic_to_f32_launch(int dtype, const void* src, float* dst, size_t n, hipStream_t s) {
  if (n == 0) return UIC_OK;
  const int g = grid_for(n, NT);
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_to_f32_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, (const bf16_t*)src, dst, n),
             hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(g), dim3(NT), 0, s, (const float*)src, dst, n));
  UIC_LAUNCH_CHECK("cast_to_f32");
  return UIC_OK;
}
namespace {
// device-to-device copy as an ordinary kernel: hipMemcpyAsync(DeviceToDevice) holds the calling host thread until the stream
// has drained up to the copy (measured: the other stream's launches behind it were enqueued ~0.2 ms late in the fused step)
__global__ void copy_words_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
}  // namespace
int uic_copy_launch(void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return UIC_OK;
  UIC_REQUIRE(bytes % 4 == 0 && ((uintptr_t)dst & 3) == 0 && ((uintptr_t)src & 3) == 0, "copy: %zu bytes / pointers must be 4-byte aligned", bytes);
  const size_t n = bytes / 4;
  const int g = grid_for(n, NT);
  hipLaunchKernelGGL(copy_words_kernel, dim3(g), dim3(NT), 0, s, (const unsigned*)src, (unsigned*)dst, n);
  UIC_LAUNCH_CHECK("copy_words");
  return UIC_OK;
}
namespace {
__global__ __launch_bounds__(256) void exp2x2_bf16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n8) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float f[8];
    uic_unpack<bf16_t>(src[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      f[j] = __builtin_amdgcn_exp2f(fminf(fmaxf(f[j] * 2.8853900817779268f, -UIC_E2_CLAMP), UIC_E2_CLAMP));
    dst[i] = uic_pack<bf16_t>(f);
  }
}
}  // namespace
int uic_exp2x2_launch(const void* src, void* dst, size_t n, hipStream_t s) {
  UIC_REQUIRE(n % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "exp2x2: n %% 8 and 16-byte alignment");
  if (n == 0) return UIC_OK;
  hipLaunchKernelGGL(exp2x2_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, n / 8);
  UIC_LAUNCH_CHECK("exp2x2");
  return UIC_OK;
}
int uic_fill_launch(void* dst, int value_byte, size_t bytes, hipStream_t s) {
  if (bytes == 0) return UIC_OK;
  return uic_check_hip(hipMemsetAsync(dst, value_byte, bytes, s), "hipMemsetAsync");
}
namespace {
template <typename T>
__global__ void fill_value_kernel(T* __restrict__ dst, size_t n, float value) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = uic_from_f<T>(value);
}
}  // namespace
int uic_fill_value_launch(int dtype, void* dst, size_t n, float value, hipStream_t s) {
  if (n == 0) return UIC_OK;
  const int g = grid_for(n, NT);
  DISPATCH_T(dtype, hipLaunchKernelGGL(fill_value_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, (bf16_t*)dst, n, value),
             hipLaunchKernelGGL(fill_value_kernel<float>, dim3(g), dim3(NT), 0, s, (float*)dst, n, value));
  UIC_LAUNCH_CHECK("fill_value");
  return UIC_OK;
}
int uic_transpose_launch(int dtype, const void* src, int rows, int cols, int lds, void* dst, int ldd, hipStream_t s) {
  UIC_REQUIRE(ldd >= rows && lds >= cols, "transpose: ldd=%d < rows=%d or lds=%d < cols=%d", ldd, rows, lds, cols);
  if (rows == 0 || cols == 0) return UIC_OK;
  dim3 grid((ldd + 63) / 64, (cols + 63) / 64);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(NT), 0, s, (const bf16_t*)src, rows, cols, lds, (bf16_t*)dst, ldd),
             hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(NT), 0, s, (const float*)src, rows, cols, lds, (float*)dst, ldd));
  UIC_LAUNCH_CHECK("transpose");
  return UIC_OK;
}
int uic_transpose_multi_launch(int dtype, int count, const UicTransposeJob* jobs, hipStream_t s) {
  UIC_REQUIRE(count >= 0 && count <= UIC_TRANSPOSE_MULTI, "transpose_multi: %d jobs (max %d)", count, UIC_TRANSPOSE_MULTI);
  TransposeMulti c;
  memset(&c, 0, sizeof(c));
  int m = 0, tiles = 0;
  for (int i = 0; i < count; ++i) {
    if (jobs[i].rows == 0 || jobs[i].cols == 0) continue;
    c.start[m] = tiles;
    c.j[m++] = jobs[i];
    tiles += ((jobs[i].ldd + 63) / 64) * ((jobs[i].cols + 63) / 64);
  }
  if (m == 0) return UIC_OK;
  for (int i = m; i <= UIC_TRANSPOSE_MULTI; ++i) c.start[i] = 0x7fffffff;     // (never reached by a block index)
  DISPATCH_T(dtype, hipLaunchKernelGGL(transpose_multi_kernel<bf16_t>, dim3(tiles), dim3(NT), 0, s, c),
             hipLaunchKernelGGL(transpose_multi_kernel<float>, dim3(tiles), dim3(NT), 0, s, c));
  UIC_LAUNCH_CHECK("transpose_multi");
  return UIC_OK;
}
int uic_colsum_launch(int src_dtype, const void* src, int rows, int cols, int lds, float* out, float* scratch,
                      size_t scratch_floats, hipStream_t s) {
  if (cols == 0) return UIC_OK;
  int nrb = (rows + 63) / 64;
  if (nrb > 32) nrb = 32;
  if (nrb < 1) nrb = 1;
  while (nrb > 1 && (size_t)nrb * cols > scratch_floats) nrb /= 2;
  UIC_REQUIRE((size_t)nrb * cols <= scratch_floats, "colsum: scratch too small (%zu floats for %d cols)", scratch_floats, cols);
  const int rpb = (rows + nrb - 1) / nrb;
  dim3 grid((cols + 63) / 64, nrb);
  DISPATCH_T(src_dtype,
             hipLaunchKernelGGL(colsum_stage1<bf16_t>, grid, dim3(NT), 0, s, (const bf16_t*)src, rows, cols, lds, rpb, scratch),
             hipLaunchKernelGGL(colsum_stage1<float>, grid, dim3(NT), 0, s, (const float*)src, rows, cols, lds, rpb, scratch));
  UIC_LAUNCH_CHECK("colsum_stage1");
  hipLaunchKernelGGL(colsum_stage2, dim3((cols + 63) / 64), dim3(NT), 0, s, scratch, nrb, cols, out);
  UIC_LAUNCH_CHECK("colsum_stage2");
  return UIC_OK;
}
int uic_sum_steps_launch(int dtype, const void* src, int T, size_t step_elems, void* dst, hipStream_t s) {
  if (step_elems == 0) return UIC_OK;
  const size_t vec = dtype == UIC_BF16 ? 8 : 4;
  if (step_elems % vec == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const int gv = grid_for(step_elems / vec, NT);
    DISPATCH_T(dtype, hipLaunchKernelGGL(sum_steps_vec_kernel<bf16_t>, dim3(gv), dim3(NT), 0, s, (const bf16_t*)src, T, step_elems, (bf16_t*)dst),
               hipLaunchKernelGGL(sum_steps_vec_kernel<float>, dim3(gv), dim3(NT), 0, s, (const float*)src, T, step_elems, (float*)dst));
    UIC_LAUNCH_CHECK("sum_steps_vec");
    return UIC_OK;
  }
  const int g = grid_for(step_elems, NT);
  DISPATCH_T(dtype, hipLaunchKernelGGL(sum_steps_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, (const bf16_t*)src, T, step_elems, (bf16_t*)dst),
             hipLaunchKernelGGL(sum_steps_kernel<float>, dim3(g), dim3(NT), 0, s, (const float*)src, T, step_elems, (float*)dst));
  UIC_LAUNCH_CHECK("sum_steps");
  return UIC_OK;
}
int uic_embed_fwd_t_launch(int dtype, const void* table, int table_dtype, int V1, int E, const int64_t* tokens, int ldtok, int N, int T,
                           float drop_p, unsigned seed, unsigned site, size_t idx_base, int relu, void* out, hipStream_t s) {
  UIC_REQUIRE(E % 4 == 0, "embed: E=%d must be a multiple of 4", E);
  UIC_REQUIRE(table_dtype == UIC_F32 || (table_dtype == UIC_BF16 && dtype == UIC_BF16), "embed: a bf16 table needs bf16 outputs");
  if (N == 0 || T == 0) return UIC_OK;
  const bool t16 = table_dtype == UIC_BF16;
  if (dtype == UIC_BF16 && E % 8 == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)table & 15) == 0 && (size_t)T * N < ((size_t)1 << 30)) {
    const int rows = T * N;
    int gw = (rows + 3) / 4;
    if (gw > 8192) gw = 8192;
    if (t16) hipLaunchKernelGGL(embed_fwd_rows_bf16_kernel<bf16_t>, dim3(gw), dim3(256), 0, s, (const bf16_t*)table, V1, E, tokens, ldtok, N, rows, drop_p, seed, site, idx_base, relu, (bf16_t*)out);
    else hipLaunchKernelGGL(embed_fwd_rows_bf16_kernel<float>, dim3(gw), dim3(256), 0, s, (const float*)table, V1, E, tokens, ldtok, N, rows, drop_p, seed, site, idx_base, relu, (bf16_t*)out);
    UIC_LAUNCH_CHECK("embed_fwd_rows");
    return UIC_OK;
  }
  const int g = grid_for((size_t)T * N * (E / 4), NT);
  if (t16) {
    hipLaunchKernelGGL((embed_fwd_kernel<bf16_t, bf16_t>), dim3(g), dim3(NT), 0, s, (const bf16_t*)table, V1, E, tokens, ldtok, N, T, drop_p, seed, site, idx_base, relu, (bf16_t*)out);
  } else {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((embed_fwd_kernel<bf16_t, float>), dim3(g), dim3(NT), 0, s, (const float*)table, V1, E, tokens, ldtok, N, T, drop_p, seed, site, idx_base, relu, (bf16_t*)out),
               hipLaunchKernelGGL((embed_fwd_kernel<float, float>), dim3(g), dim3(NT), 0, s, (const float*)table, V1, E, tokens, ldtok, N, T, drop_p, seed, site, idx_base, relu, (float*)out));
  }
  UIC_LAUNCH_CHECK("embed_fwd");
  return UIC_OK;
}
int uic_embed_fwd_launch(int dtype, const float* table, int V1, int E, const int64_t* tokens, int ldtok, int N, int T,
                         float drop_p, unsigned seed, unsigned site, size_t idx_base, int relu, void* out, hipStream_t s) {
  return uic_embed_fwd_t_launch(dtype, table, UIC_F32, V1, E, tokens, ldtok, N, T, drop_p, seed, site, idx_base, relu, out, s);
}
// scratch layout (ints): cnt [nkeys + 1] | off [nkeys + 1] | (unused) [nkeys + 1] | perm [N T] ... then, at offsets that do not
// depend on `split`: cntb [nblk, 2 V1] | part [2, nchunks, E] f32
namespace {
inline size_t emb_fixed_ints(int N, int T, int V1) { return ((3 * ((size_t)2 * V1 + 1) + (size_t)N * T + 64) + 63) & ~(size_t)63; }
inline int emb_blocks(int total) { return (total + EMB_BLK - 1) / EMB_BLK; }
inline size_t emb_cntb_ints(int N, int T, int V1) { return ((size_t)emb_blocks(N * T) * 2 * V1 + 63) & ~(size_t)63; }
inline size_t emb_chunks(int N, int T) { return (size_t)(N * T) / EMB_CH + 2; }           // two halves, each rounded up
}  // namespace
size_t uic_embed_bwd_sorted_scratch_ints(int N, int T, int V1, int E) {
  return emb_fixed_ints(N, T, V1) + emb_cntb_ints(N, T, V1) + 2 * emb_chunks(N, T) * (size_t)E;
}
// Two halves, so that a caller can do the token bucketing (which needs only the tokens) long before the gradients exist:
//   prepare: zero dtable, histogram -> prefixes -> fill of the position list into `scratch`;   gather: the sums.
// split > 0 (< T): the list holds the positions of decode steps [0, split) first, then those of [split, T); gather then takes one
// of the two halves per call (half = 0 / 1) and adds into the table, so the later steps' share can be gathered before the
// earlier steps' d xt exists.  Both calls must get the same `split` (and the same N, T, V1).
int uic_embed_bwd_sorted_prepare(const int64_t* tokens, int ldtok, int N, int T, int V1, int E, float* dtable, int* scratch, hipStream_t s, int split) {
  UIC_REQUIRE(E % 4 == 0 && scratch, "embed_bwd_sorted: E=%d must be a multiple of 4", E);
  UIC_REQUIRE(split >= 0 && split < (T > 0 ? T : 1), "embed_bwd_sorted: split=%d outside [0, %d)", split, T);
  if (V1 == 0) return UIC_OK;
  const int nkeys = (split > 0 ? 2 : 1) * V1;
  int* cnt = scratch;
  int* off = cnt + (nkeys + 1);
  int* perm = scratch + 3 * (nkeys + 1);
  int* cntb = scratch + emb_fixed_ints(N, T, V1);
  const int total = N * T;
  UIC_TRY(uic_fill_launch(dtable, 0, (size_t)V1 * E * 4, s));
  if (total == 0) return UIC_OK;
  const int nblk = emb_blocks(total);
  UIC_TRY(uic_fill_launch(cntb, 0, (size_t)nblk * nkeys * 4, s));
  hipLaunchKernelGGL(embed_hist_kernel, dim3(nblk), dim3(EMB_BLK), 0, s, tokens, ldtok, N, T, V1, split, nkeys, cntb);
  UIC_LAUNCH_CHECK("embed_hist");
  hipLaunchKernelGGL(embed_block_prefix_kernel, dim3((nkeys + NT - 1) / NT), dim3(NT), 0, s, cntb, nkeys, nblk, cnt);
  UIC_LAUNCH_CHECK("embed_block_prefix");
  hipLaunchKernelGGL(embed_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)cnt, nkeys, off);
  UIC_LAUNCH_CHECK("embed_scan");
  hipLaunchKernelGGL(embed_fill_kernel, dim3(nblk), dim3(EMB_BLK), 0, s, tokens, ldtok, N, T, V1, split, nkeys, (const int*)off, (const int*)cntb, perm);
  UIC_LAUNCH_CHECK("embed_fill");
  return UIC_OK;
}
int uic_embed_bwd_sorted_gather(int dtype, const float* dxt, const void* xt, const int64_t* tokens, int ldtok, int N, int T,
                                int V1, int E, float drop_p, long skip_token, float* dtable, int* scratch, hipStream_t s,
                                int split, int half) {
  if (V1 == 0 || N * T == 0) return UIC_OK;
  int base = 0, total = N * T, keybase = 0, nkeys = V1, chunk0 = 0;
  if (split > 0) {
    UIC_REQUIRE(split < T && (half == 0 || half == 1), "embed_bwd_sorted_gather: split=%d half=%d (T=%d)", split, half, T);
    nkeys = 2 * V1;
    base = half ? split * N : 0; total = half ? T * N : split * N; keybase = half ? V1 : 0;
    chunk0 = half ? (split * N + EMB_CH - 1) / EMB_CH : 0;
  }
  const int* off = scratch + (nkeys + 1);
  const int* perm = scratch + 3 * (nkeys + 1);
  float* part = (float*)(scratch + emb_fixed_ints(N, T, V1) + emb_cntb_ints(N, T, V1));
  const size_t plane = emb_chunks(N, T) * (size_t)E;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int gw = (total - base + EMB_CH - 1) / EMB_CH;
  const int accum = split > 0;     // (the table was zeroed by prepare; every bucket has one owner per launch)
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(embed_gather_kernel<bf16_t>, dim3(gw), dim3(128), 0, s, dxt, (const bf16_t*)xt, tokens, ldtok, N, V1, off, perm, total, E, inv_keep, skip_token, dtable, base, keybase, accum, part, chunk0, plane),
             hipLaunchKernelGGL(embed_gather_kernel<float>, dim3(gw), dim3(128), 0, s, dxt, (const float*)xt, tokens, ldtok, N, V1, off, perm, total, E, inv_keep, skip_token, dtable, base, keybase, accum, part, chunk0, plane));
  UIC_LAUNCH_CHECK("embed_gather");
  hipLaunchKernelGGL(embed_gather_finish_kernel, dim3(gw), dim3(128 * EMB_FG), 0, s, tokens, ldtok, N, V1, off, perm, total, E, inv_keep, skip_token, dtable, base, keybase, accum, (const float*)part, chunk0, plane);
  UIC_LAUNCH_CHECK("embed_gather_finish");
  return UIC_OK;
}
// dtable [V1, E] is overwritten.  scratch: uic_embed_bwd_sorted_scratch_ints ints.
int uic_embed_bwd_sorted_launch(int dtype, const float* dxt, const void* xt, const int64_t* tokens, int ldtok, int N, int T,
                                int V1, int E, float drop_p, long skip_token, float* dtable, int* scratch, hipStream_t s) {
  UIC_TRY(uic_embed_bwd_sorted_prepare(tokens, ldtok, N, T, V1, E, dtable, scratch, s, 0));
  return uic_embed_bwd_sorted_gather(dtype, dxt, xt, tokens, ldtok, N, T, V1, E, drop_p, skip_token, dtable, scratch, s, 0, 0);
}
// out[c] = sum_n part[n, c] for the ncols <= 1024 columns of a small [rows, ncols] f32 matrix, split over two destinations
// (columns [0, n0) -> out0, the rest -> out1): d w_alpha / d b_alpha from the attention accumulation's per-row partials in ONE
// launch instead of a two-stage column sum and two copies.
namespace {
__global__ __launch_bounds__(1024) void colsum_small_kernel(const float* __restrict__ part, int rows, int ncols, int n0, float* __restrict__ out0,
                                                            float* __restrict__ out1) {
  __shared__ float s_acc[16][64];
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;          // 64 columns x 16 row groups per workgroup
  const int c = blockIdx.x * 64 + lc;
  float acc = 0.f;
  if (c < ncols) {
#pragma unroll 8
    for (int r = g; r < rows; r += 16) acc += part[(size_t)r * ncols + c];
  }
  s_acc[g][lc] = acc;
  __syncthreads();
  if (g == 0 && c < ncols) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += s_acc[k][lc];
    if (c < n0) out0[c] = v; else out1[c - n0] = v;
  }
}
}  // namespace
int uic_colsum_small_launch(const float* part, int rows, int ncols, int n0, float* out0, float* out1, hipStream_t s) {
  if (ncols == 0) return UIC_OK;
  hipLaunchKernelGGL(colsum_small_kernel, dim3((ncols + 63) / 64), dim3(1024), 0, s, part, rows, ncols, n0, out0, out1);
  UIC_LAUNCH_CHECK("colsum_small");
  return UIC_OK;
}
int uic_relu_mask_bwd_launch(int dtype, const float* grad, const void* act, float scale, void* dst, size_t n, hipStream_t s) {
  if (n == 0) return UIC_OK;
  const int g = grid_for(n, NT);
  DISPATCH_T(dtype, hipLaunchKernelGGL(relu_mask_bwd_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, grad, (const bf16_t*)act, scale, (bf16_t*)dst, n),
             hipLaunchKernelGGL(relu_mask_bwd_kernel<float>, dim3(g), dim3(NT), 0, s, grad, (const float*)act, scale, (float*)dst, n));
  UIC_LAUNCH_CHECK("relu_mask_bwd");
  return UIC_OK;
}
int uic_expand_rows_launch(int dtype_out, const float* src, void* dst, int n_img, int S, size_t row, hipStream_t s) {
  UIC_REQUIRE(src && dst && S >= 1, "expand_rows: bad arguments");
  const size_t total = (size_t)n_img * row;
  if (total == 0) return UIC_OK;
  if (row % 4 != 0) {     // short odd rows (att_masks [n_img, R])
    const int g1 = grid_for(total, NT);
    DISPATCH_T(dtype_out, hipLaunchKernelGGL(expand_rows_scalar_kernel<bf16_t>, dim3(g1), dim3(NT), 0, s, src, (bf16_t*)dst, S, row, total),
               hipLaunchKernelGGL(expand_rows_scalar_kernel<float>, dim3(g1), dim3(NT), 0, s, src, (float*)dst, S, row, total));
    UIC_LAUNCH_CHECK("expand_rows_scalar");
    return UIC_OK;
  }
  const int g = grid_for(total / 4, NT);
  DISPATCH_T(dtype_out, hipLaunchKernelGGL(expand_rows_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, src, (bf16_t*)dst, S, row, total),
             hipLaunchKernelGGL(expand_rows_kernel<float>, dim3(g), dim3(NT), 0, s, src, (float*)dst, S, row, total));
  UIC_LAUNCH_CHECK("expand_rows");
  return UIC_OK;
}
int uic_expand_drop_launch(int dtype, const float* y, void* out, int n_img, int S, size_t row, float drop_p, unsigned seed,
                           unsigned site, hipStream_t s) {
  UIC_REQUIRE(y && out && S >= 1 && row % 4 == 0, "expand_drop: bad arguments");
  const size_t total = (size_t)n_img * row;
  if (total == 0) return UIC_OK;
  const int g = grid_for(total / 4, NT);
  DISPATCH_T(dtype, hipLaunchKernelGGL(expand_drop_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, y, (bf16_t*)out, S, row, total, drop_p, seed, site),
             hipLaunchKernelGGL(expand_drop_kernel<float>, dim3(g), dim3(NT), 0, s, y, (float*)out, S, row, total, drop_p, seed, site));
  UIC_LAUNCH_CHECK("expand_drop");
  return UIC_OK;
}
int uic_relu_mask_bwd_fold_launch(int dtype, const float* grad, const void* act, float scale, void* dst, int n_img, int S,
                                  size_t row, hipStream_t s) {
  UIC_REQUIRE(grad && act && dst && S >= 1, "relu_mask_bwd_fold: bad arguments");
  const size_t total = (size_t)n_img * row;
  if (total == 0) return UIC_OK;
  const int g = grid_for(total, NT);
  DISPATCH_T(dtype, hipLaunchKernelGGL(relu_mask_bwd_fold_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, grad, (const bf16_t*)act, scale, (bf16_t*)dst, S, row, total),
             hipLaunchKernelGGL(relu_mask_bwd_fold_kernel<float>, dim3(g), dim3(NT), 0, s, grad, (const float*)act, scale, (float*)dst, S, row, total));
  UIC_LAUNCH_CHECK("relu_mask_bwd_fold");
  return UIC_OK;
}
int uic_lstm_bwd_launch(const UicLstmBwdParams& p, hipStream_t s) {
  UIC_REQUIRE(p.dc && p.gates && p.c && p.dgates, "lstm_bwd: null pointer");
  UIC_REQUIRE((!p.nA || p.slabA) && (!p.nB || p.slabB) && p.nA >= 0 && p.nB >= 0, "lstm_bwd: slab sources");
  if (p.M == 0) return UIC_OK;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  // (small problems -- the pivot NMT's 64 rows -- keep one unit per lane: four times the workgroups, measured faster there)
  const bool vec = (size_t)p.M * p.H >= 65536 && p.H % 4 == 0 && al16(p.dc) && al16(p.c) && (!p.c_prev || al16(p.c_prev)) &&
                   ((uintptr_t)p.gates & 7) == 0 && ((uintptr_t)p.dgates & 7) == 0 && (p.dtype == UIC_BF16 || (al16(p.gates) && al16(p.dgates))) &&
                   (!p.dh0 || (al16(p.dh0) && p.lddh0 % 4 == 0)) && (!p.dh1 || (al16(p.dh1) && p.lddh1 % 4 == 0)) &&
                   (!p.dh2 || (al16(p.dh2) && p.lddh2 % 4 == 0)) && (size_t)p.M * p.H < ((size_t)1 << 32) && p.M <= 65535 &&
                   p.nA <= 4 && p.nB <= 4 && (!p.nA || (al16(p.slabA) && p.ldA % 4 == 0 && p.strideA % 4 == 0)) &&
                   (!p.nB || (al16(p.slabB) && p.ldB % 4 == 0 && p.strideB % 4 == 0));
  if (vec) {
    const int h4 = p.H / 4;
    const int bt = h4 >= NT ? NT : ((h4 + 63) / 64) * 64;
    const dim3 g4((unsigned)((h4 + bt - 1) / bt), (unsigned)p.M);
    const bool simple = p.dh0 && !p.dh1 && !p.dh2 && p.c_prev;
    if (simple) {
      DISPATCH_T(p.dtype, hipLaunchKernelGGL((lstm_bwd_vec4_kernel<bf16_t, true>), g4, dim3(bt), 0, s, p),
                 hipLaunchKernelGGL((lstm_bwd_vec4_kernel<float, true>), g4, dim3(bt), 0, s, p));
    } else {
      DISPATCH_T(p.dtype, hipLaunchKernelGGL((lstm_bwd_vec4_kernel<bf16_t, false>), g4, dim3(bt), 0, s, p),
                 hipLaunchKernelGGL((lstm_bwd_vec4_kernel<float, false>), g4, dim3(bt), 0, s, p));
    }
    UIC_LAUNCH_CHECK("lstm_bwd_vec4");
    return UIC_OK;
  }
  const int g = grid_for((size_t)p.M * p.H, NT);
  DISPATCH_T(p.dtype, hipLaunchKernelGGL(lstm_bwd_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, p),
             hipLaunchKernelGGL(lstm_bwd_kernel<float>, dim3(g), dim3(NT), 0, s, p));
  UIC_LAUNCH_CHECK("lstm_bwd");
  return UIC_OK;
}
namespace {
// out[m, :] = src[map[m], :] (rows of `chunks` 16-byte pieces) for m < M; rows [M, Mpad) of out are cleared
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int* __restrict__ map, uint4* __restrict__ out, int M, int Mpad,
                                                          int chunks, int limit) {
  const size_t total = (size_t)Mpad * chunks, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int m = (int)(i / chunks), c = (int)(i - (size_t)m * chunks);
    const int r = m < M ? map[m] : -1;
    out[i] = (unsigned)r < (unsigned)limit ? src[(size_t)r * chunks + c] : make_uint4(0u, 0u, 0u, 0u);
  }
}
// dst[map[m], :] = src[m, :] for m < M
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint4* __restrict__ src, const int* __restrict__ map, uint4* __restrict__ dst, int M, int chunks, int limit) {
  const size_t total = (size_t)M * chunks, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int m = (int)(i / chunks), c = (int)(i - (size_t)m * chunks);
    const int r = map[m];
    if ((unsigned)r < (unsigned)limit) dst[(size_t)r * chunks + c] = src[i];
  }
}
// row_len[n] = 1 + the last t < T with mask[n * ld + col0 + t] != 0 (0: none): a thread per row walks its T consecutive words (the
// list kernels have just read them: cache hits) -- no atomics
__device__ inline void live_row_len(const float* __restrict__ mask, int ld, int col0, int N, int T, int* __restrict__ row_len, int tid) {
  for (int n = tid; n < N; n += 1024) {
    const float* m = mask + (size_t)n * ld + col0;
    int len = 0;
    for (int t = 0; t < T; ++t) len = m[t] != 0.f ? t + 1 : len;
    row_len[n] = len;
  }
}
// out = the positions p = t * N + n, ascending, whose mask[n * ld + col0 + t] is not zero; entries behind them up to out_len: -1.
// ONE workgroup: an ordered compaction of ~1e4 positions in chunks of 1024 (ballot + wave prefix + one running offset), a few
// microseconds beside the prologue -- the list never crosses PCIe.
// inv (optional, [M]): the inverse -- inv[p] = the list index of position p, or -1; zero16 (optional): n16 16-byte words cleared on
// the way (the padding rows of the compact operand the list's consumer fills by itself, rnn_persist.hip)
__global__ __launch_bounds__(1024) void live_list_kernel(const float* __restrict__ mask, int ld, int col0, int N, int M, int* __restrict__ out, int out_len,
                                                         int* __restrict__ inv, uint4* __restrict__ zero16, int n16, int* __restrict__ row_len) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int p0 = 0; p0 < M; p0 += 1024) {
    const int p = p0 + tid;
    bool on = false;
    if (p < M) {
      const int t = p / N, n = p - t * N;
      on = mask[(size_t)n * ld + col0 + t] != 0.f;
    }
    const unsigned long long b = __ballot(on);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int off = s_base;
    for (int w2 = 0; w2 < wave; ++w2) off += s_wave[w2];
    if (on && off + before < out_len) out[off + before] = p;
    if (inv && p < M) inv[p] = on && off + before < out_len ? off + before : -1;
    __syncthreads();
    if (tid == 0) { int tot = 0; for (int w2 = 0; w2 < 16; ++w2) tot += s_wave[w2]; s_base += tot; }
    __syncthreads();
  }
  for (int i = s_base + tid; i < out_len; i += 1024) out[i] = -1;
  for (int i = tid; i < n16; i += 1024) zero16[i] = make_uint4(0u, 0u, 0u, 0u);
  if (row_len) live_row_len(mask, ld, col0, N, M / N, row_len, tid);
}
// the same for M <= 16 x 1024 positions (the captioner's 10 880, the NMT step's ~2 000) with ONE round trip to memory: every
// thread requests its <= 16 mask words at once, the per-(round, wave) counts meet in LDS, one wave scans the 256 of them
// (19 -> 4 us: the first form paid a memory latency and three barriers per 1024 positions)
constexpr int LL_R = 16;
__global__ __launch_bounds__(1024) void live_list_burst_kernel(const float* __restrict__ mask, int ld, int col0, int N, int M, int* __restrict__ out, int out_len,
                                                               int* __restrict__ inv, uint4* __restrict__ zero16, int n16, int* __restrict__ row_len) {
  __shared__ int s_cnt[LL_R * 16];
  __shared__ int s_off[LL_R * 16 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float v[LL_R];
#pragma unroll
  for (int i = 0; i < LL_R; ++i) {
    int p = i * 1024 + tid;
    p = p < M ? p : 0;                                  // (a valid address; the test below drops it)
    const int t = p / N, n = p - t * N;
    v[i] = mask[(size_t)n * ld + col0 + t];
  }
  unsigned bits = 0;
#pragma unroll
  for (int i = 0; i < LL_R; ++i) {
    const bool on = i * 1024 + tid < M && v[i] != 0.f;
    bits |= on ? 1u << i : 0u;
    const unsigned long long b = __ballot(on);
    if (lane == 0) s_cnt[i * 16 + wave] = __popcll(b);
  }
  __syncthreads();
  if (wave == 0) {                                      // exclusive prefix of the 256 counts: 4 per lane + a wave scan
    int c[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = s_cnt[lane * 4 + j]; sum += c[j]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    int run = incl - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_off[lane * 4 + j] = run; run += c[j]; }
    if (lane == 63) s_off[LL_R * 16] = run;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < LL_R; ++i) {
    const bool on = (bits >> i) & 1u;
    const unsigned long long b = __ballot(on);
    const int at = s_off[i * 16 + wave] + __popcll(b & ((1ull << lane) - 1ull));
    if (on && at < out_len) out[at] = i * 1024 + tid;
    if (inv && i * 1024 + tid < M) inv[i * 1024 + tid] = on && at < out_len ? at : -1;
  }
  for (int i = s_off[LL_R * 16] + tid; i < out_len; i += 1024) out[i] = -1;
  for (int i = tid; i < n16; i += 1024) zero16[i] = make_uint4(0u, 0u, 0u, 0u);
  if (row_len) live_row_len(mask, ld, col0, N, M / N, row_len, tid);
}
}  // namespace
int uic_live_list_launch(const float* mask, int ld, int col0, int N, int M, int* out, int out_len, hipStream_t s, int* inv, void* zero, size_t zero_bytes,
                         int* row_len) {
  UIC_REQUIRE(mask && out && N > 0 && M >= 0 && out_len >= 0, "live_list: bad arguments");
  UIC_REQUIRE(zero_bytes == 0 || (zero && ((uintptr_t)zero & 15) == 0 && zero_bytes % 16 == 0 && zero_bytes < ((size_t)1 << 30)), "live_list: the cleared region must be 16-byte aligned / sized");
  if (out_len == 0 && !inv && !row_len) return UIC_OK;
  const int n16 = (int)(zero_bytes / 16);
  if (M <= LL_R * 1024 && M > 0) hipLaunchKernelGGL(live_list_burst_kernel, dim3(1), dim3(1024), 0, s, mask, ld, col0, N, M, out, out_len, inv, (uint4*)zero, n16, row_len);
  else hipLaunchKernelGGL(live_list_kernel, dim3(1), dim3(1024), 0, s, mask, ld, col0, N, M, out, out_len, inv, (uint4*)zero, n16, row_len);
  UIC_LAUNCH_CHECK("live_list");
  return UIC_OK;
}
int uic_gather_rows_launch(const void* src, const int* map, int src_rows, void* out, int M, int Mpad, size_t row_bytes, hipStream_t s) {
  UIC_REQUIRE(row_bytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)out & 15) == 0 && Mpad >= M && M >= 0, "gather_rows: bad arguments");
  if (Mpad == 0) return UIC_OK;
  const int chunks = (int)(row_bytes / 16);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)Mpad * chunks, 256)), dim3(256), 0, s, (const uint4*)src, map, (uint4*)out, M, Mpad, chunks, src_rows);
  UIC_LAUNCH_CHECK("gather_rows");
  return UIC_OK;
}
int uic_scatter_rows_launch(const void* src, const int* map, void* dst, int dst_rows, int M, size_t row_bytes, hipStream_t s) {
  UIC_REQUIRE(row_bytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0 && M >= 0, "scatter_rows: bad arguments");
  if (M == 0) return UIC_OK;
  const int chunks = (int)(row_bytes / 16);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((size_t)M * chunks, 256)), dim3(256), 0, s, (const uint4*)src, map, (uint4*)dst, M, chunks, dst_rows);
  UIC_LAUNCH_CHECK("scatter_rows");
  return UIC_OK;
}

int uic_xe_launch(const UicXeParams& p, hipStream_t s) {
  UIC_REQUIRE(p.logits && p.N > 0, "xe: null logits or N=0");
  UIC_REQUIRE(!p.write_grad || (p.target && ((p.mask && p.inv_den) || p.grad_scale)), "xe: gradient needs target and mask+inv_den or grad_scale");
  UIC_REQUIRE(!p.write_grad || p.dlogits, "xe: null dlogits");
  UIC_REQUIRE(!p.target || p.row_loss, "xe: null row_loss");
  UIC_REQUIRE(!p.row_map || (p.write_grad && p.mask && !p.grad_scale && !p.logprobs),
              "xe: a row list goes with the masked criterion only (target, mask, gradient; no per-position scale or log-probabilities)");
  if (p.M == 0) return UIC_OK;
  const size_t row_bytes = (size_t)p.ldv * 4;
  if (p.dtype == UIC_BF16 && p.ldv % 4 == 0 && p.ldv <= XE_RCH * NT * 4 && ((uintptr_t)p.logits & 15) == 0 && p.write_grad && p.target &&
      !p.logprobs && !p.score_stats && ((uintptr_t)p.dlogits & 7) == 0) {
    hipLaunchKernelGGL(xe_reg_kernel<bf16_t>, dim3(p.M), dim3(NT), 0, s, p, p.logits, (bf16_t*)p.dlogits);
    UIC_LAUNCH_CHECK("xe_reg_kernel");
    return UIC_OK;
  }
  if (p.dtype == UIC_BF16 && p.ldv % 4 == 0 && p.ldv > XE_RCH * NT * 4 && p.ldv <= XE_WCH * XE_WTH * 4 && ((uintptr_t)p.logits & 15) == 0 &&
      p.write_grad && p.target && !p.logprobs && ((uintptr_t)p.dlogits & 7) == 0) {
    hipLaunchKernelGGL(xe_reg_wide_kernel<bf16_t>, dim3(p.M), dim3(XE_WTH), 0, s, p, p.logits, (bf16_t*)p.dlogits);
    UIC_LAUNCH_CHECK("xe_reg_wide_kernel");
    return UIC_OK;
  }
  if (p.dtype == UIC_BF16 && p.ldv % 4 == 0 && row_bytes <= 64 * 1024 && ((uintptr_t)p.logits & 15) == 0) {
    // bf16 path: LDS-staged row + hardware exp (the f32 parity path keeps libm exp)
    hipLaunchKernelGGL(xe_lds_kernel<bf16_t>, dim3(p.M), dim3(NT), row_bytes, s, p, p.logits, (bf16_t*)p.dlogits);
    UIC_LAUNCH_CHECK("xe_lds_kernel");
    return UIC_OK;
  }
  if (p.dtype == UIC_BF16 && p.ldv % 4 == 0 && ((uintptr_t)p.logits & 15) == 0 && (!p.dlogits || ((uintptr_t)p.dlogits & 7) == 0)) {
    hipLaunchKernelGGL(xe_big_kernel<bf16_t>, dim3(p.M), dim3(NT), 0, s, p, p.logits, (bf16_t*)p.dlogits);
    UIC_LAUNCH_CHECK("xe_big_kernel");
    return UIC_OK;
  }
  DISPATCH_T(p.dtype, hipLaunchKernelGGL(xe_kernel<bf16_t>, dim3(p.M), dim3(NT), 0, s, p, p.logits, (bf16_t*)p.dlogits),
             hipLaunchKernelGGL(xe_kernel<float>, dim3(p.M), dim3(NT), 0, s, p, p.logits, (float*)p.dlogits));
  UIC_LAUNCH_CHECK("xe_kernel");
  return UIC_OK;
}
int uic_logsoftmax_bwd_launch(int dtype, void* dlogits, int M, int V1, int ldv, int N, const float* g,
                              size_t g_step_stride, size_t g_row_stride, const float* logprobs, hipStream_t s) {
  UIC_REQUIRE(dlogits && g && logprobs && N > 0, "logsoftmax_bwd: null pointer");
  if (M == 0) return UIC_OK;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(logsoftmax_bwd_kernel<bf16_t>, dim3(M), dim3(NT), 0, s, (bf16_t*)dlogits, V1, ldv, N, g, g_step_stride, g_row_stride, logprobs),
             hipLaunchKernelGGL(logsoftmax_bwd_kernel<float>, dim3(M), dim3(NT), 0, s, (float*)dlogits, V1, ldv, N, g, g_step_stride, g_row_stride, logprobs));
  UIC_LAUNCH_CHECK("logsoftmax_bwd");
  return UIC_OK;
}
int uic_masked_sum_launch(const float* x, const float* mask, int ldmask, int col0, int N, int T, float* out_sum,
                          float* out_inv, hipStream_t s) {
  (void)x;
  hipLaunchKernelGGL(masked_sum_kernel, dim3(1), dim3(MS_NT), 0, s, mask, ldmask, col0, N, T, out_sum, out_inv);
  UIC_LAUNCH_CHECK("masked_sum");
  return UIC_OK;
}
int uic_zero4_launch(void* p0, size_t b0, void* p1, size_t b1, void* p2, size_t b2, void* p3, size_t b3, hipStream_t s) {
  void* ps[4] = {p0, p1, p2, p3};
  const size_t bs[4] = {b0, b1, b2, b3};
  UicZero4 z;
  size_t most = 0;
  for (int k = 0; k < 4; ++k) {
    UIC_REQUIRE(bs[k] == 0 || (ps[k] && ((uintptr_t)ps[k] & 15) == 0 && bs[k] % 16 == 0), "zero4: buffer %d must be 16-byte aligned / sized", k);
    z.p[k] = (uint4*)ps[k];
    z.n[k] = bs[k] / 16;
    if (z.n[k] > most) most = z.n[k];
  }
  if (most == 0) return UIC_OK;
  hipLaunchKernelGGL(zero4_kernel, dim3(grid_for(most, NT)), dim3(NT), 0, s, z);
  UIC_LAUNCH_CHECK("zero4");
  return UIC_OK;
}
int uic_zero_list_launch(void* const* ptrs, const size_t* bytes, int n, hipStream_t s) {
  for (int k0 = 0; k0 < n; k0 += UIC_ZERO_LIST) {
    UicZeroList z;
    memset(&z, 0, sizeof(z));
    size_t most = 0;
    int m = 0;
    for (int k = k0; k < n && k < k0 + UIC_ZERO_LIST; ++k) {
      if (bytes[k] == 0) continue;
      UIC_REQUIRE(ptrs[k] && ((uintptr_t)ptrs[k] & 15) == 0 && bytes[k] % 16 == 0, "zero_list: buffer %d must be 16-byte aligned / sized", k);
      z.p[m] = (uint4*)ptrs[k];
      z.n[m] = bytes[k] / 16;
      if (z.n[m] > most) most = z.n[m];
      ++m;
    }
    if (m == 0) continue;
    int gx = grid_for(most, NT);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(zero_list_kernel, dim3(gx, m), dim3(NT), 0, s, z);
    UIC_LAUNCH_CHECK("zero_list");
  }
  return UIC_OK;
}
int uic_reduce_sum_launch(const float* x, size_t n, float unused, const float* scale, float* out, hipStream_t s) {
  (void)unused;
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(NT), 0, s, x, n, scale, out);
  UIC_LAUNCH_CHECK("reduce_sum");
  return UIC_OK;
}
int uic_ss_sample_launch(const float* logits_prev, int N, int V1, int ldv, const int64_t* labels, int ld_labels, int t,
                         float ss_prob, unsigned seed, int64_t* used, int ld_used, hipStream_t s) {
  UIC_REQUIRE(logits_prev && labels && used && t >= 1 && t < 256, "ss_sample: bad arguments (t=%d)", t);
  if (N == 0) return UIC_OK;
  hipLaunchKernelGGL(ss_sample_kernel, dim3(N), dim3(NT), 0, s, logits_prev, V1, ldv, labels, ld_labels, t, ss_prob, seed, used, ld_used);
  UIC_LAUNCH_CHECK("ss_sample");
  return UIC_OK;
}
int uic_copy_tokens_launch(const int64_t* src, int ld_src, int N, int T, int64_t* dst, int ld_dst, hipStream_t s) {
  if (N * T == 0) return UIC_OK;
  hipLaunchKernelGGL(copy_tokens_kernel, dim3((N * T + NT - 1) / NT), dim3(NT), 0, s, src, ld_src, N, T, dst, ld_dst);
  UIC_LAUNCH_CHECK("copy_tokens");
  return UIC_OK;
}
int uic_sqnorm_launch(const float* g, size_t n, float* scratch, float* out, hipStream_t s) {
  UIC_REQUIRE(g && scratch && out, "sqnorm: null pointer");
  const int blocks = (int)(n / (NT * 8) > 1024 ? 1024 : (n / (NT * 8) ? n / (NT * 8) : 1));
  hipLaunchKernelGGL(sqnorm_part_kernel, dim3(blocks), dim3(NT), 0, s, g, n, scratch);
  UIC_LAUNCH_CHECK("sqnorm_part");
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(NT), 0, s, scratch, (size_t)blocks, (const float*)nullptr, out);
  UIC_LAUNCH_CHECK("sqnorm_final");
  return UIC_OK;
}
int uic_adam_ranges_launch(const UicAdamParams& a, const UicAdamRanges& r, void* w_out, int w_dtype, hipStream_t s) {
  if (r.total == 0) return UIC_OK;
  const int g = grid_for(r.total, NT);
  if (w_out && w_dtype == UIC_BF16) hipLaunchKernelGGL(adam_ranges_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, a, r, (bf16_t*)w_out);
  else hipLaunchKernelGGL(adam_ranges_kernel<float>, dim3(g), dim3(NT), 0, s, a, r, (float*)nullptr);   // (f32 operands ARE the masters)
  UIC_LAUNCH_CHECK("adam_ranges");
  return UIC_OK;
}
int uic_adam_launch(const UicAdamParams& a, hipStream_t s) {
  if (a.n == 0) return UIC_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(a.n, NT)), dim3(NT), 0, s, a);
  UIC_LAUNCH_CHECK("adam");
  return UIC_OK;
}
int uic_sample_step_launch(const UicSampleParams& p, hipStream_t s) {
  UIC_REQUIRE(p.logits && p.seq && p.seq_logp && p.it && p.unfinished && p.n_unfinished, "sample_step: null pointer");
  UIC_REQUIRE(p.t >= 0 && p.t < p.L, "sample_step: t=%d outside [0,%d)", p.t, p.L);
  if (p.N == 0) return UIC_OK;
  const size_t row_bytes = ((size_t)p.V1 * 4 + 15) & ~(size_t)15;
  const bool staged = row_bytes <= 60 * 1024 && p.ldv % 4 == 0 && ((uintptr_t)p.logits & 15) == 0;
  const bool fast = p.dtype == UIC_BF16;
  if (staged && fast) hipLaunchKernelGGL((sample_step_kernel<true, true>), dim3(p.N), dim3(NT), row_bytes, s, p);
  else if (staged) hipLaunchKernelGGL((sample_step_kernel<true, false>), dim3(p.N), dim3(NT), row_bytes, s, p);
  else if (fast) hipLaunchKernelGGL((sample_step_kernel<false, true>), dim3(p.N), dim3(NT), 0, s, p);
  else hipLaunchKernelGGL((sample_step_kernel<false, false>), dim3(p.N), dim3(NT), 0, s, p);
  UIC_LAUNCH_CHECK("sample_step");
  return UIC_OK;
}
int uic_maxout_lstm_bwd_launch(const UicLstmBwdParams& p, hipStream_t s) {
  UIC_REQUIRE(p.dc && p.gates && p.c && p.dgates, "maxout_lstm_bwd: null pointer");
  if (p.M == 0) return UIC_OK;
  const int g = grid_for((size_t)p.M * p.H, NT);
  DISPATCH_T(p.dtype, hipLaunchKernelGGL(maxout_lstm_bwd_kernel<bf16_t>, dim3(g), dim3(NT), 0, s, p),
             hipLaunchKernelGGL(maxout_lstm_bwd_kernel<float>, dim3(g), dim3(NT), 0, s, p));
  UIC_LAUNCH_CHECK("maxout_lstm_bwd");
  return UIC_OK;
}
int uic_sample_fixup_launch(int N, int L, int ld, const int* n_unfinished, int64_t* seq, float* seq_logp, hipStream_t s) {
  hipLaunchKernelGGL(sample_fixup_kernel, dim3(8), dim3(NT), 0, s, N, L, ld, n_unfinished, seq, seq_logp);
  UIC_LAUNCH_CHECK("sample_fixup");
  return UIC_OK;
}
int uic_dropout_mask_launch(float* out, size_t n, float p, unsigned seed, unsigned site, size_t base, hipStream_t s) {
  if (n == 0) return UIC_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, NT)), dim3(NT), 0, s, out, n, p, seed, site, base);
  UIC_LAUNCH_CHECK("dropout_mask");
  return UIC_OK;
}
