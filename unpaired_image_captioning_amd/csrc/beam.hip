// Beam search of the captioner on the device: AttModel._sample_beam (P/models/AttModel.py:167-196) +
// CaptionModel.beam_search (P/models/CaptionModel.py:33-177) with group_size = 1, ALL images of the batch at once
// (the reference decodes image by image with a full torch.sort over the vocabulary per beam and step, :61).
//
// Rows of every per-step tensor are (image, beam): row = img * B + beam.  Per decode step:
//   beam_topk_kernel   one workgroup per row: log-softmax of the logits row, the reference's modifications
//                      (previous word -> -inf with decoding_constraint :130-131, last vocabulary index -1000 :133) and the
//                      B best entries, value-descending, lowest index first on ties (the part of the sort :61 that is used);
//   beam_merge_kernel  one thread per image: candidates enumerated word-rank-major / beam-minor (:67-73), the B best by
//                      joint log-prob with the stable order of sorted() (:74), history re-threading (:83-95), finished
//                      beams copied to the done list and their running sum set to -1000 (:147-161);
//   beam_gather_kernel LSTM states re-threaded to the surviving parents (:90-92).
// beam_final_kernel ranks the done list (stable, by p) and writes the best beam's tokens and log-probs (:174-176,
// AttModel.py:193-194).
#include "uic_common.h"
#include <stdlib.h>
#include <stdint.h>

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float bmax(float v, float* s_buf) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_buf[wave] = v;
  __syncthreads();
  float r = s_buf[0];
  for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, s_buf[i]);
  return r;
}
__device__ __forceinline__ float bsum(float v, float* s_buf) {
  v = uic_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_buf[wave] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < NT / 64; ++i) r += s_buf[i];
  return r;
}

// three passes over the row (maximum, sum, candidates), one word per lane and load: the shorter rows of the captioner's
// vocabulary (9 488 words stay in the L2 between the passes); rows of >= 16 384 words take the one-pass kernel below
__global__ __launch_bounds__(NT) void beam_topk3_kernel(const UicBeamParams p) {
  __shared__ float s_buf[NT / 64];
  __shared__ float s_val[NT];
  __shared__ int s_idx[NT];
  const int row = blockIdx.x;
  const float* x = p.logits + (size_t)row * p.ldv;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < p.V1; v += NT) mx = fmaxf(mx, x[v]);
  mx = bmax(mx, s_buf);
  float sum = 0.f;
  for (int v = threadIdx.x; v < p.V1; v += NT) sum += expf(x[v] - mx);
  sum = bsum(sum, s_buf);
  const float lse = mx + logf(sum);
  long banned = -1;
  if (p.decoding_constraint && p.t > 0 && !p.plain) banned = p.beam_seq[((size_t)(row / p.B) * p.L + (p.t - 1)) * p.B + (row % p.B)];
  const int V1 = p.V1;
  auto value = [&](int v) {
    float lp = x[v] - lse;
    if (v == banned) lp = -INFINITY;
    if (v == V1 - 1 && !p.plain) lp -= 1000.f;
    return lp;
  };
  // order of the result: value descending, lower index first on ties (what the reference's sort + enumeration uses)
  auto better = [](float av, int ai, float bv_, int bi_) { return av > bv_ || (av == bv_ && ai < bi_); };
  // Selection without B passes over the row: every thread keeps the best of ITS elements (v = tid, tid + NT, ...); each
  // round the block picks the best of the 256 thread-bests, and only the wave of the thread that owned it rescans that
  // thread's elements for its next-best (everything that sorts strictly after the one just taken).
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = tid; v < V1; v += NT) {
    const float lp = value(v);
    if (better(lp, v, bv, bi)) { bv = lp; bi = v; }
  }
  float* s_wv = s_val;            // [NT / 64] per-wave winners
  int* s_wi = s_idx;
  for (int k = 0; k < p.B; ++k) {
    float wv = bv;
    int wi = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(wv, o, 64);
      const int oi = __shfl_xor(wi, o, 64);
      if (better(ov, oi, wv, wi)) { wv = ov; wi = oi; }
    }
    __syncthreads();                                  // the previous round's readers are done with s_wv / s_wi
    if (lane == 0) { s_wv[wave] = wv; s_wi[wave] = wi; }
    __syncthreads();
    float gv = s_wv[0];
    int gi = s_wi[0];
#pragma unroll
    for (int w2 = 1; w2 < NT / 64; ++w2)
      if (better(s_wv[w2], s_wi[w2], gv, gi)) { gv = s_wv[w2]; gi = s_wi[w2]; }
    if (tid == 0) {
      p.cand_val[(size_t)row * p.B + k] = gv;
      p.cand_idx[(size_t)row * p.B + k] = gi;
    }
    if (gi == 0x7fffffff) continue;                   // nothing left (only possible when B exceeds the row length)
    const int owner = gi % NT;
    if (wave == (owner >> 6)) {
      float nv = -INFINITY;
      int ni = 0x7fffffff;
      for (int v = owner + lane * NT; v < V1; v += 64 * NT) {
        const float lp = value(v);
        const bool after = lp < gv || (lp == gv && v > gi);          // sorts strictly after the element just taken
        if (after && better(lp, v, nv, ni)) { nv = lp; ni = v; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(nv, o, 64);
        const int oi = __shfl_xor(ni, o, 64);
        if (better(ov, oi, nv, ni)) { nv = ov; ni = oi; }
      }
      if (tid == owner) { bv = nv; bi = ni; }
    }
  }
}

__global__ __launch_bounds__(NT) void beam_topk_kernel(const UicBeamParams p) {
  __shared__ float s_buf[NT / 64];
  __shared__ float s_val[NT];
  __shared__ int s_idx[NT];
  const int row = blockIdx.x;
  const float* x = p.logits + (size_t)row * p.ldv;
  long banned = -1;
  if (p.decoding_constraint && p.t > 0 && !p.plain) banned = p.beam_seq[((size_t)(row / p.B) * p.L + (p.t - 1)) * p.B + (row % p.B)];
  const int V1 = p.V1;
  // the reference's modifications act on single indices and log-softmax shifts the whole row by one number, so the order of
  // the candidates can be read off the raw logits: `raw` is everything of value() except the common - lse
  auto raw = [&](int v, float xv) {
    if (v == banned) xv = -INFINITY;
    if (v == V1 - 1 && !p.plain) xv -= 1000.f;
    return xv;
  };
  // order of the result: value descending, lower index first on ties (what the reference's sort + enumeration uses)
  auto better = [](float av, int ai, float bv_, int bi_) { return av > bv_ || (av == bv_ && ai < bi_); };
  // ONE pass over the row (a 50 004-word row is 200 KB and there are ~1 000 of them per step: three passes were three trips to
  // HBM / the Infinity Cache): every thread keeps a running maximum with the sum of exponentials rescaled to it, and the best
  // of ITS elements -- element groups g = tid, tid + NT, ... of four consecutive words, one 16-byte load each.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float bv = -INFINITY, bv2 = -INFINITY;
  int bi = 0x7fffffff, bi2 = 0x7fffffff;
  float mx = -INFINITY, sum = 0.f;
  // exp(d), d <= 0, as one v_exp_f32 (2^(d log2 e)): expf's range handling is most of this pass's arithmetic
  auto fexp = [](float d) { return __builtin_amdgcn_exp2f(d * 1.44269504088896340736f); };
  const bool vec_ok = p.ldv % 4 == 0 && ((size_t)p.logits & 15) == 0;
  // four groups per trip: their loads are in flight together (one 16-byte load per trip left the kernel waiting on latency)
  for (int vb = tid * 4; vb < V1; vb += NT * 4 * 4) {
    float qq[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int v0 = vb + g * NT * 4;
      qq[g][0] = qq[g][1] = qq[g][2] = qq[g][3] = -INFINITY;
      if (vec_ok && v0 + 3 < V1) {
        const float4 f = *(const float4*)(x + v0);
        qq[g][0] = f.x; qq[g][1] = f.y; qq[g][2] = f.z; qq[g][3] = f.w;
      } else {
        for (int c = 0; c < 4 && v0 + c < V1; ++c) qq[g][c] = x[v0 + c];
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int v0 = vb + g * NT * 4;
      const float* q = qq[g];
      const float m4 = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3]));
      if (m4 > mx) { sum *= fexp(mx - m4); mx = m4; }        // (exp(-inf) = 0 on the first group)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sum += m4 == -INFINITY ? 0.f : fexp(q[c] - mx);      // slots past the row hold -inf: + 0
        const float rv = raw(v0 + c, q[c]);
        if (v0 + c < V1) {                                    // the thread's best and second best (raw values)
          if (better(rv, v0 + c, bv, bi)) { bv2 = bv; bi2 = bi; bv = rv; bi = v0 + c; }
          else if (better(rv, v0 + c, bv2, bi2)) { bv2 = rv; bi2 = v0 + c; }
        }
      }
    }
  }
  const float gmx = bmax(mx, s_buf);
  sum = bsum(sum * expf(mx - gmx), s_buf);
  const float lse = gmx + logf(sum);
  bv -= lse;
  bv2 -= lse;
  bool has2 = true;                                   // bv2 / bi2 is this thread's next element in the order (until it is used)
  auto value = [&](int v) { return raw(v, x[v]) - lse; };
  float* s_wv = s_val;            // [NT / 64] per-wave winners
  int* s_wi = s_idx;
  for (int k = 0; k < p.B; ++k) {
    float wv = bv;
    int wi = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(wv, o, 64);
      const int oi = __shfl_xor(wi, o, 64);
      if (better(ov, oi, wv, wi)) { wv = ov; wi = oi; }
    }
    __syncthreads();                                  // the previous round's readers are done with s_wv / s_wi
    if (lane == 0) { s_wv[wave] = wv; s_wi[wave] = wi; }
    __syncthreads();
    float gv = s_wv[0];
    int gi = s_wi[0];
#pragma unroll
    for (int w2 = 1; w2 < NT / 64; ++w2)
      if (better(s_wv[w2], s_wi[w2], gv, gi)) { gv = s_wv[w2]; gi = s_wi[w2]; }
    if (tid == 0) {
      p.cand_val[(size_t)row * p.B + k] = gv;
      p.cand_idx[(size_t)row * p.B + k] = gi;
    }
    if (gi == 0x7fffffff) continue;                   // nothing left (only possible when B exceeds the row length)
    const int owner = (gi >> 2) % NT;                 // the thread whose element groups hold word gi
    // its second best, kept from the pass, is its next element in the order: only a thread that wins a third time goes back to
    // its 195 words in memory (the rescan, a trip to L2 per selection round, was half of this kernel)
    if (wave == (owner >> 6) && __shfl(has2 ? 1 : 0, owner & 63, 64)) {
      if (tid == owner) { bv = bv2; bi = bi2; has2 = false; }
    } else if (wave == (owner >> 6)) {
      float nv = -INFINITY;
      int ni = 0x7fffffff;
      for (int g = owner + (lane >> 2) * NT; g * 4 < V1; g += 16 * NT) {   // 4 lanes per group, 16 groups per round
        const int v = g * 4 + (lane & 3);
        if (v >= V1) continue;
        const float lp = value(v);
        const bool after = lp < gv || (lp == gv && v > gi);          // sorts strictly after the element just taken
        if (after && better(lp, v, nv, ni)) { nv = lp; ni = v; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(nv, o, 64);
        const int oi = __shfl_xor(ni, o, 64);
        if (better(ov, oi, nv, ni)) { nv = ov; ni = oi; }
      }
      if (tid == owner) { bv = nv; bi = ni; }
    }
  }
}

__global__ void beam_merge_kernel(const UicBeamParams p) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= p.n_img) return;
  const int B = p.B, L = p.L, t = p.t;
  const int rows = t == 0 ? 1 : B;
  float* sum = p.beam_sum + (size_t)img * B;
  // histories are double-buffered: read generation t & 1, write the other
  const int64_t* seq_r = p.beam_seq_hist[t & 1] + (size_t)img * L * B;
  const float* lp_r = p.beam_lp_hist[t & 1] + (size_t)img * L * B;
  int64_t* seq_w = p.beam_seq_hist[(t & 1) ^ 1] + (size_t)img * L * B;
  float* lp_w = p.beam_lp_hist[(t & 1) ^ 1] + (size_t)img * L * B;
  bool used[UIC_BEAM_MAX * UIC_BEAM_MAX];
  for (int i = 0; i < B * rows; ++i) used[i] = false;
  float new_sum[UIC_BEAM_MAX];
  for (int vix = 0; vix < B; ++vix) {
    // stable arg-max over candidates in (c major, q minor) order
    int best = -1;
    float bp = 0.f;
    for (int c = 0; c < B; ++c)
      for (int q = 0; q < rows; ++q) {
        const int id = c * rows + q;
        if (used[id]) continue;
        const float pj = sum[q] + p.cand_val[((size_t)img * B + q) * B + c];
        if (best < 0 || pj > bp) { best = id; bp = pj; }
      }
    used[best] = true;
    const int c = best / rows, q = best - c * rows;
    const size_t cr = ((size_t)img * B + q) * B + c;
    for (int tt = 0; tt < t; ++tt) {
      seq_w[(size_t)tt * B + vix] = seq_r[(size_t)tt * B + q];
      lp_w[(size_t)tt * B + vix] = lp_r[(size_t)tt * B + q];
    }
    const int tok = p.cand_idx[cr];
    seq_w[(size_t)t * B + vix] = tok;
    lp_w[(size_t)t * B + vix] = p.cand_val[cr];
    new_sum[vix] = bp;
    p.parent[(size_t)img * B + vix] = q;
    p.it[(size_t)img * B + vix] = tok;
  }
  int cnt = p.done_count[img];
  for (int vix = 0; vix < B; ++vix) {
    float s = new_sum[vix];
    if (seq_w[(size_t)t * B + vix] == 0 || t == L - 1) {
      const size_t e = (size_t)img * L * B + cnt;
      p.done_p[e] = p.max_ppl ? s / (float)(t + 1) : s;
      for (int tt = 0; tt < L; ++tt) {
        p.done_seq[e * L + tt] = tt <= t ? seq_w[(size_t)tt * B + vix] : 0;
        p.done_lp[e * L + tt] = tt <= t ? lp_w[(size_t)tt * B + vix] : 0.f;
      }
      ++cnt;
      s = -1000.f;
    }
    sum[vix] = s;
  }
  p.done_count[img] = cnt;
}

// dst[row] = src[img * B + parent[row]] for the four state tensors
template <typename T>
__global__ void beam_gather_kernel(const int* __restrict__ parent, int B, int H, size_t total, const T* __restrict__ h1s, T* __restrict__ h1d,
                                   const T* __restrict__ h2s, T* __restrict__ h2d, const float* __restrict__ c1s, float* __restrict__ c1d,
                                   const float* __restrict__ c2s, float* __restrict__ c2d) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / H;
    const int j = (int)(i - row * H);
    const size_t src = (row / B) * B + parent[row];
    const size_t o = src * H + j;
    h1d[i] = h1s[o];
    if (c1s) c1d[i] = c1s[o];
    if (h2s) { h2d[i] = h2s[o]; c2d[i] = c2s[o]; }   // second LSTM layer (none in the FC model)
  }
}

__global__ void beam_final_kernel(const UicBeamParams p, int64_t* __restrict__ seq_out, float* __restrict__ lp_out) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= p.n_img) return;
  const int L = p.L, B = p.B;
  const int cnt = p.done_count[img];
  int best = 0;
  for (int i = 1; i < cnt; ++i)
    if (p.done_p[(size_t)img * L * B + i] > p.done_p[(size_t)img * L * B + best]) best = i;
  const size_t e = (size_t)img * L * B + best;
  for (int tt = 0; tt < L; ++tt) {
    seq_out[(size_t)img * L + tt] = cnt ? p.done_seq[e * L + tt] : 0;
    lp_out[(size_t)img * L + tt] = cnt ? p.done_lp[e * L + tt] : 0.f;
  }
}

}  // namespace

static int launch_topk(const UicBeamParams& p, hipStream_t s) {
  if (p.V1 >= 16384) hipLaunchKernelGGL(beam_topk_kernel, dim3(p.n_img * p.B), dim3(NT), 0, s, p);
  else hipLaunchKernelGGL(beam_topk3_kernel, dim3(p.n_img * p.B), dim3(NT), 0, s, p);
  UIC_LAUNCH_CHECK("beam_topk");
  return UIC_OK;
}

int uic_beam_step_launch(const UicBeamParams& p, hipStream_t s) {
  UIC_REQUIRE(p.B >= 1 && p.B <= UIC_BEAM_MAX && p.B <= p.V1, "beam search: beam_size=%d outside [1, min(%d, V1)]", p.B, UIC_BEAM_MAX);
  UIC_REQUIRE(p.n_img > 0 && p.L > 0 && p.t >= 0 && p.t < p.L, "beam search: bad sizes");
  UicBeamParams q = p;
  q.beam_seq = p.beam_seq_hist[p.t & 1];     // generation holding steps < t
  UIC_TRY(launch_topk(q, s));
  hipLaunchKernelGGL(beam_merge_kernel, dim3((p.n_img + 63) / 64), dim3(64), 0, s, q);
  UIC_LAUNCH_CHECK("beam_merge");
  return UIC_OK;
}

int uic_beam_topk_launch(const UicBeamParams& p, hipStream_t s) {
  UIC_REQUIRE(p.B >= 1 && p.B <= UIC_BEAM_MAX && p.B <= p.V1 && p.n_img > 0, "beam topk: bad sizes (beam_size=%d)", p.B);
  return launch_topk(p, s);
}

int uic_beam_gather_launch(int dtype, const int* parent, int rows, int B, int H, const void* h1s, void* h1d, const void* h2s, void* h2d,
                           const float* c1s, float* c1d, const float* c2s, float* c2d, hipStream_t s) {
  const size_t total = (size_t)rows * H;
  size_t g = (total + NT - 1) / NT;
  if (g > 4096) g = 4096;
  if (dtype == UIC_BF16)
    hipLaunchKernelGGL(beam_gather_kernel<bf16_t>, dim3((unsigned)g), dim3(NT), 0, s, parent, B, H, total, (const bf16_t*)h1s, (bf16_t*)h1d,
                       (const bf16_t*)h2s, (bf16_t*)h2d, c1s, c1d, c2s, c2d);
  else
    hipLaunchKernelGGL(beam_gather_kernel<float>, dim3((unsigned)g), dim3(NT), 0, s, parent, B, H, total, (const float*)h1s, (float*)h1d,
                       (const float*)h2s, (float*)h2d, c1s, c1d, c2s, c2d);
  UIC_LAUNCH_CHECK("beam_gather");
  return UIC_OK;
}

int uic_beam_final_launch(const UicBeamParams& p, int64_t* seq_out, float* lp_out, hipStream_t s) {
  hipLaunchKernelGGL(beam_final_kernel, dim3((p.n_img + 63) / 64), dim3(64), 0, s, p, seq_out, lp_out);
  UIC_LAUNCH_CHECK("beam_final");
  return UIC_OK;
}
