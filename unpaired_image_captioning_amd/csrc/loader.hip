// Batch assembly of the input pipeline on the device.
//
// The reference does this per image in numpy on DataLoader worker processes (P/misc/dataloader/dataloader.py:302-331,
// __getitem__) and then pads / replicates on the host (:264-283).  Here the host only reads the files into ONE packed,
// un-replicated buffer; this kernel L2-normalises the region features, builds the five box features, sorts the regions of
// an image by box area, pads to the longest image of the batch and writes the region mask -- one workgroup per output
// region row, every feature row read once and written once (HBM-bound; DESIGN.md "input pipeline").
//
// Bit-exactness: every value is produced with the same f32 operations in the same order as numpy produces it --
// np.linalg.norm(x, 2, 1) = sqrt(add.reduce(x * x)) with numpy's PAIRWISE summation (blocks of <= 128 elements summed in 8
// interleaved accumulators, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), halves split at a multiple of 8) -- so the
// batch is identical to the reference's, not just close to it.  sqrtf and / are the correctly rounded ones (hipcc's default;
// __fsqrt_rn is the 1-ulp v_sqrt_f32 here).  No fused multiply-add may be formed:
#pragma clang fp contract(off)

#include "uic_common.h"

namespace {

constexpr int LD_THREADS = 256;
constexpr int LD_MAX_LEAVES = 512;      // pairwise-sum leaves of one feature row (leaf >= 64 elements once D > 128)
constexpr int LD_MAX_D = 16384;
constexpr int LD_MAX_R = 2048;          // regions of one image (bottom-up features: 10..100)
constexpr int LD_PW_BLOCK = 128;        // numpy's PW_BLOCKSIZE

// The leaves of numpy's pairwise_sum recursion over n elements, in order, each with its depth in the recursion tree.
// (one thread; `stack` = 3 * 24 ints of LDS: private arrays indexed at run time would live in scratch memory)
__device__ int pw_leaves(int n, int* leaf_start, int* leaf_len, int* leaf_depth, int* stack) {
  int* st_s = stack; int* st_n = stack + 24; int* st_d = stack + 48;
  int sp = 0, nl = 0;
  st_s[0] = 0; st_n[0] = n; st_d[0] = 0; sp = 1;
  while (sp > 0) {
    --sp;
    const int s = st_s[sp], m = st_n[sp], d = st_d[sp];
    if (m <= LD_PW_BLOCK) {
      leaf_start[nl] = s; leaf_len[nl] = m; leaf_depth[nl] = d; ++nl;
    } else {
      int n2 = m / 2;
      n2 -= n2 % 8;
      st_s[sp] = s + n2; st_n[sp] = m - n2; st_d[sp] = d + 1; ++sp;      // right half: popped second
      st_s[sp] = s;      st_n[sp] = n2;     st_d[sp] = d + 1; ++sp;      // left half: popped first
    }
  }
  return nl;
}

// (x1/w, y1/h, x2/w, y2/h, (x2-x1)(y2-y1)/(wh)) [/ its own L2 norm]   (dataloader.py:318-323)
__device__ void box_features(const float* b, float h, float w, float wh, int norm_box, float* out) {
  const float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
  out[0] = x1 / w; out[1] = y1 / h; out[2] = x2 / w; out[3] = y2 / h;
  out[4] = ((x2 - x1) * (y2 - y1)) / wh;
  if (norm_box) {
    float s = 0.f;                                   // n < 8: numpy's plain loop
    for (int k = 0; k < 5; ++k) s = s + out[k] * out[k];
    const float nrm = sqrtf(s);
    for (int k = 0; k < 5; ++k) out[k] = out[k] / nrm;
  }
}

template <bool VEC4>
__global__ __launch_bounds__(LD_THREADS) void att_batch_assemble_kernel(
    const float* __restrict__ feat_pack, const float* __restrict__ box_pack, const int32_t* __restrict__ region_start,
    const float* __restrict__ img_hw, const int32_t* __restrict__ img_slot, int D, int norm_att, int norm_box, int Rmax,
    int ld_out, float* __restrict__ att_feats, float* __restrict__ att_masks) {
  extern __shared__ float smem[];
  float* xs = smem;                                  // [D] the raw feature row
  __shared__ int leaf_start[LD_MAX_LEAVES], leaf_len[LD_MAX_LEAVES], leaf_depth[LD_MAX_LEAVES];
  __shared__ float leaf_sum[LD_MAX_LEAVES];
  __shared__ int n_leaves_s, rank_s, dfs_stack[72];
  __shared__ float norm_s, mybox[5], merge_v[24];
  __shared__ int merge_d[24];

  const int img = blockIdx.x / Rmax, q = blockIdx.x % Rmax;
  const int r0 = region_start[img], Ri = region_start[img + 1] - r0;
  const int slot = img_slot[img];
  const int tid = threadIdx.x;
  const int Dout = box_pack ? D + 5 : D;

  if (q >= Ri) {                                     // padding row of this image: zeros, mask 0 (:279-283)
    float* dst = att_feats + ((size_t)slot * Rmax + q) * ld_out;
    for (int c = tid; c < ld_out; c += LD_THREADS) dst[c] = 0.f;
    if (tid == 0) att_masks[(size_t)slot * Rmax + q] = 0.f;
    return;
  }

  const float* src = feat_pack + (size_t)(r0 + q) * D;
  if (VEC4) {
    for (int c = tid * 4; c < D; c += LD_THREADS * 4) *(float4*)(xs + c) = *(const float4*)(src + c);
  } else {
    for (int c = tid; c < D; c += LD_THREADS) xs[c] = src[c];
  }
  if (tid == 0) {
    rank_s = 0;
    n_leaves_s = norm_att ? pw_leaves(D, leaf_start, leaf_len, leaf_depth, dfs_stack) : 0;
  }
  __syncthreads();

  // ---- where this region goes: regions sorted by the LAST column, descending, stable (:327) ----
  int rank = q;
  if (box_pack) {
    const float h = img_hw[img * 3], w = img_hw[img * 3 + 1], wh = img_hw[img * 3 + 2];
    float mine[5];
    box_features(box_pack + (size_t)(r0 + q) * 4, h, w, wh, norm_box, mine);
    int cnt = 0;
    for (int j = tid; j < Ri; j += LD_THREADS) {
      float other[5];
      box_features(box_pack + (size_t)(r0 + j) * 4, h, w, wh, norm_box, other);
      cnt += (other[4] > mine[4]) || (other[4] == mine[4] && j < q);
    }
    if (cnt) atomicAdd(&rank_s, cnt);
    if (tid == 0) for (int k = 0; k < 5; ++k) mybox[k] = mine[k];
  }

  // ---- || x ||_2 exactly as numpy sums it (:311) ----
  if (norm_att) {
    const int nl = n_leaves_s;
    const int j = tid & 7;
    for (int lf = tid >> 3; lf < nl; lf += LD_THREADS / 8) {
      const int s = leaf_start[lf], m = leaf_len[lf];
      float res;
      if (m < 8) {                                   // the whole row is shorter than 8: plain loop from zero
        res = 0.f;
        for (int i = 0; i < m; ++i) res = res + xs[s + i] * xs[s + i];
      } else {
        float r = xs[s + j] * xs[s + j];
        const int m8 = m - (m % 8);
        for (int i = 8; i < m8; i += 8) r = r + xs[s + i + j] * xs[s + i + j];
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        res = r;
        for (int i = m8; i < m; ++i) res = res + xs[s + i] * xs[s + i];
      }
      if (j == 0) leaf_sum[lf] = res;
    }
    __syncthreads();
    if (tid == 0) {                                  // pw(left) + pw(right), bottom-up: merge equal depths
      int sp = 0;
      for (int lf = 0; lf < nl; ++lf) {
        float v = leaf_sum[lf]; int d = leaf_depth[lf];
        while (sp > 0 && merge_d[sp - 1] == d) { v = merge_v[sp - 1] + v; --sp; --d; }
        merge_v[sp] = v; merge_d[sp] = d; ++sp;
      }
      norm_s = sqrtf(0.f + merge_v[0]);
    }
  }
  __syncthreads();
  if (box_pack) rank = rank_s;

  float* dst = att_feats + ((size_t)slot * Rmax + rank) * ld_out;
  const float nrm = norm_att ? norm_s : 1.f;
  if (VEC4) {
    for (int c = tid * 4; c < D; c += LD_THREADS * 4) {
      float4 v = *(const float4*)(xs + c);
      if (norm_att) { v.x = v.x / nrm; v.y = v.y / nrm; v.z = v.z / nrm; v.w = v.w / nrm; }
      *(float4*)(dst + c) = v;
    }
  } else {
    for (int c = tid; c < D; c += LD_THREADS) dst[c] = norm_att ? xs[c] / nrm : xs[c];
  }
  for (int c = D + tid; c < ld_out; c += LD_THREADS) dst[c] = (box_pack && c < Dout) ? mybox[c - D] : 0.f;
  if (tid == 0) att_masks[(size_t)slot * Rmax + rank] = 1.f;
}


// D = 128 * 2^k, k <= 5 (2048: the bottom-up features): every halving of numpy's recursion lands on a multiple of 8, so
// the tree is a perfect binary tree over 2^k leaves of 128 elements and the whole sum is xor-butterflies.  ONE WAVE per
// region row, the row held in registers (K float4 per lane, all loads in flight at once), LDS used only to hand each lane
// the 16 elements of its accumulator (leaf rows skewed by 8 floats: the 8 leaves of a pass hit 64 distinct banks); no
// serial section, every row of a 128 x 36 batch resident at once (18 workgroups of 8.5 KB per CU).
template <int K>
__global__ __launch_bounds__(64) void att_batch_assemble_wave_kernel(
    const float* __restrict__ feat_pack, const float* __restrict__ box_pack, const int32_t* __restrict__ region_start,
    const float* __restrict__ img_hw, const int32_t* __restrict__ img_slot, int D, int norm_att, int norm_box, int Rmax,
    int ld_out, float* __restrict__ att_feats, float* __restrict__ att_masks) {
  extern __shared__ float xs[];                      // [D + 8 * D / 128]
  const int img = blockIdx.x / Rmax, q = blockIdx.x % Rmax;
  const int r0 = region_start[img], Ri = region_start[img + 1] - r0;
  const int slot = img_slot[img];
  const int lane = threadIdx.x;
  const int Dout = box_pack ? D + 5 : D;

  if (q >= Ri) {
    float* dst = att_feats + ((size_t)slot * Rmax + q) * ld_out;
    for (int c = lane * 4; c < ld_out; c += 256) *(float4*)(dst + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane == 0) att_masks[(size_t)slot * Rmax + q] = 0.f;
    return;
  }

  const float* src = feat_pack + (size_t)(r0 + q) * D;
  float4 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < D)                                         // read once, written once: keep both out of the caches' way
      v[k] = make_float4(__builtin_nontemporal_load(src + c), __builtin_nontemporal_load(src + c + 1),
                         __builtin_nontemporal_load(src + c + 2), __builtin_nontemporal_load(src + c + 3));
  }

  int rank = q;
  float mine[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (box_pack) {
    const float h = img_hw[img * 3], w = img_hw[img * 3 + 1], wh = img_hw[img * 3 + 2];
    box_features(box_pack + (size_t)(r0 + q) * 4, h, w, wh, norm_box, mine);
    int cnt = 0;
    for (int j = lane; j < Ri; j += 64) {
      float other[5];
      box_features(box_pack + (size_t)(r0 + j) * 4, h, w, wh, norm_box, other);
      cnt += (other[4] > mine[4]) || (other[4] == mine[4] && j < q);
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    rank = cnt;
  }

  float nrm = 1.f;
  if (norm_att) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c = (lane + 64 * k) * 4;
      if (c < D) *(float4*)(xs + c + (c >> 7) * 8) = v[k];
    }
    __syncthreads();
    const int nl = D / LD_PW_BLOCK;                  // 1, 2, 4, .. 32 leaves; a pass sums 8 of them: lane = (leaf % 8, accumulator)
    const int j = lane & 7;
    constexpr int NP = K >= 4 ? K / 4 : 1;
    float tot[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int lf = p * 8 + (lane >> 3);
      float r = 0.f;
      if (lf < nl) {
        const float* a = xs + lf * (LD_PW_BLOCK + 8) + j;
        r = a[0] * a[0];
#pragma unroll
        for (int i = 8; i < LD_PW_BLOCK; i += 8) r = r + a[i] * a[i];
      }
      r = r + __shfl_xor(r, 1);                      // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
      r = r + __shfl_xor(r, 2);
      r = r + __shfl_xor(r, 4);
      if (nl >= 2) r = r + __shfl_xor(r, 8);         // leaf pairs, pairs of pairs, ..: the recursion, bottom-up
      if (nl >= 4) r = r + __shfl_xor(r, 16);
      if (nl >= 8) r = r + __shfl_xor(r, 32);
      tot[p] = r;
    }
    float total = tot[0];
    if (NP == 2) total = tot[0] + tot[1];
    if (NP == 4) total = (tot[0] + tot[1]) + (tot[2] + tot[NP - 1]);
    nrm = sqrtf(0.f + __shfl(total, 0));             // (with fewer than 8 leaves only the low lanes hold the sum)
  }

  float* dst = att_feats + ((size_t)slot * Rmax + rank) * ld_out;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < D) {
      float4 x = v[k];
      if (norm_att) { x.x = x.x / nrm; x.y = x.y / nrm; x.z = x.z / nrm; x.w = x.w / nrm; }
      __builtin_nontemporal_store(x.x, dst + c); __builtin_nontemporal_store(x.y, dst + c + 1);
      __builtin_nontemporal_store(x.z, dst + c + 2); __builtin_nontemporal_store(x.w, dst + c + 3);
    }
  }
  for (int c = D + lane; c < ld_out; c += 64) dst[c] = (box_pack && c < Dout) ? mine[c - D] : 0.f;
  if (lane == 0) att_masks[(size_t)slot * Rmax + rank] = 1.f;
}

}  // namespace

extern "C" {

int uic_att_batch_assemble(const float* feat_pack, const float* box_pack, const int32_t* region_start, const float* img_hw,
                           const int32_t* img_slot, int32_t n_img, int32_t D, int32_t norm_att_feat, int32_t norm_box_feat,
                           int32_t Rmax, int32_t ld_out, float* att_feats, float* att_masks, void* stream) {
  UIC_REQUIRE(feat_pack && region_start && img_slot && att_feats && att_masks, "att_batch_assemble: null pointer");
  UIC_REQUIRE(!box_pack || img_hw, "att_batch_assemble: box features need the image sizes (img_hw)");
  UIC_REQUIRE(n_img >= 1 && Rmax >= 1 && Rmax <= LD_MAX_R, "att_batch_assemble: n_img=%d Rmax=%d (max %d regions per image)", n_img, Rmax, LD_MAX_R);
  UIC_REQUIRE(D >= 1 && D <= LD_MAX_D, "att_batch_assemble: D=%d (max %d)", D, LD_MAX_D);
  UIC_REQUIRE(ld_out >= D + (box_pack ? 5 : 0), "att_batch_assemble: ld_out=%d is shorter than the %d output columns", ld_out, D + (box_pack ? 5 : 0));
  UIC_REQUIRE((int64_t)n_img * Rmax < (int64_t)1 << 31, "att_batch_assemble: %d x %d output rows", n_img, Rmax);
  const bool vec4 = D % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)feat_pack % 16) == 0 && ((uintptr_t)att_feats % 16) == 0;
  const size_t lds = sizeof(float) * (size_t)((D + 3) / 4 * 4);
  hipStream_t s = (hipStream_t)stream;
  const int leaves = D / LD_PW_BLOCK;
  const bool regular = vec4 && D % LD_PW_BLOCK == 0 && (leaves & (leaves - 1)) == 0 && leaves <= 32;
#define UIC_ASSEMBLE(V)                                                                                                \
  hipLaunchKernelGGL((att_batch_assemble_kernel<V>), dim3(n_img * Rmax), dim3(LD_THREADS), lds, s, feat_pack, box_pack,   \
                     region_start, img_hw, img_slot, D, norm_att_feat, norm_box_feat, Rmax, ld_out, att_feats, att_masks)
#define UIC_ASSEMBLE_WAVE(K)                                                                                           \
  hipLaunchKernelGGL((att_batch_assemble_wave_kernel<K>), dim3(n_img * Rmax), dim3(64), sizeof(float) * (D + D / 16), s, \
                     feat_pack, box_pack, region_start, img_hw, img_slot, D, norm_att_feat, norm_box_feat, Rmax, ld_out,  \
                     att_feats, att_masks)
  if (regular) {
    if (D <= 256) UIC_ASSEMBLE_WAVE(1);
    else if (D == 512) UIC_ASSEMBLE_WAVE(2);
    else if (D == 1024) UIC_ASSEMBLE_WAVE(4);
    else if (D == 2048) UIC_ASSEMBLE_WAVE(8);
    else UIC_ASSEMBLE_WAVE(16);
  } else if (vec4) UIC_ASSEMBLE(true);
  else UIC_ASSEMBLE(false);
#undef UIC_ASSEMBLE
#undef UIC_ASSEMBLE_WAVE
  UIC_LAUNCH_CHECK("att_batch_assemble");
  return UIC_OK;
}

}  // extern "C"
