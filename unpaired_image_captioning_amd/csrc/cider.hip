// CIDEr-D reward of the self-critical step on the device: the scorer of
// P/misc/cider/pyciderevalcap/ciderD/ciderD_scorer.py:116-209 as get_self_critical_reward drives it
// (P/misc/rewards.py:37-81), on integer token rows instead of strings of token ids, so that the sampled and the greedy
// captions never leave the GPU (the reference round-trips them through the host and scores them in python loops,
// which serialises the step).
//
// One workgroup of six waves per hypothesis (below).  A caption's words are its tokens up to and including the first 0 (array_to_str,
// rewards.py:29-35).  For the hypothesis and then for each reference of its image:
//   cook    lanes over the (order k, position i) n-grams: term frequency = number of equal n-grams of the caption, kept at
//           the FIRST occurrence only (precook's dict, :13-28); tf-idf weight tf * (ref_len - log(max(1, df))) with
//           log(max(1, df)) from an open-addressing hash table keyed by the n-gram's tokens (counts2vec, :117-139);
//   match   lanes over the hypothesis' n-grams: the equal n-gram of the reference -> min(w_hyp, w_ref) * w_ref (:159-161).
// The sums that shape the result (norms, similarities, the mean over n and over references, :133,161-197) are added by
// one lane per order in the reference's own order (dict insertion order = first occurrence), in f64, and the Gaussian
// length penalty comes from a host-made table of e^(-delta^2 / (2 sigma^2)) -- so scores agree with the python scorer to
// the last bits.  "length" is the number of bigrams (:134-135), as in the reference.
#include "uic_common.h"

namespace {

constexpr int MAXW = 64;        // words per caption (seq_length <= 64)
constexpr int NG = 4;           // n-gram orders 1..4

__host__ __device__ inline uint64_t ngram_hash(int t0, int t1, int t2, int t3) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  const int t[4] = {t0, t1, t2, t3};
  for (int i = 0; i < 4; ++i) {
    h ^= (uint64_t)(uint32_t)t[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33;
  }
  return h;
}

struct CiderParams {
  const int64_t* hyp; int n_hyp, L;             // [n_hyp, L]
  int batch_size, seq_per_img;                   // hypothesis h is scored against image (h % batch_size) / seq_per_img
  const int64_t* ref_tok; int Lr;                // [n_refs_total, Lr]
  const int32_t* ref_start;                      // [n_img + 1]
  const int32_t* slot_keys; const double* slot_vals; int64_t slots;   // slots = 0: every n-gram has df 0
  double ref_len;
  const double* penalty; int pen_half;           // penalty[delta + pen_half]
  double* scores;
};

__device__ __forceinline__ double table_lookup(const CiderParams& p, int t0, int t1, int t2, int t3) {
  if (p.slots == 0) return 0.0;
  const uint64_t mask = (uint64_t)p.slots - 1;
  uint64_t s = ngram_hash(t0, t1, t2, t3) & mask;
  for (int64_t probe = 0; probe < p.slots; ++probe) {
    const int4 k = *(const int4*)(p.slot_keys + s * 4);
    if (k.x == t0 && k.y == t1 && k.z == t2 && k.w == t3) return p.slot_vals[s];
    if (k.x < 0) return 0.0;                     // empty slot: the n-gram is not in the table, log(max(1, 0)) = 0
    s = (s + 1) & mask;
  }
  return 0.0;
}

// words of a caption row into LDS by one wave; returns the word count (tokens up to and including the first 0).  One load per
// lane and a ballot (every lane walking the row by itself was 16 dependent loads per caption).  No barrier inside.
__device__ __forceinline__ int load_words_wave(const int64_t* row, int L, int* tok, int lane) {
  const int64_t v = lane < L ? row[lane] : 1;
  const unsigned long long z = __ballot(lane < L && v == 0);
  const int w = z ? (int)__ffsll((long long)z) : L;
  if (lane < w) tok[lane] = (int)v;
  return w;
}
__device__ __forceinline__ int load_words(const int64_t* row, int L, int* tok, int lane) {
  const int w = load_words_wave(row, L, tok, lane);
  __syncthreads();
  return w;
}

// one wave: tf-idf weights of a caption's n-grams, kept at the first occurrence (no barrier inside)
__device__ __forceinline__ void cook(const CiderParams& p, const int* tok, int W, double* w, unsigned char* first, int lane) {
  for (int g = lane; g < NG * MAXW; g += 64) {
    const int k = g / MAXW, i = g - k * MAXW;      // order k + 1
    double wt = 0.0;
    unsigned char f = 0;
    if (i + k < W) {
      int tf = 0;
      f = 1;
      for (int j = 0; j + k < W; ++j) {
        bool m = true;
        for (int q = 0; q <= k; ++q) m = m && tok[i + q] == tok[j + q];
        tf += m;
        if (m && j < i) f = 0;
      }
      if (f) {
        const double logdf = table_lookup(p, tok[i], k >= 1 ? tok[i + 1] : -1, k >= 2 ? tok[i + 2] : -1, k >= 3 ? tok[i + 3] : -1);
        wt = (double)tf * (p.ref_len - logdf);
      }
    }
    w[g] = wt;
    first[g] = f;
  }
}

__device__ __forceinline__ void norms(const double* w, const unsigned char* first, int W, double* norm, int lane) {
  if (lane < NG) {
    const int k = lane;
    double s = 0.0;
    for (int i = 0; i + k < W; ++i)
      if (first[k * MAXW + i]) s += w[k * MAXW + i] * w[k * MAXW + i];
    norm[k] = sqrt(s);
  }
}

// Six waves per hypothesis: wave 0 cooks the hypothesis while waves 1..5 cook one reference each (images with more than five
// references: rounds of five); then waves 1..5 match their reference against the hypothesis, and one lane per order adds the
// references' values in reference order -- the same f64 operations in the same order as with one wave walking hypothesis and
// references one after the other (263 us per launch then: a latency chain of 6 cooks + 5 matches), so scores are unchanged to the bit.
constexpr int CW = 6, RPR = CW - 1;

__global__ __launch_bounds__(64 * CW) void ciderd_kernel(const CiderParams p) {
  __shared__ int tok[CW][MAXW];
  __shared__ double w[CW][NG * MAXW], contrib[CW][NG * MAXW];
  __shared__ unsigned char first[CW][NG * MAXW];
  __shared__ double norm[CW][NG], vals[RPR][NG], score[NG];
  __shared__ int Wd[CW];
  const int h = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int img = (h % p.batch_size) / p.seq_per_img;
  const int r0 = p.ref_start[img], r1 = p.ref_start[img + 1];
  if (wave == 0 && lane < NG) score[lane] = 0.0;
  for (int base = r0, round = 0;; base += RPR, ++round) {
    const int r = base + wave - 1;
    const bool hyp_turn = wave == 0 && round == 0, ref_turn = wave >= 1 && r < r1;
    int W = 0;
    if (hyp_turn) W = load_words_wave(p.hyp + (size_t)h * p.L, p.L, tok[0], lane);
    if (ref_turn) W = load_words_wave(p.ref_tok + (size_t)r * p.Lr, p.Lr, tok[wave], lane);
    if ((hyp_turn || ref_turn) && lane == 0) Wd[wave] = W;
    __syncthreads();
    if (hyp_turn || ref_turn) cook(p, tok[wave], W, w[wave], first[wave], lane);
    __syncthreads();
    if (hyp_turn || ref_turn) norms(w[wave], first[wave], W, norm[wave], lane);
    __syncthreads();
    const int Wh = Wd[0];
    if (ref_turn) {
      for (int g = lane; g < NG * MAXW; g += 64) {
        const int k = g / MAXW, i = g - k * MAXW;
        double c = 0.0;
        if (first[0][g]) {
          double wr = 0.0;
          for (int j = 0; j + k < W; ++j) {
            if (!first[wave][k * MAXW + j]) continue;
            bool m = true;
            for (int q = 0; q <= k; ++q) m = m && tok[0][i + q] == tok[wave][j + q];
            if (m) { wr = w[wave][k * MAXW + j]; break; }
          }
          c = fmin(w[0][g], wr) * wr;
        }
        contrib[wave][g] = c;
      }
    }
    __syncthreads();
    if (ref_turn && lane < NG) {
      const int k = lane;
      double val = 0.0;
      for (int i = 0; i + k < Wh; ++i)
        if (first[0][k * MAXW + i]) val += contrib[wave][k * MAXW + i];
      if (norm[0][k] != 0.0 && norm[wave][k] != 0.0) val /= (norm[0][k] * norm[wave][k]);
      const int len_h = Wh >= 2 ? Wh - 1 : 0, len_r = W >= 2 ? W - 1 : 0;
      val *= p.penalty[len_h - len_r + p.pen_half];
      vals[wave - 1][k] = val;
    }
    __syncthreads();
    if (wave == 0 && lane < NG) {
      const int n = r1 - base < RPR ? r1 - base : RPR;
      for (int j = 0; j < n; ++j) score[lane] += vals[j][lane];
    }
    if (base + RPR >= r1) break;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double avg = ((score[0] + score[1]) + score[2]) + score[3];
    avg /= 4.0;
    avg /= (double)(r1 - r0);
    avg *= 10.0;
    p.scores[h] = avg;
  }
}

// ---- per-sentence BLEU-4 (the bleu_reward_weight half of the reward): BleuScorer.compute_score(option='closest'),
// P/AI_Challenger/Evaluation/caption_eval/coco_caption/pycxevalcap/bleu/bleu_scorer.py:23-88,199-240.  One wave per
// hypothesis; all counting in integers, the final products / pow / exp in f64 in the scorer's order.
struct BleuParams {
  const int64_t* hyp; int L, batch_size, seq_per_img;
  const int64_t* ref_tok; int Lr; const int32_t* ref_start;
  double* scores;
};

__global__ __launch_bounds__(64) void bleu_kernel(const BleuParams p) {
  __shared__ int tok_h[MAXW], tok_r[MAXW];
  __shared__ int cnt_h[NG * MAXW], ref_max[NG * MAXW];
  __shared__ unsigned char first_h[NG * MAXW];
  __shared__ int correct[NG];
  const int h = blockIdx.x, lane = threadIdx.x;
  const int img = (h % p.batch_size) / p.seq_per_img;
  const int Wh = load_words(p.hyp + (size_t)h * p.L, p.L, tok_h, lane);
  for (int g = lane; g < NG * MAXW; g += 64) {       // precook(test): count of every n-gram, flagged at its first position
    const int k = g / MAXW, i = g - k * MAXW;
    int c = 0;
    unsigned char f = 0;
    if (i + k < Wh) {
      f = 1;
      for (int j = 0; j + k < Wh; ++j) {
        bool m = true;
        for (int q = 0; q <= k; ++q) m = m && tok_h[i + q] == tok_h[j + q];
        c += m;
        if (m && j < i) f = 0;
      }
    }
    cnt_h[g] = c; first_h[g] = f; ref_max[g] = 0;
  }
  int best_diff = 1 << 30, reflen = 0;               // min((abs(l - testlen), l) for l in reflens)[1]  (:76)
  const int r0 = p.ref_start[img], r1 = p.ref_start[img + 1];
  for (int r = r0; r < r1; ++r) {
    __syncthreads();
    const int Wr = load_words(p.ref_tok + (size_t)r * p.Lr, p.Lr, tok_r, lane);
    const int diff = Wr > Wh ? Wr - Wh : Wh - Wr;
    if (diff < best_diff || (diff == best_diff && Wr < reflen)) { best_diff = diff; reflen = Wr; }
    for (int g = lane; g < NG * MAXW; g += 64) {     // cook_refs: the most often any one reference holds the n-gram
      if (!first_h[g]) continue;
      const int k = g / MAXW, i = g - k * MAXW;
      int c = 0;
      for (int j = 0; j + k < Wr; ++j) {
        bool m = true;
        for (int q = 0; q <= k; ++q) m = m && tok_h[i + q] == tok_r[j + q];
        c += m;
      }
      if (c > ref_max[g]) ref_max[g] = c;
    }
  }
  __syncthreads();
  if (lane < NG) {                                   // cook_test: clipped matches per order
    int c = 0;
    for (int i = 0; i + lane < Wh; ++i)
      if (first_h[lane * MAXW + i]) c += min(ref_max[lane * MAXW + i], cnt_h[lane * MAXW + i]);
    correct[lane] = c;
  }
  __syncthreads();
  if (lane == 0) {
    const double small = 1e-9, tiny = 1e-15;
    double bleu = 1.0;
    for (int k = 0; k < NG; ++k) {
      const int guess = Wh - k > 0 ? Wh - k : 0;
      bleu *= ((double)correct[k] + tiny) / ((double)guess + small);
    }
    double b4 = pow(bleu, 1.0 / NG);
    const double ratio = ((double)Wh + tiny) / ((double)reflen + small);
    if (ratio < 1.0) b4 *= exp(1.0 - 1.0 / ratio);
    p.scores[h] = b4;
  }
}

__global__ void reward_kernel(const double* scores, int N, int L, float weight, float* reward) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * L) return;
  const int n = i / L;
  reward[i] = (float)((double)weight * scores[n] - (double)weight * scores[N + n]);
}

}  // namespace

extern "C" {

int64_t uic_ciderd_table_slots(int64_t n_entries) {
  int64_t s = 16;
  while (s < 2 * n_entries) s <<= 1;
  return s;
}

int uic_ciderd_table_build(const int32_t* keys, const double* values, int64_t n, int32_t* slot_keys, double* slot_vals,
                           int64_t slots) {
  UIC_REQUIRE(n >= 0 && (n == 0 || (keys && values)) && slot_keys && slot_vals, "ciderd_table_build: null pointer");
  UIC_REQUIRE(slots >= 16 && (slots & (slots - 1)) == 0 && slots >= 2 * n, "ciderd_table_build: slots=%lld must be a power of two >= 2 * %lld",
              (long long)slots, (long long)n);
  for (int64_t s = 0; s < slots; ++s) {
    slot_keys[s * 4] = slot_keys[s * 4 + 1] = slot_keys[s * 4 + 2] = slot_keys[s * 4 + 3] = -1;
    slot_vals[s] = 0.0;
  }
  const uint64_t mask = (uint64_t)slots - 1;
  for (int64_t e = 0; e < n; ++e) {
    const int32_t* k = keys + e * 4;
    UIC_REQUIRE(k[0] >= 0, "ciderd_table_build: entry %lld has a negative first token", (long long)e);
    uint64_t s = ngram_hash(k[0], k[1], k[2], k[3]) & mask;
    while (slot_keys[s * 4] >= 0 &&
           !(slot_keys[s * 4] == k[0] && slot_keys[s * 4 + 1] == k[1] && slot_keys[s * 4 + 2] == k[2] && slot_keys[s * 4 + 3] == k[3]))
      s = (s + 1) & mask;
    slot_keys[s * 4] = k[0]; slot_keys[s * 4 + 1] = k[1]; slot_keys[s * 4 + 2] = k[2]; slot_keys[s * 4 + 3] = k[3];
    slot_vals[s] = values[e];
  }
  return UIC_OK;
}

int uic_ciderd_scores(const int64_t* hyp, int32_t n_hyp, int32_t L, int32_t batch_size, int32_t seq_per_img,
                      const int64_t* ref_tok, int32_t Lr, const int32_t* ref_start, int32_t n_img,
                      const int32_t* slot_keys, const double* slot_vals, int64_t slots, double ref_len,
                      const double* penalty, int32_t pen_half, double* scores, void* stream) {
  UIC_REQUIRE(hyp && ref_tok && ref_start && penalty && scores, "ciderd_scores: null pointer");
  UIC_REQUIRE(slots == 0 || (slot_keys && slot_vals && (slots & (slots - 1)) == 0), "ciderd_scores: bad table (slots=%lld)", (long long)slots);
  UIC_REQUIRE(L >= 1 && L <= MAXW && Lr >= 1 && Lr <= MAXW, "ciderd_scores: caption rows of %d / %d tokens (max %d)", L, Lr, MAXW);
  UIC_REQUIRE(batch_size >= 1 && seq_per_img >= 1 && batch_size % seq_per_img == 0 && batch_size / seq_per_img == n_img,
              "ciderd_scores: batch_size=%d seq_per_img=%d n_img=%d do not agree", batch_size, seq_per_img, n_img);
  UIC_REQUIRE(pen_half >= (L > Lr ? L : Lr), "ciderd_scores: penalty table covers |delta| <= %d, captions have up to %d words", pen_half, L > Lr ? L : Lr);
  if (n_hyp == 0) return UIC_OK;
  CiderParams p;
  p.hyp = hyp; p.n_hyp = n_hyp; p.L = L; p.batch_size = batch_size; p.seq_per_img = seq_per_img;
  p.ref_tok = ref_tok; p.Lr = Lr; p.ref_start = ref_start;
  p.slot_keys = slot_keys; p.slot_vals = slot_vals; p.slots = slots; p.ref_len = ref_len;
  p.penalty = penalty; p.pen_half = pen_half; p.scores = scores;
  hipLaunchKernelGGL(ciderd_kernel, dim3(n_hyp), dim3(64 * CW), 0, (hipStream_t)stream, p);
  UIC_LAUNCH_CHECK("ciderd_kernel");
  return UIC_OK;
}

int uic_bleu_scores(const int64_t* hyp, int32_t n_hyp, int32_t L, int32_t batch_size, int32_t seq_per_img,
                    const int64_t* ref_tok, int32_t Lr, const int32_t* ref_start, int32_t n_img, double* scores, void* stream) {
  UIC_REQUIRE(hyp && ref_tok && ref_start && scores, "bleu_scores: null pointer");
  UIC_REQUIRE(L >= 1 && L <= MAXW && Lr >= 1 && Lr <= MAXW, "bleu_scores: caption rows of %d / %d tokens (max %d)", L, Lr, MAXW);
  UIC_REQUIRE(n_hyp >= 1 && batch_size >= 1 && seq_per_img >= 1 && batch_size % seq_per_img == 0 && batch_size / seq_per_img == n_img,
              "bleu_scores: batch_size=%d seq_per_img=%d n_img=%d do not agree", batch_size, seq_per_img, n_img);
  BleuParams p{hyp, L, batch_size, seq_per_img, ref_tok, Lr, ref_start, scores};
  hipLaunchKernelGGL(bleu_kernel, dim3(n_hyp), dim3(64), 0, (hipStream_t)stream, p);
  UIC_LAUNCH_CHECK("bleu_scores");
  return UIC_OK;
}

int uic_ciderd_reward(const double* scores, int32_t N, int32_t L, float weight, float* reward, void* stream) {
  UIC_REQUIRE(scores && reward && N >= 0 && L >= 1, "ciderd_reward: bad arguments");
  if (N == 0) return UIC_OK;
  const int n = N * L;
  hipLaunchKernelGGL(reward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, scores, N, L, weight, reward);
  UIC_LAUNCH_CHECK("reward_kernel");
  return UIC_OK;
}

}  // extern "C"
