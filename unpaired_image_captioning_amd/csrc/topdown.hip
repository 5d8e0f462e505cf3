// Host-side sequencing of the TopDown captioner step on one MI355X: every launch of
// AttModel._forward / _sample, LanguageModelCriterion and their backward, enqueued on the
// caller's HIP stream with no host synchronisation.  (The fused training step also forks onto two library-owned side streams
// and joins them before returning.  Capturing it into a hipGraph is NOT supported: with every stream joined the capture itself
// goes through, but hipStreamEndCapture crashes inside the ROCm 7.2 runtime -- tools/graph_capture_probe.py; an un-joined stream
// is reported properly as hipErrorStreamCaptureUnjoined.  The step is GPU-bound with the host 2.5 ms ahead, so a graph would not
// shorten it.)
//
// Restructuring relative to the reference's per-step Python loop (P/models/AttModel.py:119-165):
//   * teacher forcing makes xt_t and fc' known up front, so their share of the att_lstm gate
//     GEMM is batched over all T steps (Gx, Gfc); only K = 2H stays in the recurrence;
//   * the logit GEMM, log_softmax and the criterion run once over all T*N rows, never
//     materialising [N,T,V1] log-probs unless the caller asks for them;
//   * in backward only dX GEMMs and the attention step stay in the BPTT loop; every weight
//     gradient is one GEMM over the T*N stacked rows afterwards, and the attention's [N,R,*]
//     gradients are produced by one deferred pass (attention.hip).
#include "uic_common.h"
#include <mutex>
#include "uic_host.h"
#include "../../include/uic_hip.h"
#include <string.h>
#include <stdlib.h>

namespace {

#ifndef UIC_WG_CHUNK
#define UIC_WG_CHUNK 4
#endif
constexpr int WG_CHUNK = UIC_WG_CHUNK;   // decode steps per hand-off between the two streams of the fused training step
#ifndef UIC_BPTT_SPLIT
#define UIC_BPTT_SPLIT 4
#endif
#ifndef UIC_BPTT_FUSE_CELL
#define UIC_BPTT_FUSE_CELL 1
#endif
#ifndef UIC_GFC_SEPARATE
#define UIC_GFC_SEPARATE 1
#endif
constexpr int BPTT_SPLIT = UIC_BPTT_SPLIT;   // K slices of the BPTT loop's d x GEMMs (Step::bptt_split)

struct Layout {
  // forward activations
  void* fcT; void* attT; int* row_len;
  float* ypre; float* amask_rep; int* row_len_rep;   // seq_per_img > 1: per-image relu(att_embed) (f32), att_masks / region counts per caption row
  void* fcp; void* attp; void* patt;
  void* eatt;                                  // bf16: e^{2 p_att}, what the persistent training recurrence's attention reads (rnn_persist.hip)
  void* ybn; float* bn_stat0; float* bn_stat4; float* bn_part; float* bn_red;   // use_bn: pre-BN4 activations, {mean, rstd}, scratch
  void* xt_all; float* gx; float* gfc;
  int64_t* tok_used;   // [N, T] inputs actually fed to the embedding (differs from labels under scheduled sampling)
  void* h_att; void* h_lang; float* c_att; float* c_lang;   // [(T+1), N, H]
  void* gates1; void* gates2;
  float* atth_all; float* alpha_all; void* ctx_all; void* hdrop_all;
  float* logits; void* dlogits; float* row_loss; float* scalars;
  int* cap_len;                                 // ... [N]: 1 + the last decode step at which the row has a live position (the BPTT steps behind it are zeros)
  int* live_inv;                                // ... and its inverse [T N]: position -> list index or -1 (the recurrence stores hdrop rows by it)
  int* live_map;                                // the live list made on the device (uic_topdown_batch.live_rows == NULL): [T N + 128]
  void* hc; float* dhc;                         // live-position logit layer (uic_topdown_batch.live_rows): the live rows of hdrop, operand dtype
                                                // [T N + 128, H]; their d hdrop, f32 [T N, H]
  void* lh[UIC_MAX_LOGIT_LAYERS - 1];       // logit_layers > 1: dropout(relu(hidden logit block l)) [T*N, H]
  void* dlh_pre[UIC_MAX_LOGIT_LAYERS - 1];  // its pre-activation gradient [T*N, H] (operand dtype), kept for the weight gradient
  float* dlh;                               // gradient flowing between the logit blocks [T*N, H] f32
  void* s_lh[UIC_MAX_LOGIT_LAYERS - 1];     // the same activations of one decode step [N, H]
  // backward
  float* dhdrop; float* dx2_all; float* dx1; float* dc_att; float* dc_lang;
  void* dg1_all; void* dg2_all; float* de_all; void* datth_all;
  void* dgfc; float* dfcp; void* dfcpre; float* dxt;
  float* d_att; void* d_patt; float* dwalpha_part; void* d_pre;
  void* tA; void* tB; float* colscratch; size_t colscratch_floats; float* small;
  float* slab; size_t slab_bytes;
  void* tLA; void* tLB; float* colscratchL;   // scratch of the logit-layer weight gradients (side stream)
  void* tSA; void* tSB; float* slab2;         // scratch of the per-chunk recurrent weight gradients (side stream)
  void* tTA; void* tTB; float* slab3;         // the same for the third stream (att_lstm / h2att share of a chunk)
  void* tUA; void* tUB; float* slab4;         // ... and the fourth (default order: the chunks' shares run on streams 3 and 4)
  int* embed_scratch;                         // uic_embed_bwd_sorted_launch
  void* fcwT; void* attwT;                     // fc_embed / att_embed weights transposed ([Dfc, H], [D, H]): only for the optional input-feature gradients
  void* ones_blk; size_t ones_rows;           // [max(WG_CHUNK * N, N * R), 128] operand dtype, all ones: the "input" whose weight gradient is the bias gradient
  unsigned* rnn_sync; unsigned long long* rnn_dbg;   // persistent recurrence (rnn_persist.hip): sync block, optional time stamps
  float* bp_slab2[2]; float* bp_slab1;       // split-K partial slabs of the BPTT loop's d x2 (two generations) and d x1 GEMMs: [BPTT_SPLIT][N, 3H] / [N, 2H]
  unsigned* rnn_bwd_sync; size_t rnn_bwd_sync_bytes; unsigned long long* rnn_bwd_dbg;   // persistent BPTT (rnn_bwd_persist.hip): one sync block per launch of a step
  // sampling
  void* s_h_att[2]; void* s_h_lang[2]; float* s_c_att[2]; float* s_c_lang[2];
  void* s_xt; float* s_atth; float* s_alpha; void* s_ctx; void* s_hdrop; float* s_logits;
  int64_t* s_it; int* s_unf; int* s_nunf;
  float* dec_part; int* dec_tok; void* dec_embed_relu;   // persistent decode (rnn_persist.hip): per-workgroup logit partials, exchanged tokens, relu(embed) in bf16
  // beam search bookkeeping (rows = (image, beam); sizes depend on N and T only)
  float* bm_cand_val; int* bm_cand_idx; int64_t* bm_seq[2]; float* bm_lp[2]; float* bm_sum; int* bm_parent;
  int* bm_done_count; float* bm_done_p; int64_t* bm_done_seq; float* bm_done_lp;
  size_t total;
};

Layout make_layout(const uic_topdown_dims& d, void* ws) {
  Layout L;
  memset(&L, 0, sizeof(L));
  Bump b{(char*)ws, 0};
  const size_t S = uic_dtype_size(d.dtype);
  const size_t N = d.N, R = d.R, D = d.D, Dfc = d.Dfc, H = d.H, E = d.E, A = d.A, V1 = d.V1, T = d.T;
  const size_t M = T * N, Mp = rup8(M), Np = rup8(N), NR = N * R, NRp = rup8(NR), V1p = vpad(V1);
  L.fcT = b.take(N * Dfc * S);
  L.attT = b.take(NR * D * S);
  L.row_len = (int*)b.take(N * sizeof(int));
  if (d.seq_per_img > 1) {
    L.amask_rep = (float*)b.take(NR * 4);
    L.row_len_rep = (int*)b.take(N * sizeof(int));
    L.ypre = (float*)b.take(N / d.seq_per_img * R * H * 4);
  }
  L.fcp = b.take(N * H * S);
  L.attp = b.take(NR * H * S);
  L.patt = b.take(NR * A * S);
  L.eatt = d.dtype == UIC_BF16 ? b.take(NR * A * S) : nullptr;
  if (d.use_bn) {
    L.bn_stat0 = (float*)b.take(2 * D * 4);
    L.bn_stat4 = (float*)b.take(2 * H * 4);
    const size_t p0 = uic_bn_scratch_floats((int)NR, (int)D), p4 = uic_bn_scratch_floats((int)NR, (int)H);
    L.bn_part = (float*)b.take((p0 > p4 ? p0 : p4) * 4);
    L.bn_red = (float*)b.take(3 * (D > H ? D : H) * 4);
    if (d.use_bn == 2) L.ybn = b.take(NR * H * S);
  }
  L.tok_used = (int64_t*)b.take(N * T * 8);
  L.xt_all = b.take(M * E * S);
  L.gx = (float*)b.take(M * 4 * H * 4);
  L.gfc = (float*)b.take(N * 4 * H * 4);
  L.h_att = b.take((T + 1) * N * H * S);
  L.h_lang = b.take((T + 1) * N * H * S);
  L.c_att = (float*)b.take((T + 1) * N * H * 4);
  L.c_lang = (float*)b.take((T + 1) * N * H * 4);
  L.gates1 = b.take(M * 4 * H * S);
  L.gates2 = b.take(M * 4 * H * S);
  L.atth_all = (float*)b.take(M * A * 4);
  L.alpha_all = (float*)b.take(M * R * 4);
  L.ctx_all = b.take(M * H * S);
  L.hdrop_all = b.take(M * H * S);
  // (+128 rows: the live-position list is padded to whole 128-row tiles, uic_topdown_batch.live_rows)
  L.logits = (float*)b.take((M + 128) * V1p * 4);
  L.dlogits = b.take((M + 128) * V1p * S);
  L.row_loss = (float*)b.take(M * 4);
  for (int l = 0; l + 1 < d.logit_layers; ++l) {
    L.lh[l] = b.take(M * H * S);
    L.dlh_pre[l] = b.take(M * H * S);
    L.s_lh[l] = b.take(N * H * S);
  }
  if (d.logit_layers > 1) L.dlh = (float*)b.take(M * H * 4);
  L.scalars = (float*)b.take(64);
  L.dhdrop = (float*)b.take(M * H * 4);
  L.live_map = (int*)b.take((M + 128) * 4);
  L.live_inv = (int*)b.take(M * 4);
  L.cap_len = (int*)b.take(N * 4);
  L.hc = b.take((M + 128) * H * S);
  L.dhc = (float*)b.take((M + 128) * H * 4);
  L.dx2_all = (float*)b.take(M * 3 * H * 4);
  L.dx1 = (float*)b.take(N * 2 * H * 4);
  L.dc_att = (float*)b.take(N * H * 4);
  L.dc_lang = (float*)b.take(N * H * 4);
  L.dg1_all = b.take(M * 4 * H * S);
  L.dg2_all = b.take(M * 4 * H * S);
  L.de_all = (float*)b.take(M * R * 4);
  L.datth_all = b.take(M * A * S);
  L.dgfc = b.take(N * 4 * H * S);
  L.dfcp = (float*)b.take(N * H * 4);
  L.dfcpre = b.take(N * H * S);
  L.dxt = (float*)b.take(M * E * 4);
  L.d_att = (float*)b.take(NR * H * 4);
  L.d_patt = b.take(NR * A * S);
  L.dwalpha_part = (float*)b.take(N * (A + 1) * 4);
  L.d_pre = b.take(NR * H * S);
  size_t ta = 4 * H * Mp;
  if (A * Mp > ta) ta = A * Mp;
  if (A * NRp > ta) ta = A * NRp;
  if (H * NRp > ta) ta = H * NRp;
  if (4 * H * Np > ta) ta = 4 * H * Np;
  size_t tb = (2 * H + (E > H ? E : H)) * Mp;   // three stacked right operands of the merged LSTM weight-gradient GEMMs
  if ((H > D ? H : D) * NRp > tb) tb = (H > D ? H : D) * NRp;
  if ((H > Dfc ? H : Dfc) * Np > tb) tb = (H > Dfc ? H : Dfc) * Np;
  tb += 128 * (NRp > Mp ? NRp : Mp);            // + the ones segment of the bias gradients
  L.tA = b.take(ta * S);
  L.tB = b.take(tb * S);
  size_t maxcols = V1p;
  if (4 * H > maxcols) maxcols = 4 * H;
  if (A + 1 > maxcols) maxcols = A + 1;
  L.colscratch_floats = 128 * maxcols;
  L.colscratch = (float*)b.take(L.colscratch_floats * 4);
  L.small = (float*)b.take((A + 8) * 4);
  L.tLA = b.take((d.logit_layers > 1 && H > V1 ? H : V1) * (Mp + 128) * S);   // dlogits^T [V1, M]; hidden logit blocks: d pre^T [H, M]
  L.tLB = b.take(H * (Mp + 128) * S);
  L.colscratchL = (float*)b.take(L.colscratch_floats * 4);
  {
    const size_t Kc = rup8((size_t)WG_CHUNK * N);
    size_t rb = 2 * H + E + 128;                // att_lstm inputs [h_lang | xt | h_att]; lang_lstm inputs are 3H wide; + the ones block
    if (3 * H + 128 > rb) rb = 3 * H + 128;
    if (Dfc > rb) rb = Dfc;
    const size_t la = 4 * H > A ? 4 * H : A;    // left operands: dG [rows, 4H] of the LSTMs, d att_h [rows, A] of h2att
    L.tSA = b.take(la * Kc * S);
    L.tSB = b.take(rb * Kc * S);
    L.tTA = b.take(la * Kc * S);
    L.tTB = b.take(rb * Kc * S);
    L.tUA = b.take(la * Kc * S);
    L.tUB = b.take(rb * Kc * S);
  }
  {  // split-K partial slabs: room for 4 slices of the largest merged weight gradient [4H, 2H + E]
    size_t sl = 4 * (4 * H) * (2 * H + E) * 4;
    const size_t cap = (size_t)256 << 20;
    if (sl > cap) sl = cap;
    L.slab_bytes = sl;
    L.slab = (float*)b.take(sl);
    L.slab2 = (float*)b.take(sl);
    L.slab3 = (float*)b.take(sl);
    L.slab4 = (float*)b.take(sl);
  }
  L.embed_scratch = (int*)b.take(uic_embed_bwd_sorted_scratch_ints(N, T, V1, E) * 4);
  L.fcwT = b.take(Dfc * H * S);
  L.attwT = b.take(D * H * S);
  L.ones_rows = rup8((size_t)WG_CHUNK * N) > (size_t)N * R ? rup8((size_t)WG_CHUNK * N) : (size_t)N * R;
  L.ones_blk = b.take(L.ones_rows * 128 * S);
  L.rnn_sync = (unsigned*)b.take(uic_rnn_persist_sync_bytes());
  L.rnn_dbg = (unsigned long long*)b.take((size_t)256 * T * 16 * 8);
  for (int i = 0; i < 2; ++i) L.bp_slab2[i] = (float*)b.take((size_t)BPTT_SPLIT * N * 3 * H * 4);
  L.bp_slab1 = (float*)b.take((size_t)BPTT_SPLIT * N * 2 * H * 4);
  L.rnn_bwd_sync_bytes = T * uic_rnn_persist_sync_bytes();
  L.rnn_bwd_sync = (unsigned*)b.take(L.rnn_bwd_sync_bytes);
  L.rnn_bwd_dbg = (unsigned long long*)b.take((size_t)256 * T * 16 * 8);
  for (int i = 0; i < 2; ++i) {
    L.s_h_att[i] = b.take(N * H * S);
    L.s_h_lang[i] = b.take(N * H * S);
    L.s_c_att[i] = (float*)b.take(N * H * 4);
    L.s_c_lang[i] = (float*)b.take(N * H * 4);
  }
  L.s_xt = b.take(N * E * S);
  L.s_atth = (float*)b.take(N * A * 4);
  L.s_alpha = (float*)b.take(N * R * 4);
  L.s_ctx = b.take(N * H * S);
  L.s_hdrop = b.take(N * H * S);
  L.s_logits = (float*)b.take(N * V1p * 4);
  L.s_it = (int64_t*)b.take(N * 8);
  L.s_unf = (int*)b.take(N * 4);
  L.s_nunf = (int*)b.take((T + 2) * UIC_NUNF_STRIPES * 4);
  L.dec_part = (float*)b.take(uic_rnn_decode_part_floats((int)N) * 4);
  L.dec_tok = (int*)b.take(N * 4);
  L.dec_embed_relu = b.take(V1 * E * 2);
  L.bm_cand_val = (float*)b.take(N * UIC_BEAM_MAX * 4);
  L.bm_cand_idx = (int*)b.take(N * UIC_BEAM_MAX * 4);
  for (int i = 0; i < 2; ++i) {
    L.bm_seq[i] = (int64_t*)b.take(N * T * 8);
    L.bm_lp[i] = (float*)b.take(N * T * 4);
  }
  L.bm_sum = (float*)b.take(N * 4);
  L.bm_parent = (int*)b.take(N * 4);
  L.bm_done_count = (int*)b.take(N * 4);
  L.bm_done_p = (float*)b.take(N * T * 4);
  L.bm_done_seq = (int64_t*)b.take(N * T * T * 8);
  L.bm_done_lp = (float*)b.take(N * T * T * 4);
  L.total = (b.off + 255) & ~(size_t)255;
  return L;
}

// operand-dtype and transposed weight copies
struct Derived {
  const void* embed_w;   // [V1, E] the embedding table in the operand dtype (the f32 master itself in f32 runs)
  const void* fc_w; const void* att_w; const void* ctx2att_w; const void* logit_w;
  const void* att_w_ih; const void* att_w_hh; const void* lang_w_ih; const void* lang_w_hh; const void* h2att_w;
  void* logit_wT;    // [H, V1p]
  void* w2T;         // [3H, 4H] = [lang_w_ih^T ; lang_w_hh^T]
  void* w1recT;      // [2H, 4H] = [att_w_ih[:, 0:H]^T ; att_w_hh^T]
  void* wxT;         // [E, 4H]  = att_w_ih[:, 2H:]^T
  void* wfcpT;       // [H, 4H]  = att_w_ih[:, H:2H]^T
  void* h2attT;      // [H, A]
  void* ctx2attT;    // [H, A]
  float* att_beff;   // use_bn: b' = att_b + att_w bn0_beta  [H]  (att_w then points at W' = att_w diag(bn0_gamma))
  const void* logit_h_w[UIC_MAX_LOGIT_LAYERS - 1];   // hidden logit blocks [H, H]
  void* logit_h_wT[UIC_MAX_LOGIT_LAYERS - 1];        // their transposes for the dX GEMMs
  size_t total;
};

Derived make_derived(const uic_topdown_dims& d, const uic_topdown_weights* w, void* base) {
  Derived v;
  memset(&v, 0, sizeof(v));
  Bump b{(char*)base, 0};
  const size_t S = uic_dtype_size(d.dtype);
  const size_t D = d.D, Dfc = d.Dfc, H = d.H, E = d.E, A = d.A, V1 = d.V1, V1p = vpad(V1);
  const bool bf = d.dtype == UIC_BF16;
  auto copy = [&](const float* master, size_t n) -> const void* {
    if (!bf) return master;
    return b.take(n * S);
  };
  v.fc_w = copy(w ? w->fc_w : nullptr, H * Dfc);
  if (d.use_bn) {
    v.att_w = b.take(H * D * S);
    v.att_beff = (float*)b.take(H * 4);
  } else {
    v.att_w = copy(w ? w->att_w : nullptr, H * D);
  }
  v.ctx2att_w = copy(w ? w->ctx2att_w : nullptr, A * H);
  v.logit_w = copy(w ? w->logit_w : nullptr, V1 * H);
  v.att_w_ih = copy(w ? w->att_lstm_w_ih : nullptr, 4 * H * (E + 2 * H));
  v.att_w_hh = copy(w ? w->att_lstm_w_hh : nullptr, 4 * H * H);
  v.lang_w_ih = copy(w ? w->lang_lstm_w_ih : nullptr, 4 * H * 2 * H);
  v.lang_w_hh = copy(w ? w->lang_lstm_w_hh : nullptr, 4 * H * H);
  v.h2att_w = copy(w ? w->h2att_w : nullptr, A * H);
  v.embed_w = copy(w ? w->embed_w : nullptr, V1 * E);
  v.logit_wT = b.take(H * V1p * S);
  v.w2T = b.take(3 * H * 4 * H * S);
  v.w1recT = b.take(2 * H * 4 * H * S);
  v.wxT = b.take(E * 4 * H * S);
  v.wfcpT = b.take(H * 4 * H * S);
  v.h2attT = b.take(H * A * S);
  v.ctx2attT = b.take(H * A * S);
  for (int l = 0; l + 1 < d.logit_layers; ++l) {
    v.logit_h_w[l] = copy(w ? w->logit_h_w[l] : nullptr, H * H);
    v.logit_h_wT[l] = b.take(H * H * S);
  }
  v.total = (b.off + 255) & ~(size_t)255;
  return v;
}

int check_dims(const uic_topdown_dims* d) {
  UIC_REQUIRE(d != nullptr, "null dims");
  UIC_REQUIRE(d->dtype == UIC_F32 || d->dtype == UIC_BF16, "bad dtype %d", d->dtype);
  UIC_REQUIRE(d->N > 0 && d->R > 0 && d->T > 0 && d->V1 > 1, "bad sizes N=%d R=%d T=%d V1=%d", d->N, d->R, d->T, d->V1);
  UIC_REQUIRE(d->D % 8 == 0 && d->Dfc % 8 == 0 && d->H % 8 == 0 && d->E % 8 == 0 && d->A % 8 == 0,
              "D=%d Dfc=%d H=%d E=%d A=%d must all be multiples of 8", d->D, d->Dfc, d->H, d->E, d->A);
  UIC_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "drop_p=%f outside [0,1)", (double)d->drop_p);
  UIC_REQUIRE(d->use_bn >= 0 && d->use_bn <= 2, "use_bn=%d outside {0,1,2}", d->use_bn);
  UIC_REQUIRE(d->logit_layers >= 0 && d->logit_layers <= UIC_MAX_LOGIT_LAYERS, "logit_layers=%d outside [0,%d]", d->logit_layers,
              UIC_MAX_LOGIT_LAYERS);
  UIC_REQUIRE(d->seq_per_img >= 0 && (d->seq_per_img <= 1 || d->N % d->seq_per_img == 0),
              "seq_per_img=%d must divide N=%d", d->seq_per_img, d->N);
  return UIC_OK;
}

constexpr float BN_MOMENTUM = 0.1f, BN_EPS = 1e-5f;   // nn.BatchNorm1d defaults (AttModel.py:79,83)

__global__ void rowlen_kernel(const float* mask, int N, int R, int* out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int c = 0;
  for (int r = 0; r < R; ++r) c += mask[(size_t)n * R + r] != 0.f;
  out[n] = c;
}

// _prepare_feature (P/models/AttModel.py:107-117) + operand casts.  part: 0 = everything; 1 = only the fc_embed branch
// (fc' = dropout(relu(fc_embed(fc)))), 2 = only the att_embed / ctx2att branch -- the two are independent, the fused
// training step runs them on its two streams.
int prepare_features(const uic_topdown_dims& d, const uic_topdown_weights* w, const Derived& dv, const uic_topdown_batch* b,
                     const Layout& L, int training, float drop_p, unsigned seed, const void** fc_in_out, const void** att_in_out,
                     hipStream_t s, int part = 0) {
  const int dt = d.dtype;
  const int N = d.N, R = d.R, H = d.H, A = d.A;
  const bool bn_train = (training & 1) != 0, bn_update = bn_train && !(training & 2);
  // seq_per_img = S > 1: fc_feats / att_feats / att_masks hold one row per IMAGE; caption row n belongs to image n / S
  const int S = d.seq_per_img > 1 ? d.seq_per_img : 1, Ni = N / S;
  const bool fold = S > 1;                  // att_embed's [BatchNorm +] Linear runs once per image; dropout is applied per caption row
  const float* att_src = b->att_feats;
  const float* amask = b->att_masks;        // per caption row
  const int Na = fold ? Ni : N;             // rows of att_src
  const int* row_len = b->att_masks ? L.row_len : nullptr;          // per row of att_src
  const int* row_len_cap = row_len;                                 // per caption row
  const bool do_fc = part != 2, do_att = part != 1;
  if (b->att_masks && !do_att) {       // (pointers only)
    if (fold) { amask = L.amask_rep; row_len_cap = L.row_len_rep; }
  } else if (b->att_masks) {
    hipLaunchKernelGGL(rowlen_kernel, dim3((Na + 255) / 256), dim3(256), 0, s, b->att_masks, Na, R, L.row_len);
    UIC_LAUNCH_CHECK("rowlen_kernel");
    if (fold) {
      UIC_TRY(uic_expand_rows_launch(UIC_F32, b->att_masks, L.amask_rep, Ni, S, (size_t)R, s));
      amask = L.amask_rep;
      hipLaunchKernelGGL(rowlen_kernel, dim3((N + 255) / 256), dim3(256), 0, s, amask, N, R, L.row_len_rep);
      UIC_LAUNCH_CHECK("rowlen_kernel");
      row_len_cap = L.row_len_rep;
    }
  }
  const void* fc_in = b->fc_feats;
  const void* att_in = att_src;
  if (S > 1) {
    if (do_fc) UIC_TRY(uic_expand_rows_launch(dt, b->fc_feats, L.fcT, Ni, S, (size_t)d.Dfc, s));
    fc_in = L.fcT;
  } else if (dt == UIC_BF16) {
    if (do_fc) UIC_TRY(uic_cast_f32_launch(dt, b->fc_feats, L.fcT, (size_t)N * d.Dfc, s));
    fc_in = L.fcT;
  }
  if (d.use_bn || dt == UIC_BF16) att_in = L.attT;
  // bf16 without BatchNorm in front: att_embed's GEMM takes the f32 features as they are, rounds them on their way to LDS and
  // leaves the bf16 image in L.attT for the weight gradient (gemm_pp.hip, f32 A operand) -- no separate 283 MB cast pass
  bool att_f32a = false;
  if (do_att && !d.use_bn && dt == UIC_BF16) {
    UicGemmParams g = gemm_base(dt, Na * R, H);
    add_seg(g, att_src, d.D, dv.att_w, d.D, d.D);
    g.C = L.attp; g.ldc = H; g.a_f32 = 1; g.a_copy = L.attT; g.ld_a_copy = d.D;
    att_f32a = uic_gemm_pp_eligible(g) && (size_t)Na * R >= 2048 && !(d.recurrence & UIC_REC_NO_F32A);
  }
  if (do_att && d.use_bn) {
    // BatchNorm1d(D) over the packed live regions; xhat goes to the GEMM, the affine part lives in W' / b'.  Features given
    // once per image stand for S identical caption rows each: same mean / biased variance, `rep` fixes the unbiased one.
    if (bn_train)
      UIC_TRY(uic_bn_stats_launch(UIC_F32, att_src, Na * R, R, d.D, row_len, L.bn_part, BN_MOMENTUM, BN_EPS, L.bn_stat0,
                                  bn_update ? w->att_bn0_rm : nullptr, bn_update ? w->att_bn0_rv : nullptr, s, (float)S));
    else
      UIC_TRY(uic_bn_stats_running_launch(w->att_bn0_rm, w->att_bn0_rv, d.D, BN_EPS, L.bn_stat0, s));
    UIC_TRY(uic_bn_apply_launch(UIC_F32, dt, att_src, Na * R, R, d.D, row_len, L.bn_stat0, nullptr, nullptr, 0, L.attT, s));
  } else if (do_att && dt == UIC_BF16 && !att_f32a) {
    UIC_TRY(uic_cast_f32_launch(dt, att_src, L.attT, (size_t)Na * R * d.D, s));
  }
  *fc_in_out = fc_in;
  *att_in_out = att_in;
  if (do_fc) {
    UicGemmParams g = gemm_base(dt, N, H);
    add_seg(g, fc_in, d.Dfc, dv.fc_w, d.Dfc, d.Dfc);
    g.C = L.fcp; g.ldc = H; g.bias = w->fc_b; g.flags = UIC_GEMM_RELU;
    g.drop_p = drop_p; g.seed = seed; g.site = UIC_SITE_FC;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  if (!do_att) return UIC_OK;
  void* const att_out = d.use_bn == 2 ? L.ybn : L.attp;
  if (fold) {
    // relu(W att + b) once per image (f32), then S caption rows with their own dropout masks -- element for element
    // what the S-fold replicated GEMM epilogue writes
    UicGemmParams g = gemm_base(dt, Ni * R, H);
    add_seg(g, att_in, d.D, dv.att_w, d.D, d.D);
    if (att_f32a) { g.seg[0].A = att_src; g.a_f32 = 1; g.a_copy = L.attT; g.ld_a_copy = d.D; }
    g.C = L.ypre; g.ldc = H; g.bias = d.use_bn ? dv.att_beff : w->att_b; g.flags = UIC_GEMM_RELU | UIC_GEMM_OUT_F32;
    if (b->att_masks) { g.row_len = L.row_len; g.R = R; }
    UIC_TRY(uic_gemm_launch(g, s));
    UIC_TRY(uic_expand_drop_launch(dt, L.ypre, att_out, Ni, S, (size_t)R * H, drop_p, seed, UIC_SITE_ATT, s));
  } else {
    UicGemmParams g = gemm_base(dt, N * R, H);
    add_seg(g, att_in, d.D, dv.att_w, d.D, d.D);
    if (att_f32a) { g.seg[0].A = att_src; g.a_f32 = 1; g.a_copy = L.attT; g.ld_a_copy = d.D; }
    g.C = att_out; g.ldc = H; g.bias = d.use_bn ? dv.att_beff : w->att_b; g.flags = UIC_GEMM_RELU;
    if (b->att_masks) { g.row_len = L.row_len; g.R = R; }
    g.drop_p = drop_p; g.seed = seed; g.site = UIC_SITE_ATT;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  if (d.use_bn == 2) {   // BatchNorm1d(H) after the Dropout; padded regions stay zero (pad_unsort_packed_sequence)
    if (bn_train)
      UIC_TRY(uic_bn_stats_launch(dt, L.ybn, N * R, R, H, row_len_cap, L.bn_part, BN_MOMENTUM, BN_EPS, L.bn_stat4,
                                  bn_update ? w->att_bn4_rm : nullptr, bn_update ? w->att_bn4_rv : nullptr, s));
    else
      UIC_TRY(uic_bn_stats_running_launch(w->att_bn4_rm, w->att_bn4_rv, H, BN_EPS, L.bn_stat4, s));
    UIC_TRY(uic_bn_apply_launch(dt, dt, L.ybn, N * R, R, H, row_len_cap, L.bn_stat4, w->att_bn4_w, w->att_bn4_b, 1, L.attp, s));
  }
  {
    UicGemmParams g = gemm_base(dt, N * R, A);
    add_seg(g, L.attp, H, dv.ctx2att_w, H, H);
    g.C = L.patt; g.ldc = A; g.bias = w->ctx2att_b;
    // e^{2 p_att} for the persistent recurrence's attention (rnn_persist.hip): from the ping-pong kernel's epilogue where that
    // kernel takes the problem, else one element-wise pass -- the same values either way
    g.C_exp2 = L.eatt;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  return UIC_OK;
}

int attention_step(const uic_topdown_dims& d, const uic_topdown_weights* w, const Derived& dv, const uic_topdown_batch* b,
                   const Layout& L, const void* h_att, float* att_h, float* alpha, void* ctx, hipStream_t s) {
  UicGemmParams g = gemm_base(d.dtype, d.N, d.A);
  add_seg(g, h_att, d.H, dv.h2att_w, d.H, d.H);
  g.C = att_h; g.ldc = d.A; g.bias = w->h2att_b; g.flags = UIC_GEMM_OUT_F32;
  UIC_TRY(uic_gemm_launch(g, s));
  UicAttnParams a;
  memset(&a, 0, sizeof(a));
  a.dtype = d.dtype; a.N = d.N; a.R = d.R; a.A = d.A; a.H = d.H;
  a.att_h = att_h; a.p_att = L.patt; a.att = L.attp; a.w_alpha = w->alpha_w; a.b_alpha = w->alpha_b;
  a.mask = b->att_masks ? (d.seq_per_img > 1 ? L.amask_rep : b->att_masks) : nullptr; a.ldmask = d.R; a.alpha = alpha; a.ctx = ctx; a.ldctx = d.H;
  return uic_attention_fwd_launch(a, s);
}

// Side stream + events for the fused training step: the logit layer of finished decode steps (logit GEMM,
// log-softmax/criterion, dH GEMM, later dW_logit) runs beside the latency-bound recurrence on a second HIP stream.
constexpr int MAX_CHUNKS = 64;
struct SideStream {
  hipStream_t stream = nullptr;
  hipStream_t stream3 = nullptr;    // a chunk's att_lstm / h2att weight gradients beside its lang_lstm ones (independent GEMMs)
  hipEvent_t ev_s3 = nullptr;       // stream3 -> side: that share of every chunk so far is done
  hipStream_t stream4 = nullptr;    // default order: lang_lstm share of a chunk on stream3, att_lstm / h2att share on stream4
  hipEvent_t ev_s4 = nullptr;       // stream4 -> stream3: its share of every chunk is done
  hipEvent_t ev_prep = nullptr;     // side -> stream3: the embedding gradient's token bucketing is done
  hipEvent_t ev_den = nullptr, ev_done = nullptr;
  hipEvent_t ev_pro3 = nullptr;     // third stream: its branch of the forward prologue (fc_embed, Gfc, initial state) is through
  hipEvent_t ev_pro = nullptr;      // side: its branch of the forward prologue (fc_embed, embedding, batched input GEMM) is through
  hipEvent_t ev_logit = nullptr;    // side: the logit layer's gradients and the loss are final (start of the BPTT loop)
  hipEvent_t ev_lstm = nullptr;     // side: lang_lstm.weight_{ih,hh} and att_lstm.weight_hh are final (right after the BPTT loop)
  hipEvent_t ev_early = nullptr;    // main: every gradient except the late group (see uic_topdown_grad_ready_wait) is final
  bool early_recorded = false;
  hipEvent_t ev_embed = nullptr;    // the embedding table's gradient and the fc' columns of att_lstm.weight_ih are final (gradient group 3)
  bool embed_recorded = false;
  hipEvent_t ev_r0 = nullptr, ev_refresh = nullptr;   // uic_topdown_refresh_weights: main -> side, side -> consumers
  bool refresh_pending = false;
  // The refresh in two halves: the operand-dtype copies (all the forward pass needs; ev_cast) are enqueued by the call
  // itself, the transposes (backward pass only) by the first consumer -- the fused step puts its side-stream prologue branch
  // in front of them, everything else gets them through wait_refresh.
  hipEvent_t ev_cast = nullptr;
  bool cast_recorded = false, transposes_pending = false;
  // uic_topdown_refresh_weights_gathered (deferred form): the side stream's copies of gather group 2 (embedding table, fc_embed,
  // att_lstm.weight_ih) are done -- the fused step's third stream (fc_embed, Gfc) waits for it; cleared by every other refresh
  hipEvent_t ev_g2 = nullptr;
  bool g2_recorded = false;
  // ... and its late half: the copies of gather groups 1 and 0 (recurrent LSTM matrices, logit layer), the last bytes to arrive.
  // Deferred to the fused step, which enqueues them BEHIND its third stream's prologue branch -- in front of it they would hold
  // that branch back until the last all-gather is through; nothing before the recurrence reads them (ev_gl: they are done)
  bool gl_pending = false, gl_recorded = false;
  hipEvent_t ev_gl = nullptr;
  const void* gl_src[UIC_CAST_MULTI]; void* gl_dst[UIC_CAST_MULTI]; size_t gl_bytes[UIC_CAST_MULTI]; int gl_count = 0;
  void* gl_ready[2] = {nullptr, nullptr};
  uic_topdown_dims tp_d; uic_topdown_weights tp_w; void* tp_derived = nullptr;
  hipEvent_t ev_main[MAX_CHUNKS];   // main  -> side: decode steps of chunk c are finished
  hipEvent_t ev_side[MAX_CHUNKS];   // side  -> main: d hdrop of chunk c is ready
  bool ready = false;
  // uic_topdown_step_marks: timing events of the last fused step (only recorded while marks_on)
  bool marks_on = false, marks_valid = false;
  hipEvent_t mark[UIC_STEP_MARKS];
};
SideStream g_side[16];
std::mutex g_side_mutex;   // guards the one-time creation of a device's streams / events (calls themselves are per device:
                           // one host thread drives a device at a time, as with any stream-ordered library state)

int get_side(SideStream** out) {
  int dev = 0;
  UIC_TRY(uic_check_hip(hipGetDevice(&dev), "hipGetDevice"));
  UIC_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
  SideStream& ss = g_side[dev];
  std::lock_guard<std::mutex> lock(g_side_mutex);
  if (!ss.ready) {
    // A plain second stream at the lowest priority.  Measured alternatives: confining it to a subset of the CUs
    // (hipExtStreamCreateWithCUMask, 64..224 CUs) makes the whole step 2x SLOWER; the priority itself is neutral.
    int least = 0, greatest = 0;
    UIC_TRY(uic_check_hip(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange"));
    UIC_TRY(uic_check_hip(hipStreamCreateWithPriority(&ss.stream, hipStreamNonBlocking, least), "hipStreamCreateWithPriority"));
    UIC_TRY(uic_check_hip(hipStreamCreateWithPriority(&ss.stream3, hipStreamNonBlocking, least), "hipStreamCreateWithPriority"));
    // NO fourth stream: a process gets full-speed dispatch from four HIP streams in all (tools/queue_probe.py: with the caller's
    // stream, these two and ONE more stream of the caller -- its communication stream -- a busy extra stream costs nothing; with a
    // fifth stream in existence, used or not, the same extra stream makes the step 0.7 ms longer).  The last chunk's second share
    // therefore runs behind the first on stream 3.
    ss.stream4 = ss.stream3;
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_s3, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_s4, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_prep, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_den, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_done, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_pro, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_pro3, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_early, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_embed, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_logit, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_lstm, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_r0, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_refresh, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_cast, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_g2, hipEventDisableTiming), "hipEventCreate"));
    UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_gl, hipEventDisableTiming), "hipEventCreate"));
    for (int i = 0; i < UIC_STEP_MARKS; ++i) UIC_TRY(uic_check_hip(hipEventCreate(&ss.mark[i]), "hipEventCreate"));
    for (int i = 0; i < MAX_CHUNKS; ++i) {
      UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_main[i], hipEventDisableTiming), "hipEventCreate"));
      UIC_TRY(uic_check_hip(hipEventCreateWithFlags(&ss.ev_side[i], hipEventDisableTiming), "hipEventCreate"));
    }
    ss.ready = true;
  }
  *out = &ss;
  return UIC_OK;
}

// the late half of uic_topdown_refresh_weights_gathered (see SideStream::gl_pending), on `st`
int flush_gathered_late(SideStream* ss, hipStream_t st) {
  if (!ss->gl_pending) return UIC_OK;
  ss->gl_pending = false;
  for (int i = 0; i < 2; ++i)
    if (ss->gl_ready[i]) UIC_TRY(uic_check_hip(hipStreamWaitEvent(st, (hipEvent_t)ss->gl_ready[i], 0), "hipStreamWaitEvent(gathered)"));
  UIC_TRY(uic_copy_multi_launch(ss->gl_count, ss->gl_src, ss->gl_dst, ss->gl_bytes, st));
  UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_gl, st), "hipEventRecord"));
  ss->gl_recorded = true;
  return UIC_OK;
}

// the deferred half of uic_topdown_refresh_weights: the transposed copies the backward pass reads, on the side stream
int flush_transposes(SideStream* ss) {
  if (ss->gl_pending) UIC_TRY(flush_gathered_late(ss, ss->stream));
  if (!ss->transposes_pending) return UIC_OK;
  ss->transposes_pending = false;
  if (ss->gl_recorded) UIC_TRY(uic_check_hip(hipStreamWaitEvent(ss->stream, ss->ev_gl, 0), "hipStreamWaitEvent(gathered late)"));
  const uic_topdown_dims* d = &ss->tp_d;
  const Derived v = make_derived(*d, &ss->tp_w, ss->tp_derived);
  hipStream_t s2 = ss->stream;
  const int dt = d->dtype, H = d->H, E = d->E, A = d->A, V1 = d->V1;
  const int V1p = (int)vpad(V1), H4 = 4 * H, ldih = E + 2 * H;
  // one launch for all of them (the launches, not the 10 MB, are what the side stream would spend its time on)
  UicTransposeJob jobs[UIC_TRANSPOSE_MULTI];
  int nj = 0;
  auto job = [&](const void* src, int rows, int cols, int lds, void* dst, int ldd) { jobs[nj++] = UicTransposeJob{src, dst, rows, cols, lds, ldd}; };
  job(v.logit_w, V1, H, H, v.logit_wT, V1p);
  // w2T rows [0,2H) <- lang_w_ih^T, rows [2H,3H) <- lang_w_hh^T
  job(v.lang_w_ih, H4, 2 * H, 2 * H, v.w2T, H4);
  job(v.lang_w_hh, H4, H, H, offw(v.w2T, (size_t)2 * H * H4, dt), H4);
  job(v.att_w_ih, H4, H, ldih, v.w1recT, H4);
  job(v.att_w_hh, H4, H, H, offw(v.w1recT, (size_t)H * H4, dt), H4);
  job(off(v.att_w_ih, 2 * H, dt), H4, E, ldih, v.wxT, H4);
  job(off(v.att_w_ih, H, dt), H4, H, ldih, v.wfcpT, H4);
  job(v.h2att_w, A, H, H, v.h2attT, A);
  job(v.ctx2att_w, A, H, H, v.ctx2attT, A);
  for (int l = 0; l + 1 < d->logit_layers; ++l) job(v.logit_h_w[l], H, H, H, v.logit_h_wT[l], H);
  UIC_TRY(uic_transpose_multi_launch(dt, nj, jobs, s2));
  UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_refresh, s2), "hipEventRecord"));
  ss->refresh_pending = true;
  return UIC_OK;
}

// every consumer of the derived weights first lets its stream wait for the side-stream part of the last refresh
int wait_refresh(hipStream_t s) {
  SideStream* ss = nullptr;
  UIC_TRY(get_side(&ss));
  UIC_TRY(flush_transposes(ss));
  if (ss->refresh_pending) UIC_TRY(uic_check_hip(hipStreamWaitEvent(s, ss->ev_refresh, 0), "hipStreamWaitEvent(refresh)"));
  return UIC_OK;
}

}  // namespace

extern "C" {

size_t uic_topdown_workspace_bytes(const uic_topdown_dims* d) {
  if (check_dims(d)) return 0;
  return make_layout(*d, nullptr).total;
}
size_t uic_topdown_derived_bytes(const uic_topdown_dims* d) {
  if (check_dims(d)) return 0;
  return make_derived(*d, nullptr, nullptr).total;
}

void* uic_topdown_workspace_ptr(const uic_topdown_dims* d, void* workspace, const char* name) {
  if (check_dims(d) || !workspace || !name) return nullptr;
  const Layout L = make_layout(*d, workspace);
  struct { const char* n; void* p; } tab[] = {
      {"tok_used", L.tok_used}, {"d_pre", L.d_pre}, {"fc_embed", L.fcp}, {"att_embed", L.attp}, {"p_att", L.patt}, {"e_att", L.eatt}, {"xt", L.xt_all}, {"gx", L.gx}, {"gfc", L.gfc},
      {"h_att", L.h_att}, {"h_lang", L.h_lang}, {"c_att", L.c_att}, {"c_lang", L.c_lang}, {"gates1", L.gates1},
      {"gates2", L.gates2}, {"att_h", L.atth_all}, {"alpha", L.alpha_all}, {"ctx", L.ctx_all}, {"hdrop", L.hdrop_all},
      {"logits", L.logits}, {"dlogits", L.dlogits}, {"row_loss", L.row_loss}, {"scalars", L.scalars},
      {"dhdrop", L.dhdrop}, {"dx2", L.dx2_all}, {"dg1", L.dg1_all}, {"dg2", L.dg2_all}, {"de", L.de_all},
      {"datth", L.datth_all}, {"d_att", L.d_att}, {"d_p_att", L.d_patt}, {"dxt", L.dxt}, {"rnn_dbg", L.rnn_dbg}, {"rnn_bwd_dbg", L.rnn_bwd_dbg},
      // ("dx1", and columns [H, 3H) of "dx2": written only when the d x GEMMs are not split-K, i.e. by the persistent BPTT kernel
      // or at shapes where bptt_split() == 0 -- the split chain keeps these sums in its slabs)
      {"dx1", L.dx1}, {"dc_att", L.dc_att}, {"dc_lang", L.dc_lang}};
  for (auto& e : tab)
    if (!strcmp(e.n, name)) return e.p;
  return nullptr;
}

int uic_topdown_refresh_weights(const uic_topdown_dims* d, const uic_topdown_weights* w, void* derived, void* stream) {
  UIC_TRY(uic_topdown_refresh_weights_deferred(d, w, derived, stream));
  SideStream* ss = nullptr;
  UIC_TRY(get_side(&ss));
  return flush_transposes(ss);
}

int uic_topdown_refresh_weights_deferred(const uic_topdown_dims* d, const uic_topdown_weights* w, void* derived, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived, "refresh_weights: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const Derived v = make_derived(*d, w, derived);
  const int dt = d->dtype;
  const int D = d->D, Dfc = d->Dfc, H = d->H, E = d->E, A = d->A, V1 = d->V1;
  const int V1p = (int)vpad(V1);
  // The copies the feature projection and the batched input GEMMs need come first, on the caller's stream; everything
  // the recurrence, the logit layer and the backward pass need is produced on the side stream meanwhile (consumers
  // wait for ev_refresh, see wait_refresh), so a training step does not start with ~0.1 ms of serial weight shuffling.
  SideStream* ss = nullptr;
  UIC_TRY(get_side(&ss));
  hipStream_t s2 = ss->stream;
  if (ss->transposes_pending && ss->tp_derived != derived) UIC_TRY(flush_transposes(ss));   // another model's deferred half
  ss->g2_recorded = false;
  ss->gl_pending = false;                             // (a gathered refresh nobody consumed: this refresh rewrites every copy)
  ss->gl_recorded = false;
  UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_r0, s), "hipEventRecord"));          // the master weights are final
  UIC_TRY(uic_check_hip(hipStreamWaitEvent(s2, ss->ev_r0, 0), "hipStreamWaitEvent"));
  if (dt == UIC_BF16) {   // one launch per stream (the step waits for launches here, not for bytes)
    {
      const float* src[4] = {w->fc_w, d->use_bn ? nullptr : w->att_w, w->ctx2att_w, w->att_lstm_w_ih};
      void* dst[4] = {(void*)v.fc_w, (void*)v.att_w, (void*)v.ctx2att_w, (void*)v.att_w_ih};
      const size_t n[4] = {(size_t)H * Dfc, d->use_bn ? 0 : (size_t)H * D, (size_t)A * H, (size_t)4 * H * (E + 2 * H)};
      UIC_TRY(uic_cast_f32_multi_launch(dt, 4, src, dst, n, s));
    }
    {
      // (the embedding table first: the side stream's branch of the fused step's prologue starts with the lookup)
      const float* src[6] = {w->embed_w, w->logit_w, w->att_lstm_w_hh, w->lang_lstm_w_ih, w->lang_lstm_w_hh, w->h2att_w};
      void* dst[6] = {(void*)v.embed_w, (void*)v.logit_w, (void*)v.att_w_hh, (void*)v.lang_w_ih, (void*)v.lang_w_hh, (void*)v.h2att_w};
      const size_t n[6] = {(size_t)V1 * E, (size_t)V1 * H, (size_t)4 * H * H, (size_t)4 * H * 2 * H, (size_t)4 * H * H, (size_t)A * H};
      UIC_TRY(uic_cast_f32_multi_launch(dt, 6, src, dst, n, s2));
    }
  }
  if (d->use_bn) {
    UIC_REQUIRE(w->att_bn0_w && w->att_bn0_b && w->att_bn0_rm && w->att_bn0_rv, "use_bn=%d needs the att_embed.0 BatchNorm tensors", d->use_bn);
    UIC_REQUIRE(d->use_bn < 2 || (w->att_bn4_w && w->att_bn4_b && w->att_bn4_rm && w->att_bn4_rv), "use_bn=2 needs the att_embed.4 BatchNorm tensors");
    UIC_TRY(uic_bn_fold_weight_launch(dt, w->att_w, w->att_bn0_w, w->att_bn0_b, w->att_b, H, D, (void*)v.att_w, v.att_beff, s));
  }
  // the side stream's later work (the transposes of att_w_ih among it) reads the copies made on `s` above
  UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_r0, s), "hipEventRecord"));
  UIC_TRY(uic_check_hip(hipStreamWaitEvent(s2, ss->ev_r0, 0), "hipStreamWaitEvent"));
  for (int l = 0; l + 1 < d->logit_layers; ++l) {
    UIC_REQUIRE(w->logit_h_w[l] && w->logit_h_b[l], "logit_layers=%d needs the hidden logit block %d", d->logit_layers, l);
    if (dt == UIC_BF16) UIC_TRY(uic_cast_f32_launch(dt, w->logit_h_w[l], (void*)v.logit_h_w[l], (size_t)H * H, s2));
  }
  UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_cast, s2), "hipEventRecord"));
  ss->cast_recorded = true;
  // the transposes (backward pass only) follow when the first consumer asks for them: flush_transposes
  ss->tp_d = *d; ss->tp_w = *w; ss->tp_derived = derived;
  ss->transposes_pending = true;
  return UIC_OK;
}


// The refresh of a data-parallel rank whose optimizer is sharded (reduce-scatter -> Adam on the rank's shard -> all-gather of the
// updated weights in the operand dtype): the operand copies are not cast from the f32 masters -- a rank's masters are current only
// inside its own shard -- but copied from the all-gathered operand-dtype arena (`g`), each gather group behind the event the
// caller recorded after that group's all-gather.  f32 operands: the gathered tensors ARE the masters (make_derived points at
// them); the call only orders the streams behind the events.
int uic_topdown_refresh_weights_gathered(const uic_topdown_dims* d, const uic_topdown_weights* w, const uic_topdown_gathered* g,
                                         void* derived, int32_t deferred, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && g && derived, "refresh_weights_gathered: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const Derived v = make_derived(*d, w, derived);
  const int dt = d->dtype;
  const size_t D = d->D, Dfc = d->Dfc, H = d->H, E = d->E, A = d->A, V1 = d->V1, S = uic_dtype_size(dt);
  SideStream* ss = nullptr;
  UIC_TRY(get_side(&ss));
  hipStream_t s2 = ss->stream;
  if (ss->transposes_pending && ss->tp_derived != derived) UIC_TRY(flush_transposes(ss));
  ss->gl_pending = false;                             // (a gathered refresh nobody consumed: this one rewrites every copy)
  ss->g2_recorded = false;
  ss->gl_recorded = false;
  auto wait_group = [&](hipStream_t st, int grp) -> int {
    if (g->ready[grp]) UIC_TRY(uic_check_hip(hipStreamWaitEvent(st, (hipEvent_t)g->ready[grp], 0), "hipStreamWaitEvent(gathered)"));
    return UIC_OK;
  };
  if (dt != UIC_BF16) {
    // f32 operands: nothing to copy; every consumer stream is ordered behind `s`, which waits for all four groups
    UIC_REQUIRE((!g->logit_w || g->logit_w == w->logit_w) && (!g->embed_w || g->embed_w == w->embed_w) && (!g->att_w || g->att_w == w->att_w),
                "refresh_weights_gathered: with f32 operands the gathered tensors must be the masters themselves");
    for (int grp = 3; grp >= 0; --grp) UIC_TRY(wait_group(s, grp));
    return deferred ? uic_topdown_refresh_weights_deferred(d, w, derived, stream) : uic_topdown_refresh_weights(d, w, derived, stream);
  }
  UIC_REQUIRE(g->embed_w && g->fc_w && g->logit_w && g->ctx2att_w && g->att_lstm_w_ih && g->att_lstm_w_hh && g->lang_lstm_w_ih &&
                  g->lang_lstm_w_hh && g->h2att_w && (g->att_w || d->use_bn),
              "refresh_weights_gathered: a gathered tensor is missing");
  UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_r0, s), "hipEventRecord"));          // whatever the caller enqueued before (the replicated f32 tensors' update)
  UIC_TRY(uic_check_hip(hipStreamWaitEvent(s2, ss->ev_r0, 0), "hipStreamWaitEvent"));
  {   // gather group 3 on the caller's stream: all the main branch of the fused step's prologue reads (att_embed, ctx2att) + h2att
    UIC_TRY(wait_group(s, 3));
    const void* src[3] = {d->use_bn ? nullptr : g->att_w, g->ctx2att_w, g->h2att_w};
    void* dst[3] = {(void*)v.att_w, (void*)v.ctx2att_w, (void*)v.h2att_w};
    const size_t n[3] = {d->use_bn ? 0 : H * D * S, A * H * S, A * H * S};
    UIC_TRY(uic_copy_multi_launch(3, src, dst, n, s));
  }
  {   // group 2 on the side stream (its prologue branch -- embedding lookup, batched input GEMM -- follows there; the third stream's
      // fc_embed / Gfc wait for ev_g2), then the recurrence's and the logit layer's groups
    UIC_TRY(wait_group(s2, 2));
    const void* src[3] = {g->embed_w, g->fc_w, g->att_lstm_w_ih};
    void* dst[3] = {(void*)v.embed_w, (void*)v.fc_w, (void*)v.att_w_ih};
    const size_t n[3] = {V1 * E * S, H * Dfc * S, 4 * H * (E + 2 * H) * S};
    UIC_TRY(uic_copy_multi_launch(3, src, dst, n, s2));
    UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_g2, s2), "hipEventRecord"));
    ss->g2_recorded = true;
  }
  {
    UIC_REQUIRE(d->logit_layers - 1 + 4 <= UIC_CAST_MULTI, "refresh_weights_gathered: too many logit layers (%d)", d->logit_layers);
    int k = 0;
    auto add = [&](const void* src, const void* dst, size_t bytes) { ss->gl_src[k] = src; ss->gl_dst[k] = (void*)dst; ss->gl_bytes[k] = bytes; ++k; };
    add(g->att_lstm_w_hh, v.att_w_hh, 4 * H * H * S);
    add(g->lang_lstm_w_ih, v.lang_w_ih, 4 * H * 2 * H * S);
    add(g->lang_lstm_w_hh, v.lang_w_hh, 4 * H * H * S);
    add(g->logit_w, v.logit_w, V1 * H * S);
    for (int l = 0; l + 1 < d->logit_layers; ++l) {
      UIC_REQUIRE(g->logit_h_w[l] && w->logit_h_b[l], "logit_layers=%d needs the hidden logit block %d", d->logit_layers, l);
      add(g->logit_h_w[l], v.logit_h_w[l], H * H * S);
    }
    ss->gl_count = k;
    ss->gl_ready[0] = g->ready[1]; ss->gl_ready[1] = g->ready[0];
    ss->gl_pending = true;
    ss->gl_recorded = false;
    if (!deferred) UIC_TRY(flush_gathered_late(ss, s2));
  }
  if (d->use_bn) {   // att_embed's Linear with the BatchNorm folded in needs the f32 master: the caller keeps that tensor replicated
    UIC_REQUIRE(w->att_bn0_w && w->att_bn0_b && w->att_bn0_rm && w->att_bn0_rv, "use_bn=%d needs the att_embed.0 BatchNorm tensors", d->use_bn);
    UIC_TRY(uic_bn_fold_weight_launch(dt, w->att_w, w->att_bn0_w, w->att_bn0_b, w->att_b, (int)H, (int)D, (void*)v.att_w, v.att_beff, s));
  }
  UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_r0, s), "hipEventRecord"));          // the side stream's transposes read ctx2att / h2att copied on `s`
  UIC_TRY(uic_check_hip(hipStreamWaitEvent(s2, ss->ev_r0, 0), "hipStreamWaitEvent"));
  UIC_TRY(uic_check_hip(hipEventRecord(ss->ev_cast, s2), "hipEventRecord"));
  ss->cast_recorded = true;
  ss->tp_d = *d; ss->tp_w = *w; ss->tp_derived = derived;
  ss->transposes_pending = true;
  if (!deferred) {
    // any consumer may follow on `s` (decode passes, single calls): everything the side stream copied is ordered in front of it
    UIC_TRY(flush_transposes(ss));
    UIC_TRY(uic_check_hip(hipStreamWaitEvent(s, ss->ev_refresh, 0), "hipStreamWaitEvent(refresh)"));
  }
  return UIC_OK;
}

}  // extern "C" (re-opened below)

namespace {

// One teacher-forced step of the captioner on one GPU, split into the pieces the public entry points
// (and the two-stream fused training step) sequence.
struct Step {
  uic_topdown_dims d;
  const uic_topdown_weights* w;
  const uic_topdown_batch* b;
  const uic_topdown_weights* G;
  Layout L;
  Derived dv;
  int dt, N, R, D, Dfc, H, E, A, V1, V1p, H4, ldih, t_run, Meff, Mp, Np, NR, NRp;
  size_t S, NH;
  float drop_p, inv_keep;
  unsigned seed;
  int training;          // bit 0: train mode; bit 1: keep the BatchNorm running statistics untouched
  const void* fc_in;
  const void* att_in;

  void init(const uic_topdown_dims* d_, const uic_topdown_weights* w_, const void* derived, const uic_topdown_batch* b_,
            int t_run_, int training_, unsigned seed_, void* workspace, const uic_topdown_weights* G_) {
    d = *d_; w = w_; b = b_; G = G_;
    L = make_layout(d, workspace);
    dv = make_derived(d, w, (void*)derived);
    dt = d.dtype; N = d.N; R = d.R; D = d.D; Dfc = d.Dfc; H = d.H; E = d.E; A = d.A; V1 = d.V1;
    V1p = (int)vpad(V1); H4 = 4 * H; ldih = E + 2 * H; t_run = t_run_;
    Meff = t_run * N; Mp = (int)rup8(Meff); Np = (int)rup8(N); NR = N * R; NRp = (int)rup8(NR);
    S = uic_dtype_size(dt); NH = (size_t)N * H;
    training = training_;
    drop_p = (training & 1) ? d.drop_p : 0.f;
    inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    seed = seed_;
    fc_in = (dt == UIC_BF16 || d.seq_per_img > 1) ? L.fcT : (const void*)b->fc_feats;
    att_in = (dt == UIC_BF16 || d.use_bn) ? L.attT : (const void*)b->att_feats;   // per image when seq_per_img > 1
  }

  // scheduled sampling (AttModel.py:130-143) is active in train mode only
  bool ss_on() const { return (training & 1) && b->ss_prob > 0.f; }
  bool embed_prepared = false;   // the fused step bucketed the tokens (uic_embed_bwd_sorted_prepare) while the side stream was idle
  // Live-position logit layer (round 6).  uic_topdown_batch.live_rows / live_count list the (step, row) positions whose mask is not
  // zero; positions behind a caption's end -- a quarter of the benchmark's, a third of COCO's -- have loss 0 and gradient 0, so the
  // logit layer (GEMM, criterion, d hdrop, its weight gradient) runs over the listed rows only: gathered into L.hc, logits / d
  // logits / d hdrop compact, d hdrop scattered back into the (zeroed) step-major buffer the BPTT loop reads.
  bool compact = false, build_live = false;
  const int32_t* live_rows = nullptr;
  int live_off[UIC_MAX_LIVE_STEPS + 1];
  int live_total() const { return live_off[t_run]; }
  int live_pad() const { return (live_total() + 127) & ~127; }
  void init_live() {
    compact = false;
    build_live = false;
    if (!b->live_count || !b->masks || t_run > UIC_MAX_LIVE_STEPS || ss_on() || b->grad_scale || nlh() != 0) return;
    if (((size_t)H * uic_dtype_size(dt)) % 16 != 0) return;     // (rows move as 16-byte pieces)
    live_off[0] = 0;
    for (int t = 0; t < t_run; ++t) {
      const int c = b->live_count[t];
      if (c < 0 || c > N) return;                       // (not a list of this batch: the plain path)
      live_off[t + 1] = live_off[t] + c;
    }
    // (no list from the caller: one small launch compacts the masks on the device, uic_live_list_launch -- the caller then
    // ships nothing but the counts it sizes the launches with)
    build_live = b->live_rows == nullptr;
    live_rows = build_live ? L.live_map : b->live_rows;
    compact = true;
  }
  // fused_gather: the persistent recurrence stores every live row of hdrop into the compact operand itself (rnn_persist.hip, by the
  // list's inverse): no gather launches -- one of them sat on the main stream in front of the BPTT loop.  The list kernel clears the
  // operand's padding rows then.
  bool fused_gather = false;
  // have_len: L.cap_len holds every row's caption length (1 + the last decode step with a non-zero mask / gradient weight): the BPTT
  // loop's attention backward and the attention accumulation skip the steps behind it.  Made with the list, or -- a step that
  // computes every position with per-position weights (the self-critical step: a sampled caption's positions behind its end weigh
  // zero) -- by the list kernel alone.
  bool have_len = false;
  int len_build(hipStream_t s) {
    return uic_live_list_launch(b->grad_scale, b->ld_grad_scale, 0, N, t_run * N, L.live_map, 0, s, nullptr, nullptr, 0, L.cap_len);
  }
  int live_build(hipStream_t s) {
    const size_t S = uic_dtype_size(dt);
    return uic_live_list_launch(b->masks, b->ld_masks, 1, N, t_run * N, L.live_map, live_pad(), s, fused_gather ? L.live_inv : nullptr,
                                fused_gather ? offw(L.hc, (size_t)live_total() * H, dt) : nullptr,
                                fused_gather ? (size_t)(live_pad() - live_total()) * H * S : 0, L.cap_len);
  }
  int embed_split = 0;           // > 0: ... into the halves [0, embed_split) / [embed_split, t_run) of the decode steps (embed_grad)
  // d.recurrence & UIC_REC_EARLY_GRADS (fused step): the order of the gradient work a data-parallel caller may prefer -- see
  // uic_topdown_xe_train_step
  bool early_grads() const { return (d.recurrence & UIC_REC_EARLY_GRADS) != 0; }
  const int64_t* embed_tokens() const { return ss_on() ? L.tok_used : b->labels; }
  int embed_ldtok() const { return ss_on() ? d.T : b->ld_labels; }

  // ---------------------------------------------------------------- forward
  // part: 0 = everything on one stream; 1 = the branch that ends in the batched input GEMM (fc_embed, embedding, Gfc, Gx,
  // initial state), 2 = the att_embed / ctx2att branch -- independent of each other (the fused step forks them)
  // gfc_separate: the persistent recurrence adds the caption row's fc' term (Gfc) itself, so the batched input GEMM (the long
  // pole of the prologue's side-stream branch) no longer queues behind cast -> fc_embed -> Gfc; the launch chain keeps Gfc
  // folded into Gx (one operand less per step).  Same f32 additions in the same order either way.
  bool gfc_separate() const { return UIC_GFC_SEPARATE && persist_ok(); }
  int fwd_embed(hipStream_t s) {
    // xt_t = dropout(relu(embed[labels[:, t]])) for all steps (AttModel.py:145,160)
    return uic_embed_fwd_t_launch(dt, dv.embed_w, dt, V1, E, b->labels, b->ld_labels, N, t_run, drop_p, seed, UIC_SITE_EMBED, 0, 1, L.xt_all, s);
  }
  int fwd_gx(hipStream_t s, bool with_gfc) {
    // Gx = xt W_ih[:, 2H:]^T + b_ih + b_hh [+ Gfc (every step's rows get their caption row's fc' term)], all steps
    UicGemmParams g = gemm_base(dt, Meff, H4);
    add_seg(g, L.xt_all, E, off(dv.att_w_ih, 2 * H, dt), ldih, E);
    g.C = L.gx; g.ldc = H4; g.bias = w->att_lstm_b_ih; g.bias2 = w->att_lstm_b_hh; g.flags = UIC_GEMM_OUT_F32;
    if (with_gfc) { g.addend = L.gfc; g.add_mod = N; g.ld_add = H4; }
    return uic_gemm_launch(g, s);
  }
  // part: 0 = everything; 1 = the side stream's branch (embedding, batched input GEMM, fc_embed, Gfc, initial state); 2 = the
  // main stream's (att_embed, ctx2att); with gfc_separate() the side branch splits again into 3 = embedding + batched input
  // GEMM and 4 = fc_embed + Gfc + initial state (independent of each other: the fused step gives 4 to its third stream)
  int fwd_prologue(hipStream_t s, int part = 0) {
    const void *f, *a;
    const bool sep = gfc_separate();
    UIC_REQUIRE(part < 3 || sep, "fwd_prologue: parts 3 / 4 need the separate Gfc");
    if (part != 2 && part != 4 && sep) {
      UIC_TRY(fwd_embed(s));
      UIC_TRY(fwd_gx(s, false));
    }
    if (part == 3) return UIC_OK;
    UIC_TRY(prepare_features(d, w, dv, b, L, training, drop_p, seed, &f, &a, s, part == 4 ? 1 : part));
    if (part == 2) return UIC_OK;
    if (!sep) UIC_TRY(fwd_embed(s));
    UIC_TRY(fwd_gfc(s));
    if (!sep) UIC_TRY(fwd_gx(s, true));
    if (ss_on()) UIC_TRY(uic_copy_tokens_launch(b->labels, b->ld_labels, N, t_run, L.tok_used, d.T, s));
    // init_hidden (AttModel.py:94-97): slot 0 of the four state buffers, one launch
    return uic_zero4_launch(L.h_att, NH * S, L.h_lang, NH * S, L.c_att, NH * 4, L.c_lang, NH * 4, s);
  }

  // step t's embedding row block and its slice of Gx from the tokens tok[n * ld] (the prologue made them from the labels)
  int fwd_step_inputs(int t, const int64_t* tok, int ld, hipStream_t s) {
    UIC_TRY(uic_embed_fwd_t_launch(dt, dv.embed_w, dt, V1, E, tok, ld, N, 1, drop_p, seed, UIC_SITE_EMBED,
                                 (size_t)t * N * E, 1, offw(L.xt_all, (size_t)t * N * E, dt), s));
    UicGemmParams g = gemm_base(dt, N, H4);
    add_seg(g, off(L.xt_all, (size_t)t * N * E, dt), E, off(dv.att_w_ih, 2 * H, dt), ldih, E);
    g.C = L.gx + (size_t)t * N * H4; g.ldc = H4; g.bias = w->att_lstm_b_ih; g.bias2 = w->att_lstm_b_hh; g.flags = UIC_GEMM_OUT_F32;
    g.addend = L.gfc; g.add_mod = N; g.ld_add = H4;
    return uic_gemm_launch(g, s);
  }
  // Gfc = fc' W_ih[:, H:2H]^T, the caption row's share of every step's att_lstm pre-activations
  int fwd_gfc(hipStream_t s, bool with_bias = false) {
    UicGemmParams g = gemm_base(dt, N, H4);
    add_seg(g, L.fcp, H, off(dv.att_w_ih, H, dt), ldih, H);
    g.C = L.gfc; g.ldc = H4; g.flags = UIC_GEMM_OUT_F32;
    if (with_bias) { g.bias = w->att_lstm_b_ih; g.bias2 = w->att_lstm_b_hh; }
    return uic_gemm_launch(g, s);
  }

  // ---------------------------------------------------------------- decode as one persistent launch
  // AttModel._sample (P/models/AttModel.py:198-253) with beam_size = 1: all Lsteps decode steps -- embedding of the previous
  // step's token, att_lstm, attention, lang_lstm, the logit layer and the greedy / multinomial choice -- in ONE launch of
  // rnn_persist.hip's decode mode instead of six launches per step.  bf16, the default temperature, no decoding constraint,
  // one logit layer; everything else keeps the per-step chain.  keep: the training layout's activations, the embedded inputs
  // and every step's logits stay in the workspace for uic_topdown_xe_train_step(training | 4).
  bool decode_persist_ok(int sample_max, float temperature, int decoding_constraint) const {
    return !(d.recurrence & UIC_REC_FWD_CHAIN) && !decoding_constraint && (sample_max || temperature == 1.f) && d.logit_layers <= 1 &&
           uic_rnn_decode_persist_eligible(dt, N, H, A, R, E, V1);
  }
  int decode_persist(int Lsteps, int sample_max, const int64_t* forced, bool keep, int64_t* seq, float* seq_logp, hipStream_t s) {
    UIC_TRY(fwd_gfc(s, true));
    UIC_TRY(uic_rnn_decode_embed_relu_launch(dv.embed_w, dt, L.dec_embed_relu, V1, E, s));
    UIC_TRY(uic_fill_launch(L.dec_tok, 0, (size_t)N * 4, s));          // <bos> = 0 (AttModel.py:214-215)
    UicRnnFwdParams p;
    memset(&p, 0, sizeof(p));
    p.dtype = dt; p.N = N; p.R = R; p.t0 = 0; p.t1 = Lsteps;
    p.gx = nullptr; p.gfc = L.gfc;
    p.att_w_ih = dv.att_w_ih; p.ld_att_ih = ldih; p.att_w_hh = dv.att_w_hh;
    p.lang_w_ih = dv.lang_w_ih; p.lang_w_hh = dv.lang_w_hh;
    p.lang_b_ih = w->lang_lstm_b_ih; p.lang_b_hh = w->lang_lstm_b_hh;
    p.h2att_w = dv.h2att_w; p.h2att_b = w->h2att_b;
    p.w_alpha = w->alpha_w; p.b_alpha = w->alpha_b;
    p.p_att = L.patt; p.att = L.attp;
    p.mask = b->att_masks ? (d.seq_per_img > 1 ? L.amask_rep : b->att_masks) : nullptr; p.ldmask = R;
    p.h_att = L.h_att; p.h_lang = L.h_lang; p.c_att = L.c_att; p.c_lang = L.c_lang;
    p.gates1 = keep ? L.gates1 : nullptr; p.gates2 = keep ? L.gates2 : nullptr;
    p.att_h_all = L.atth_all; p.alpha_all = L.alpha_all; p.ctx_all = L.ctx_all; p.hdrop_all = L.hdrop_all;
    p.drop_p = drop_p; p.seed = seed;
    p.sync = L.rnn_sync;
    p.force_safe = (d.recurrence & UIC_REC_SAFE) != 0; p.status = d.rnn_status;
    p.dbg = (d.recurrence & UIC_REC_STAMPS) ? L.rnn_dbg : nullptr; p.dbg_T = d.T;
    p.dec = 1;
    p.dec_embed_relu = L.dec_embed_relu; p.dec_xw = off(dv.att_w_ih, 2 * H, dt); p.dec_ld_xw = ldih;
    p.dec_xt_drop = drop_p; p.dec_xt_all = keep ? L.xt_all : nullptr;
    p.dec_logit_w = dv.logit_w; p.dec_logit_b = w->logit_b; p.dec_V1 = V1; p.dec_V1p = V1p;
    p.dec_logits = keep ? L.logits : L.s_logits; p.dec_logits_step = keep ? (size_t)N * V1p : 0;
    p.dec_part = L.dec_part; p.dec_tok = L.dec_tok; p.dec_unf = L.s_unf;
    p.dec_seq = seq; p.dec_seq_logp = seq_logp; p.dec_ld_out = Lsteps;
    p.dec_forced = sample_max ? nullptr : forced; p.dec_ld_forced = Lsteps;
    p.dec_sample_max = sample_max; p.dec_draw_seed = seed;
    UIC_TRY(uic_rnn_fwd_persist_launch(p, s));
    return uic_rnn_decode_finish_launch(seq, seq_logp, N, Lsteps, Lsteps, (const int*)d.rnn_status, s);
  }

  // inline_inputs: xt_t and fc' enter att_lstm's GEMM as K segments of their own (with both biases) instead of through the
  // prologue's Gx -- the sampling pass, whose step-t embedding exists only once step t - 1 has drawn its tokens
  int fwd_step(int t, hipStream_t s, bool inline_inputs = false) {
    if (ss_on() && t >= 1) {
      // choose this step's input tokens from the previous step's distribution, then redo the step's embedding row block
      // and its slice of Gx (the prologue's teacher-forced values for the rows that keep their label are recomputed too)
      UIC_TRY(uic_ss_sample_launch(L.logits + (size_t)(t - 1) * N * V1p, N, V1, V1p, b->labels, b->ld_labels, t, b->ss_prob, seed,
                                   L.tok_used, d.T, s));
      UIC_TRY(fwd_step_inputs(t, L.tok_used + t, d.T, s));
    }
    const void* h_att_prev = off(L.h_att, t * NH, dt);
    void* h_att_new = offw(L.h_att, (t + 1) * NH, dt);
    const void* h_lang_prev = off(L.h_lang, t * NH, dt);
    void* h_lang_new = offw(L.h_lang, (t + 1) * NH, dt);
    {  // att_lstm on cat([h_lang_prev, fc', xt]) (AttModel.py:431-434)
      UicGemmParams g = gemm_base(dt, N, H4);
      g.lstm = 1; g.H = H;
      add_seg(g, h_lang_prev, H, dv.att_w_ih, ldih, H);
      if (inline_inputs) {
        add_seg(g, L.fcp, H, off(dv.att_w_ih, H, dt), ldih, H);
        add_seg(g, off(L.xt_all, (size_t)t * N * E, dt), E, off(dv.att_w_ih, 2 * H, dt), ldih, E);
        g.bias = w->att_lstm_b_ih; g.bias2 = w->att_lstm_b_hh;
      } else {
        g.pre1 = L.gx + (size_t)t * N * H4; g.ldpre1 = H4;   // (Gfc is already folded into Gx)
      }
      add_seg(g, h_att_prev, H, dv.att_w_hh, H, H);
      g.c_prev = L.c_att + t * NH; g.c_out = L.c_att + (t + 1) * NH;
      g.h_out = h_att_new; g.ldh = H;
      g.gates_out = offw(L.gates1, (size_t)t * N * H4, dt);
      UIC_TRY(uic_gemm_launch(g, s));
    }
    UIC_TRY(attention_step(d, w, dv, b, L, h_att_new, L.atth_all + (size_t)t * N * A, L.alpha_all + (size_t)t * N * R,
                           offw(L.ctx_all, t * NH, dt), s));
    {  // lang_lstm on cat([att_res, h_att]) (AttModel.py:438-441) + output dropout (:443)
      UicGemmParams g = gemm_base(dt, N, H4);
      g.lstm = 1; g.H = H;
      add_seg(g, off(L.ctx_all, t * NH, dt), H, dv.lang_w_ih, 2 * H, H);
      add_seg(g, h_att_new, H, off(dv.lang_w_ih, H, dt), 2 * H, H);
      add_seg(g, h_lang_prev, H, dv.lang_w_hh, H, H);
      g.bias = w->lang_lstm_b_ih; g.bias2 = w->lang_lstm_b_hh;
      g.c_prev = L.c_lang + t * NH; g.c_out = L.c_lang + (t + 1) * NH;
      g.h_out = h_lang_new; g.ldh = H;
      g.h_drop = offw(L.hdrop_all, t * NH, dt); g.ldhd = H;
      g.drop_p = drop_p; g.seed = seed; g.site = UIC_SITE_OUT0 + (unsigned)t;
      g.gates_out = offw(L.gates2, (size_t)t * N * H4, dt);
      UIC_TRY(uic_gemm_launch(g, s));
    }
    if (ss_on()) UIC_TRY(logits_rows_now(t, t + 1, s));   // the next step samples from this step's distribution
    return UIC_OK;
  }

  // decode steps [t0, t1) of the recurrence: one persistent launch (rnn_persist.hip) when the shapes allow, else the
  // per-step chain.  Scheduled sampling needs the logits of step t - 1 on the host-sequenced path.
  // The persistent kernel holds every CU (160 KB of LDS, the whole register file): inside the two-stream training step the
  // logit layer cannot run beside it; it follows chunk by chunk, last chunk first, beside the BPTT loop (see
  // uic_topdown_xe_train_step).  d.recurrence & UIC_REC_FWD_CHAIN keeps the per-step launches.
  bool persist_ok() const {
    return !ss_on() && !(d.recurrence & UIC_REC_FWD_CHAIN) && uic_rnn_persist_eligible(dt, N, H, A, R);
  }
  int fwd_steps(int t0, int t1, hipStream_t s) {
    if (!persist_ok()) {
      for (int t = t0; t < t1; ++t) UIC_TRY(fwd_step(t, s));
      return UIC_OK;
    }
    UicRnnFwdParams p;
    memset(&p, 0, sizeof(p));
    p.dtype = dt; p.N = N; p.R = R; p.t0 = t0; p.t1 = t1;
    p.gx = L.gx; p.gfc = gfc_separate() ? L.gfc : nullptr;
    p.att_w_ih = dv.att_w_ih; p.ld_att_ih = ldih; p.att_w_hh = dv.att_w_hh;
    p.lang_w_ih = dv.lang_w_ih; p.lang_w_hh = dv.lang_w_hh;
    p.lang_b_ih = w->lang_lstm_b_ih; p.lang_b_hh = w->lang_lstm_b_hh;
    p.h2att_w = dv.h2att_w; p.h2att_b = w->h2att_b;
    p.w_alpha = w->alpha_w; p.b_alpha = w->alpha_b;
    p.p_att = L.patt; p.att = L.attp; p.e_att = L.eatt;
    p.mask = b->att_masks ? (d.seq_per_img > 1 ? L.amask_rep : b->att_masks) : nullptr; p.ldmask = R;
    p.h_att = L.h_att; p.h_lang = L.h_lang; p.c_att = L.c_att; p.c_lang = L.c_lang;
    p.gates1 = L.gates1; p.gates2 = L.gates2;
    p.att_h_all = L.atth_all; p.alpha_all = L.alpha_all; p.ctx_all = L.ctx_all; p.hdrop_all = L.hdrop_all;
    p.drop_p = drop_p; p.seed = seed;
    if (fused_gather) { p.live_inv = L.live_inv; p.hdrop_live = L.hc; }
    p.sync = L.rnn_sync; p.sync_zeroed = fwd_sync_clean ? 1 : 0;
    fwd_sync_clean = false;            // (one launch's worth)
    p.force_safe = (d.recurrence & UIC_REC_SAFE) != 0; p.status = d.rnn_status;
    p.dbg = (d.recurrence & UIC_REC_STAMPS) ? L.rnn_dbg : nullptr; p.dbg_T = d.T;
    return uic_rnn_fwd_persist_launch(p, s);
  }

  // logits of decode steps [t0, t1) (AttModel.py:163); under scheduled sampling every step already produced its own
  int logits_rows(int t0, int t1, hipStream_t s) {
    if (ss_on()) return UIC_OK;
    return logits_rows_now(t0, t1, s);
  }
  // logit_layers = n > 1 (AttModel.py:90-91): n - 1 blocks Linear(H, H) + ReLU + Dropout(0.5) in front of the vocabulary layer
  int nlh() const { return d.logit_layers > 1 ? d.logit_layers - 1 : 0; }
  const void* logit_in_all() const { return nlh() ? L.lh[nlh() - 1] : L.hdrop_all; }
  // rows [c0, c0 + rows) of the live list = the live positions of decode steps [t0, t1); the list's last chunk carries the padding
  // rows up to a multiple of 128 (zero operand rows, zero gradient rows: the weight-gradient GEMM's K runs over whole tiles)
  // d hdrop of the positions the list leaves out is zero: the whole buffer is cleared before the chunks scatter into it
  int live_begin(hipStream_t s) { return uic_zero4_launch(L.dhdrop, (size_t)Meff * H * 4, nullptr, 0, nullptr, 0, nullptr, 0, s); }
  int live_c0(int t0) const { return live_off[t0]; }
  int live_rows_of(int t0, int t1) const { return (t1 == t_run ? live_pad() : live_off[t1]) - live_off[t0]; }
  int logits_rows_live(int t0, int t1, hipStream_t s) {
    const int c0 = live_c0(t0), real = live_off[t1] - c0, rows = live_rows_of(t0, t1);
    if (!fused_gather) UIC_TRY(uic_gather_rows_launch(L.hdrop_all, live_rows + c0, Meff, offw(L.hc, (size_t)c0 * H, dt), real, rows, (size_t)H * uic_dtype_size(dt), s));
    if (rows == 0) return UIC_OK;
    UicGemmParams g = gemm_base(dt, rows, V1);
    add_seg(g, off(L.hc, (size_t)c0 * H, dt), H, dv.logit_w, H, H);
    g.C = L.logits + (size_t)c0 * V1p; g.ldc = V1p; g.bias = w->logit_b; g.flags = UIC_GEMM_OUT_F32;
    return uic_gemm_launch(g, s);
  }
  int logits_rows_now(int t0, int t1, hipStream_t s) {
    if (compact) return logits_rows_live(t0, t1, s);
    const int rows = (t1 - t0) * N;
    for (int l = 0; l < nlh(); ++l) {
      UicGemmParams g = gemm_base(dt, rows, H);
      add_seg(g, off(l ? L.lh[l - 1] : L.hdrop_all, t0 * NH, dt), H, dv.logit_h_w[l], H, H);
      g.C = offw(L.lh[l], t0 * NH, dt); g.ldc = H; g.bias = w->logit_h_b[l]; g.flags = UIC_GEMM_RELU;
      if (training & 1) { g.drop_p = 0.5f; g.seed = seed; g.site = UIC_SITE_LOGIT_H0 + (unsigned)l; g.drop_row0 = t0 * N; }
      UIC_TRY(uic_gemm_launch(g, s));
    }
    UicGemmParams g = gemm_base(dt, rows, V1);
    add_seg(g, off(logit_in_all(), t0 * NH, dt), H, dv.logit_w, H, H);
    g.C = L.logits + (size_t)t0 * N * V1p; g.ldc = V1p; g.bias = w->logit_b; g.flags = UIC_GEMM_OUT_F32;
    return uic_gemm_launch(g, s);
  }
  // log-softmax + LanguageModelCriterion rows of steps [t0, t1): row losses and d logits
  int xe_rows(int t0, int t1, const float* inv, float* logprobs_out, int with_loss, hipStream_t s) {
    UicXeParams x;
    memset(&x, 0, sizeof(x));
    const size_t r0 = (size_t)t0 * N;
    x.dtype = dt; x.M = (t1 - t0) * N; x.V1 = V1; x.ldv = V1p; x.N = N;
    x.logits = L.logits + r0 * V1p;
    if (compact && with_loss) {
      // the live list's rows of these steps: logits / d logits / row losses are indexed by list position, labels and masks by the
      // (step, row) the list names (steps counted from 0: the column offsets below are those of step 0)
      const int c0 = live_c0(t0);
      x.M = live_rows_of(t0, t1);
      if (x.M == 0) return UIC_OK;
      x.row_map = live_rows + c0; x.row_map_limit = Meff;
      x.logits = L.logits + (size_t)c0 * V1p;
      x.dlogits = offw(L.dlogits, (size_t)c0 * V1p, dt);
      x.target = b->labels; x.ldtarget = b->ld_labels; x.target_col0 = 1;
      x.mask = b->masks; x.ldmask = b->ld_masks; x.mask_col0 = 1;
      x.inv_den = inv; x.row_loss = L.row_loss + c0; x.write_grad = 1;
      return uic_xe_launch(x, s);
    }
    if (with_loss) {
      x.dlogits = offw(L.dlogits, r0 * V1p, dt);
      x.target = b->labels; x.ldtarget = b->ld_labels; x.target_col0 = 1 + t0;
      x.mask = b->masks; x.ldmask = b->ld_masks; x.mask_col0 = 1 + t0;
      x.inv_den = inv; x.row_loss = L.row_loss + r0; x.write_grad = 1;
      x.grad_scale = b->grad_scale; x.ldscale = b->ld_grad_scale; x.scale_col0 = t0;
    }
    if (logprobs_out) {
      x.logprobs = logprobs_out + (size_t)t0 * V1; x.lp_step_stride = V1; x.lp_row_stride = (size_t)d.T * V1;
    }
    return uic_xe_launch(x, s);
  }
  // d hdrop rows of steps [t0, t1) = d logits W_logit
  // (long K = V1, few output tiles: split-K over workgroups on the LDS-DMA GEMM when the shape allows; `side` picks
  // the side stream's slab)
  int dh_rows(int t0, int t1, hipStream_t s, bool side = false) {
    if (compact) {
      const int c0 = live_c0(t0), real = live_off[t1] - c0;
      if (real == 0) return UIC_OK;
      const WDest dc{L.dhc + (size_t)c0 * H, H, 0, H};
      // (a chunk of a few rows -- the captions' last steps -- runs as one whole 128-row tile of the split-K kernel instead of the
      // generic one, 50 -> 17 us on the main stream in front of the BPTT loop: the rows behind the chunk's are the next chunk's or
      // padding, inside both buffers, and what they produce is not scattered)
      const int rows = real < 128 ? 128 : real;
      // (the split-K reduce places the rows itself; the direct GEMM -- shapes off the split-K path -- leaves compact rows to scatter)
      WRows wr{live_rows + c0, real, Meff, L.dhdrop, H, false};
      UIC_TRY(wgrad_multi(side ? L.slab2 : L.slab, L.slab_bytes, dt, off(L.dlogits, (size_t)c0 * V1p, dt), rows, dv.logit_wT, H, V1p, &dc, 1, s, false, &wr));
      if (wr.used) return UIC_OK;
      return uic_scatter_rows_launch(L.dhc + (size_t)c0 * H, live_rows + c0, L.dhdrop, Meff, real, (size_t)H * 4, s);
    }
    const size_t r0 = (size_t)t0 * N;
    const int rows = (t1 - t0) * N;
    const WDest d1{(nlh() ? L.dlh : L.dhdrop) + r0 * H, H, 0, H};
    UIC_TRY(wgrad_multi(side ? L.slab2 : L.slab, L.slab_bytes, dt, off(L.dlogits, r0 * V1p, dt), rows, dv.logit_wT, H, V1p, &d1, 1, s));
    // back through the hidden blocks: Dropout(0.5) + ReLU mask, then d x = d pre W
    for (int l = nlh() - 1; l >= 0; --l) {
      UIC_TRY(uic_relu_mask_bwd_launch(dt, L.dlh + r0 * H, off(L.lh[l], r0 * H, dt), (training & 1) ? 2.f : 1.f,
                                       offw(L.dlh_pre[l], r0 * H, dt), (size_t)rows * H, s));
      UicGemmParams g = gemm_base(dt, rows, H);
      add_seg(g, off(L.dlh_pre[l], r0 * H, dt), H, dv.logit_h_wT[l], H, H);
      g.C = (l ? L.dlh : L.dhdrop) + r0 * H; g.ldc = H; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    return UIC_OK;
  }
  // d W_logit, d b_logit over all executed steps (own scratch buffers: may run beside the BPTT loop)
  int logit_weight_grads(hipStream_t s, bool side = false) {
    if (compact) {
      const int K = live_pad();
      if (K == 0) {
        // (a batch without a single live position: no gradient; plain memsets -- V1 need not be a multiple of four)
        UIC_TRY(uic_check_hip(hipMemsetAsync(G->logit_w, 0, (size_t)V1 * H * 4, s), "hipMemsetAsync(d logit.weight)"));
        return uic_check_hip(hipMemsetAsync(G->logit_b, 0, (size_t)V1 * 4, s), "hipMemsetAsync(d logit.bias)");
      }
      const UicGemmTnSeg seg{L.hc, H, H};
      const WDest d1{G->logit_w, H, 0, H};
      UIC_TRY(wgrad_group(side ? L.slab2 : L.slab, L.dlogits, V1p, V1, &seg, 1, K, &d1, 1, s, false, L.tLA, L.tLB));
      return uic_colsum_launch(dt, L.dlogits, K, V1, V1p, G->logit_b, L.colscratchL, L.colscratch_floats, s);
    }
    {
      const UicGemmTnSeg seg{logit_in_all(), H, H};
      const WDest d1{G->logit_w, H, 0, H};
      UIC_TRY(wgrad_group(side ? L.slab2 : L.slab, L.dlogits, V1p, V1, &seg, 1, Meff, &d1, 1, s, false, L.tLA, L.tLB));
    }
    for (int l = 0; l < nlh(); ++l) {
      UIC_REQUIRE(G->logit_h_w[l] && G->logit_h_b[l], "backward: logit_layers=%d needs gradient tensors for hidden block %d", d.logit_layers, l);
      const UicGemmTnSeg seg{l ? L.lh[l - 1] : L.hdrop_all, H, H};
      const WDest d1{G->logit_h_w[l], H, 0, H};
      UIC_TRY(wgrad_group(side ? L.slab2 : L.slab, L.dlh_pre[l], H, H, &seg, 1, Meff, &d1, 1, s, false, L.tLA, L.tLB));
      UIC_TRY(uic_colsum_launch(dt, L.dlh_pre[l], Meff, H, H, G->logit_h_b[l], L.colscratchL, L.colscratch_floats, s));
    }
    return uic_colsum_launch(dt, L.dlogits, Meff, V1, V1p, G->logit_b, L.colscratchL, L.colscratch_floats, s);
  }

  // ---------------------------------------------------------------- backward
  // with_fwd_sync (the fused step, whose third stream runs this beside the prologue): the forward recurrence's sync block is
  // cleared here too, so that no memset node sits between the prologue's join and the persistent launch on the main stream
  bool fwd_sync_clean = false;
  // with_live (the fused step over a live-position list, whose BPTT is the launch chain): d hdrop is cleared in the same launch --
  // the positions the list leaves out have gradient zero, the chunks scatter the others into it
  int bwd_begin(hipStream_t s, bool with_fwd_sync = false, bool with_live = false) {
    if (bias_in_chunks()) UIC_TRY(uic_fill_value_launch(dt, L.ones_blk, L.ones_rows * 128, 1.f, s));
    bwd_launches = 0;
    fwd_sync_clean = with_fwd_sync;
    if (with_live && bwd_persist_ok()) { UIC_TRY(live_begin(s)); with_live = false; }      // (no free slot below)
    void* p2 = bwd_persist_ok() ? (void*)L.rnn_bwd_sync : with_live ? (void*)L.dhdrop : nullptr;
    const size_t b2 = bwd_persist_ok() ? L.rnn_bwd_sync_bytes : with_live ? (size_t)Meff * H * 4 : 0;
    return uic_zero4_launch(L.dc_att, NH * 4, L.dc_lang, NH * 4, p2, b2,
                            with_fwd_sync ? L.rnn_sync : nullptr, with_fwd_sync ? uic_rnn_persist_sync_bytes() : 0, s);
  }

  // BPTT of decode steps [t_lo, t_hi), latest first: ONE persistent launch (rnn_bwd_persist.hip) when the shapes allow, else
  // the chain of five launches per step.  The choice is constant within a step (bwd_persist_ok depends on dims only), and it has
  // to be: with bptt_split() > 0 the chain carries d h between decode steps in the split-K slabs (bp_slab1 / bp_slab2), not in
  // L.dx1 / columns [H, 3H) of dx2_all where the persistent kernel keeps it -- a chunk of one cannot hand over to the other.
  bool bwd_persist_ok() const { return (d.recurrence & UIC_REC_BWD_PERSIST) && uic_rnn_bwd_persist_eligible(dt, N, H, A, R); }
  int bwd_launches = 0;
  int bwd_steps(int t_lo, int t_hi, hipStream_t s) {
    if (!bwd_persist_ok()) {
      for (int t = t_hi - 1; t >= t_lo; --t) UIC_TRY(bwd_step(t, s));
      return UIC_OK;
    }
    UicRnnBwdParams p;
    memset(&p, 0, sizeof(p));
    p.N = N; p.R = R; p.t_lo = t_lo; p.t_hi = t_hi; p.first = t_hi == t_run;
    p.w2T = dv.w2T; p.w1recT = dv.w1recT; p.h2attT = dv.h2attT; p.w_alpha = w->alpha_w;
    p.p_att = L.patt; p.att = L.attp;
    p.gates1 = L.gates1; p.gates2 = L.gates2; p.c_att = L.c_att; p.c_lang = L.c_lang;
    p.att_h_all = L.atth_all; p.alpha_all = L.alpha_all; p.dhdrop = L.dhdrop;
    p.drop_p = drop_p; p.seed = seed;
    p.dg1_all = L.dg1_all; p.dg2_all = L.dg2_all; p.dx2_all = L.dx2_all; p.dx1 = L.dx1;
    p.dc_att = L.dc_att; p.dc_lang = L.dc_lang; p.de_all = L.de_all; p.datth_all = L.datth_all;
    UIC_REQUIRE(bwd_launches < d.T, "backward: more BPTT launches than decode steps");
    p.sync = L.rnn_bwd_sync + (size_t)bwd_launches * (uic_rnn_persist_sync_bytes() / 4);
    p.sync_zeroed = 1;                 // (bwd_begin zeroed every launch's block of this step)
    ++bwd_launches;
    p.force_safe = (d.recurrence & UIC_REC_SAFE) != 0; p.status = d.rnn_status;
    p.dbg = (d.recurrence & UIC_REC_STAMPS) ? L.rnn_bwd_dbg : nullptr; p.dbg_T = d.T;
    return uic_rnn_bwd_persist_launch(p, s);
  }

  // The two d x GEMMs of a BPTT step (640 x 1536 / 1024 outputs, K = 4H) run as BPTT_SPLIT K slices on the LDS-DMA kernel with
  // 128 x 128 tiles (half the operand traffic of the 64 x 64 skinny tiles, a quarter of the K rounds per workgroup); the slices
  // land in partial slabs and are summed by whoever reads them next -- the cell backward kernels and the attention backward
  // kernel (which leaves the summed d att_res where the deferred accumulation pass expects it): no reduction launch.
  // 15.2 -> 12.7 us and 14.5 -> 11.6 us per launch isolated (profiles/r03_v2_gemm_headroom.txt).  0: small problems, direct GEMMs.
  int bptt_split() const {
    const int rounds = H4 / (dt == UIC_BF16 ? 64 : 32);
    return (BPTT_SPLIT > 1 && N >= 256 && H % 128 == 0 && uic_gemm_glds_eligible(dt, H4) && rounds % BPTT_SPLIT == 0 && rounds / BPTT_SPLIT >= 4) ? BPTT_SPLIT : 0;
  }

  int bwd_step(int t, hipStream_t s) {
    const bool last = t == t_run - 1;
    const int S = bptt_split();
    float* dx2 = L.dx2_all + (size_t)t * N * 3 * H;
    const float* dx2_next = L.dx2_all + (size_t)(t + 1) * N * 3 * H;
    float* slab2 = L.bp_slab2[t & 1];
    const float* slab2_next = L.bp_slab2[(t + 1) & 1];
    const size_t st2 = (size_t)N * 3 * H, st1 = (size_t)N * 2 * H;
    {
      UicLstmBwdParams p;
      memset(&p, 0, sizeof(p));
      p.dtype = dt; p.M = N; p.H = H;
      p.dh0 = L.dhdrop + t * NH; p.lddh0 = H;
      p.drop_p = drop_p; p.seed = seed; p.site = UIC_SITE_OUT0 + (unsigned)t;
      if (!last && S) {
        p.slabA = slab2_next + 2 * H; p.ldA = 3 * H; p.nA = S; p.strideA = st2;
        p.slabB = L.bp_slab1; p.ldB = 2 * H; p.nB = S; p.strideB = st1;
      } else if (!last) { p.dh1 = dx2_next + 2 * H; p.lddh1 = 3 * H; p.dh2 = L.dx1; p.lddh2 = 2 * H; }
      p.dc = L.dc_lang; p.gates = off(L.gates2, (size_t)t * N * H4, dt);
      p.c_prev = L.c_lang + t * NH; p.c = L.c_lang + (t + 1) * NH;
      p.dgates = offw(L.dg2_all, (size_t)t * N * H4, dt);
      UIC_TRY(uic_lstm_bwd_launch(p, s));
    }
    {  // d[att_res | h_att | h_lang_prev] = dG2 [W_ih | W_hh]
      UicGemmParams g = gemm_base(dt, N, 3 * H);
      add_seg(g, off(L.dg2_all, (size_t)t * N * H4, dt), H4, dv.w2T, H4, H4);
      if (S) { g.slab = slab2; g.splitk = S; }
      else { g.C = dx2; g.ldc = 3 * H; g.flags = UIC_GEMM_OUT_F32; }
      UIC_TRY(uic_gemm_launch(g, s));
    }
    {
      UicAttnParams a;
      memset(&a, 0, sizeof(a));
      a.dtype = dt; a.N = N; a.R = R; a.A = A; a.H = H;
      a.att_h = L.atth_all + (size_t)t * N * A; a.p_att = L.patt; a.att = L.attp; a.w_alpha = w->alpha_w;
      a.alpha = L.alpha_all + (size_t)t * N * R;
      a.dctx = S ? slab2 : dx2; a.lddctx = 3 * H;
      if (S) { a.dctx_nslab = S; a.dctx_slab_stride = st2; a.dctx_sum = dx2; a.ld_dctx_sum = 3 * H; }
      a.de = L.de_all + (size_t)t * N * R;
      a.d_att_h = offw(L.datth_all, (size_t)t * N * A, dt);
#ifndef UIC_NO_DEAD_ROWS              // (A/B builds)
      if (have_len) { a.row_len = L.cap_len; a.step = t; }
#endif
      UIC_TRY(uic_attention_bwd_step_launch(a, s));
    }
    UicH2attCellParams hc;
    memset(&hc, 0, sizeof(hc));
    if (S) {   // d h_att = d_att_h W_h2att + the d x2 / d x1 slices, then the att_lstm cell backward: one launch (bptt_fused.hip)
      hc.dtype = dt; hc.N = N; hc.H = H; hc.A = A;
      hc.datth = off(L.datth_all, (size_t)t * N * A, dt); hc.h2attT = dv.h2attT;
      hc.slabA = slab2 + H; hc.ldA = 3 * H; hc.nA = S; hc.strideA = st2;
      if (!last) { hc.slabB = L.bp_slab1 + H; hc.ldB = 2 * H; hc.nB = S; hc.strideB = st1; }
      hc.dc = L.dc_att; hc.gates = off(L.gates1, (size_t)t * N * H4, dt);
      hc.c_prev = L.c_att + t * NH; hc.c = L.c_att + (t + 1) * NH;
      hc.dgates = offw(L.dg1_all, (size_t)t * N * H4, dt);
    }
    const bool fused_cell = UIC_BPTT_FUSE_CELL && S && uic_h2att_cell_bwd_eligible(hc);
    if (fused_cell) {
      UIC_TRY(uic_h2att_cell_bwd_launch(hc, s));
    } else {
    {  // dh_att += d_att_h W_h2att
      UicGemmParams g = gemm_base(dt, N, H);
      add_seg(g, off(L.datth_all, (size_t)t * N * A, dt), A, dv.h2attT, A, A);
      g.C = dx2 + H; g.ldc = 3 * H; g.flags = UIC_GEMM_OUT_F32 | (S ? 0 : UIC_GEMM_ACCUM);   // (split: the d x2 share is added by the cell backward below)
      UIC_TRY(uic_gemm_launch(g, s));
    }
    {
      UicLstmBwdParams p;
      memset(&p, 0, sizeof(p));
      p.dtype = dt; p.M = N; p.H = H;
      p.dh0 = dx2 + H; p.lddh0 = 3 * H;
      if (S) {
        p.slabA = slab2 + H; p.ldA = 3 * H; p.nA = S; p.strideA = st2;
        if (!last) { p.slabB = L.bp_slab1 + H; p.ldB = 2 * H; p.nB = S; p.strideB = st1; }
      } else if (!last) { p.dh1 = L.dx1 + H; p.lddh1 = 2 * H; }
      p.dc = L.dc_att; p.gates = off(L.gates1, (size_t)t * N * H4, dt);
      p.c_prev = L.c_att + t * NH; p.c = L.c_att + (t + 1) * NH;
      p.dgates = offw(L.dg1_all, (size_t)t * N * H4, dt);
      UIC_TRY(uic_lstm_bwd_launch(p, s));
    }
    }
    if (t > 0) {  // d[h_lang_prev | h_att_prev] = dG1 [W_ih[:, :H] | W_hh]
      UicGemmParams g = gemm_base(dt, N, 2 * H);
      add_seg(g, off(L.dg1_all, (size_t)t * N * H4, dt), H4, dv.w1recT, H4, H4);
      if (S) { g.slab = L.bp_slab1; g.splitk = S; }
      else { g.C = L.dx1; g.ldc = 2 * H; g.flags = UIC_GEMM_OUT_F32; }
      UIC_TRY(uic_gemm_launch(g, s));
    }
    return UIC_OK;
  }

  // weight gradients of everything except the logit layer, over all executed steps; split in two so that a
  // data-parallel caller can start exchanging the early group (LSTMs, embedding, fc_embed; with the logit layer
  // > 85 % of the bytes) while the late group (h2att, alpha_net, ctx2att, att_embed) is still being computed
  int bwd_epilogue(hipStream_t s) {
    UIC_TRY(bwd_epilogue_early(s));
    return bwd_epilogue_late(s);
  }
  int wgrad_group(float* slab, const void* A, int lda, int lrows, const UicGemmTnSeg* segs, int nseg, int rows, const WDest* dst,
                  int nd, hipStream_t s, bool accumulate, void* tA, void* tB) {
    return ::wgrad_group(slab, L.slab_bytes, dt, A, lda, lrows, segs, nseg, rows, dst, nd, s, accumulate, tA, tB);
  }

  // recurrent weight gradients (both LSTMs' weights and h2att) restricted to decode steps [t0, t1): one chunk of the
  // stacked-row GEMMs, accumulated into G unless `first`.  Used by the fused step on the side stream, chunk by chunk
  // behind the BPTT loop, so that only the last chunk's share is left when the loop ends.
  // sb (optional): a second stream for the att_lstm / h2att share, with scratch of its own (the GEMMs are independent)
  // own34: both shares away from the side stream -- lang_lstm on `s` with the third stream's scratch, att_lstm / h2att on `sb`
  // with the fourth's (the default order of the fused step: the side stream keeps the logit layer)
  int wgrad_chunk(int t0, int t1, bool first, hipStream_t s, hipStream_t sb = nullptr, bool own34 = false) {
    float* const slabA = own34 ? L.slab3 : L.slab2;
    void* const tAA = own34 ? L.tTA : L.tSA;
    void* const tAB = own34 ? L.tTB : L.tSB;
    float* const slabB = own34 ? L.slab4 : sb ? L.slab3 : L.slab2;
    void* const tBA = own34 ? L.tUA : sb ? L.tTA : L.tSA;
    void* const tBB = own34 ? L.tUB : sb ? L.tTB : L.tSB;
    if (!sb) sb = s;
    const int rows = (t1 - t0) * N;
    const size_t r0 = (size_t)t0 * N;
    const void* ctx = off(L.ctx_all, r0 * H, dt);
    const void* h_att_new = off(L.h_att, NH + r0 * H, dt);
    const void* h_att_prev = off(L.h_att, r0 * H, dt);
    const void* h_lang_prev = off(L.h_lang, r0 * H, dt);
    if (bias_in_chunks()) {
      // the bias gradients (column sums of dG) ride in the same GEMMs: a fourth "input" segment of ones whose first column's
      // weight gradient IS the column sum -- one more column tile per row tile instead of two column-sum passes in the tail
      const UicGemmTnSeg segs[4] = {{ctx, H, H}, {h_att_new, H, H}, {h_lang_prev, H, H}, {L.ones_blk, 128, 128}};
      const WDest dd[3] = {{G->lang_lstm_w_ih, 2 * H, 0, 2 * H}, {G->lang_lstm_w_hh, H, 2 * H, H}, {G->lang_lstm_b_ih, 1, 3 * H, 1}};
      UIC_TRY(wgrad_group(slabA, off(L.dg2_all, r0 * H4, dt), H4, H4, segs, 4, rows, dd, 3, s, !first, tAA, tAB));
    } else {  // lang_lstm: dG2^T x [att_res | h_att | h_lang_prev]
      const UicGemmTnSeg segs[3] = {{ctx, H, H}, {h_att_new, H, H}, {h_lang_prev, H, H}};
      const WDest dd[2] = {{G->lang_lstm_w_ih, 2 * H, 0, 2 * H}, {G->lang_lstm_w_hh, H, 2 * H, H}};
      UIC_TRY(wgrad_group(slabA, off(L.dg2_all, r0 * H4, dt), H4, H4, segs, 3, rows, dd, 2, s, !first, tAA, tAB));
    }
    if (bias_in_chunks() && A % 128 == 0) {  // h2att: d_att_h^T x [h_att | ones]
      const UicGemmTnSeg segs[2] = {{h_att_new, H, H}, {L.ones_blk, 128, 128}};
      const WDest dd[2] = {{G->h2att_w, H, 0, H}, {G->h2att_b, 1, H, 1}};
      UIC_TRY(wgrad_group(slabB, off(L.datth_all, r0 * A, dt), A, A, segs, 2, rows, dd, 2, sb, !first, tBA, tBB));
    } else {  // h2att: d_att_h^T x h_att
      const UicGemmTnSeg seg{h_att_new, H, H};
      const WDest d1{G->h2att_w, H, 0, H};
      UIC_TRY(wgrad_group(slabB, off(L.datth_all, r0 * A, dt), A, A, &seg, 1, rows, &d1, 1, sb, !first, tBA, tBB));
    }
    if (bias_in_chunks()) {
      const UicGemmTnSeg segs[4] = {{h_lang_prev, H, H}, {off(L.xt_all, r0 * E, dt), E, E}, {h_att_prev, H, H}, {L.ones_blk, 128, 128}};
      const WDest dd[4] = {{G->att_lstm_w_ih, ldih, 0, H}, {G->att_lstm_w_ih + 2 * H, ldih, H, E}, {G->att_lstm_w_hh, H, H + E, H},
                           {G->att_lstm_b_ih, 1, 2 * H + E, 1}};
      UIC_TRY(wgrad_group(slabB, off(L.dg1_all, r0 * H4, dt), H4, H4, segs, 4, rows, dd, 4, sb, !first, tBA, tBB));
    } else {  // att_lstm: dG1^T x [h_lang_prev | xt | h_att_prev]
      const UicGemmTnSeg segs[3] = {{h_lang_prev, H, H}, {off(L.xt_all, r0 * E, dt), E, E}, {h_att_prev, H, H}};
      const WDest dd[3] = {{G->att_lstm_w_ih, ldih, 0, H}, {G->att_lstm_w_ih + 2 * H, ldih, H, E}, {G->att_lstm_w_hh, H, H + E, H}};
      UIC_TRY(wgrad_group(slabB, off(L.dg1_all, r0 * H4, dt), H4, H4, segs, 3, rows, dd, 3, sb, !first, tBA, tBB));
    }
    return UIC_OK;
  }
  // true when EVERY chunk's LSTM weight-gradient GEMM takes the direct transposing-read (TN) path with room for one more
  // 128-column segment: bf16, whole 64-row K rounds per decode step, 128-multiples everywhere
  bool bias_in_chunks() const {
    return dt == UIC_BF16 && N % 64 == 0 && H % 128 == 0 && E % 128 == 0 && (3 * H + 128) / 128 * (H4 / 128) >= 160 && (2 * H + E + 128) / 128 * (H4 / 128) >= 160;
  }

  // chunked == true: wgrad_chunk already produced the LSTM / h2att weight gradients
  // side == true: runs on the fused step's side stream with that stream's own scratch buffers
  // The two big tensors of the early group, on their own so that the fused step can finish them FIRST after the BPTT loop
  // (with gradient group 1, uic_topdown_grad_ready_wait):
  //   fc_cols_grad: dGfc = sum_t dG1_t (left in L.dgfc) -> the fc' columns of att_lstm.weight_ih, which completes that matrix;
  //   embed_grad(half): d xt = dG1 Wx of the decode steps of one half of the bucketed position list (embed_split; 0: all
  //                     steps at once) -> their share of the embedding table.
  int fc_cols_grad(hipStream_t s, float* slab, void* tA, void* tB) {
    UIC_TRY(uic_sum_steps_launch(dt, L.dg1_all, t_run, (size_t)N * H4, L.dgfc, s));
    const UicGemmTnSeg seg{L.fcp, H, H};
    const WDest d1{G->att_lstm_w_ih + H, ldih, 0, H};
    return wgrad_group(slab, L.dgfc, H4, H4, &seg, 1, N, &d1, 1, s, false, tA, tB);
  }
  int embed_grad(int half, hipStream_t s) {
    const int t0 = embed_split && half ? embed_split : 0, t1 = embed_split && !half ? embed_split : t_run;
    const size_t r0 = (size_t)t0 * N;
    UicGemmParams g = gemm_base(dt, (t1 - t0) * N, E);
    add_seg(g, off(L.dg1_all, r0 * H4, dt), H4, dv.wxT, H4, H4);
    g.C = L.dxt + r0 * E; g.ldc = E; g.flags = UIC_GEMM_OUT_F32;
    UIC_TRY(uic_gemm_launch(g, s));
    if (!embed_prepared) {
      UIC_REQUIRE(!embed_split, "embed_grad: the position list was not prepared for split=%d", embed_split);
      UIC_TRY(uic_embed_bwd_sorted_prepare(embed_tokens(), embed_ldtok(), N, t_run, V1, E, G->embed_w, L.embed_scratch, s));
      embed_prepared = true;
    }
    return uic_embed_bwd_sorted_gather(dt, L.dxt, L.xt_all, embed_tokens(), embed_ldtok(), N, t_run, V1, E, drop_p, -1, G->embed_w,
                                       L.embed_scratch, s, embed_split, half);
  }
  // first_done == true: the caller already ran fc_cols_grad and embed_grad (the fused step, right behind the last chunk)
  // bias_copies == false: the caller copies bias_ih -> bias_hh of the two cells itself (the fused step's default order: the
  // chunks' gradients -- the bias columns among them -- come from ANOTHER stream and are joined after this call)
  int bias_hh_copies(hipStream_t s) {
    UIC_TRY(uic_copy_launch(G->lang_lstm_b_hh, G->lang_lstm_b_ih, (size_t)H4 * 4, s));
    return uic_copy_launch(G->att_lstm_b_hh, G->att_lstm_b_ih, (size_t)H4 * 4, s);
  }
  int bwd_epilogue_early(hipStream_t s, bool chunked = false, bool side = false, bool first_done = false, bool bias_copies = true) {
    void* const tA = side ? L.tSA : L.tA;
    void* const tB = side ? L.tSB : L.tB;
    float* const colscratch = side ? L.colscratchL : L.colscratch;
    float* const slab = side ? L.slab2 : L.slab;
    if (chunked && !(bias_in_chunks() && A % 128 == 0))   // h2att.bias belongs to the early group then (its weight came from wgrad_chunk)
      UIC_TRY(uic_colsum_launch(dt, L.datth_all, Meff, A, A, G->h2att_b, colscratch, L.colscratch_floats, s));
    // per LSTM ONE GEMM dG^T [4H, T*N] x [stacked inputs]^T; lang_lstm inputs [att_res | h_att | h_lang_prev]
    if (!chunked) {
      const UicGemmTnSeg segs[3] = {{L.ctx_all, H, H}, {off(L.h_att, NH, dt), H, H}, {L.h_lang, H, H}};
      const WDest dd[2] = {{G->lang_lstm_w_ih, 2 * H, 0, 2 * H}, {G->lang_lstm_w_hh, H, 2 * H, H}};
      UIC_TRY(wgrad_group(slab, L.dg2_all, H4, H4, segs, 3, Meff, dd, 2, s, false, tA, tB));
    }
    if (!(chunked && bias_in_chunks()))
      UIC_TRY(uic_colsum_launch(dt, L.dg2_all, Meff, H4, H4, G->lang_lstm_b_ih, colscratch, L.colscratch_floats, s));
    if (bias_copies) UIC_TRY(uic_copy_launch(G->lang_lstm_b_hh, G->lang_lstm_b_ih, (size_t)H4 * 4, s));
    // att_lstm inputs [h_lang_prev | xt | h_att_prev]  (the fc' columns are handled below from dGfc)
    if (!chunked) {
      const UicGemmTnSeg segs[3] = {{L.h_lang, H, H}, {L.xt_all, E, E}, {L.h_att, H, H}};
      const WDest dd[3] = {{G->att_lstm_w_ih, ldih, 0, H}, {G->att_lstm_w_ih + 2 * H, ldih, H, E}, {G->att_lstm_w_hh, H, H + E, H}};
      UIC_TRY(wgrad_group(slab, L.dg1_all, H4, H4, segs, 3, Meff, dd, 3, s, false, tA, tB));
    }
    if (!(chunked && bias_in_chunks()))
      UIC_TRY(uic_colsum_launch(dt, L.dg1_all, Meff, H4, H4, G->att_lstm_b_ih, colscratch, L.colscratch_floats, s));
    if (bias_copies) UIC_TRY(uic_copy_launch(G->att_lstm_b_hh, G->att_lstm_b_ih, (size_t)H4 * 4, s));
    if (!first_done) {
      UIC_TRY(fc_cols_grad(s, slab, tA, tB));
      UIC_TRY(embed_grad(0, s));
    }
    {  // d fc' from dGfc (left in L.dgfc by fc_cols_grad)
      UicGemmParams g = gemm_base(dt, N, H);
      add_seg(g, L.dgfc, H4, dv.wfcpT, H4, H4);
      g.C = L.dfcp; g.ldc = H; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    UIC_TRY(uic_relu_mask_bwd_launch(dt, L.dfcp, L.fcp, inv_keep, L.dfcpre, NH, s));
    {
      const UicGemmTnSeg seg{fc_in, Dfc, Dfc};
      const WDest d1{G->fc_w, Dfc, 0, Dfc};
      UIC_TRY(wgrad_group(slab, L.dfcpre, H, H, &seg, 1, N, &d1, 1, s, false, tA, tB));
    }
    UIC_TRY(uic_colsum_launch(dt, L.dfcpre, N, H, H, G->fc_b, colscratch, L.colscratch_floats, s));
    if (b->d_fc_feats) {   // optional: d fc_feats = d fc_pre W_fc, the S caption rows of an image summed (row i S + k of d fc_pre: lda = S H)
      const int S = d.seq_per_img > 1 ? d.seq_per_img : 1;
      UIC_TRY(uic_transpose_launch(dt, dv.fc_w, H, Dfc, Dfc, L.fcwT, H, s));
      for (int k = 0; k < S; ++k) {
        UicGemmParams g = gemm_base(dt, N / S, Dfc);
        add_seg(g, off(L.dfcpre, (size_t)k * H, dt), S * H, L.fcwT, H, H);
        g.C = b->d_fc_feats; g.ldc = Dfc; g.flags = UIC_GEMM_OUT_F32 | (k ? UIC_GEMM_ACCUM : 0);
        UIC_TRY(uic_gemm_launch(g, s));
      }
    }
    return UIC_OK;
  }
  // part: 0 = everything, 1 = only the deferred attention accumulation, 2 = everything else.  (Measured: running the
  // accumulation -- whole-CU workgroups -- before releasing the side stream's tail is 3 % SLOWER than letting both run.)
  int bwd_epilogue_late(hipStream_t s, bool chunked = false, int part = 0) {
    // h2att
    if (!chunked && part != 1) {
      const UicGemmTnSeg seg{off(L.h_att, NH, dt), H, H};
      const WDest d1{G->h2att_w, H, 0, H};
      UIC_TRY(wgrad_group(L.slab, L.datth_all, A, A, &seg, 1, Meff, &d1, 1, s, false, L.tA, L.tB));
      UIC_TRY(uic_colsum_launch(dt, L.datth_all, Meff, A, A, G->h2att_b, L.colscratch, L.colscratch_floats, s));
    }
    if (part != 2) {  // attention: deferred accumulation over steps
      UicAttnAccumParams a;
      memset(&a, 0, sizeof(a));
      a.dtype = dt; a.N = N; a.R = R; a.A = A; a.H = H; a.T = t_run;
      a.att_h_all = L.atth_all; a.alpha_all = L.alpha_all; a.de_all = L.de_all;
      a.dctx_all = L.dx2_all; a.lddctx = 3 * H; a.dctx_step_stride = (size_t)N * 3 * H;
      a.p_att = L.patt; a.w_alpha = w->alpha_w;
      a.d_att = L.d_att; a.d_p_att = L.d_patt; a.d_walpha_part = L.dwalpha_part;
#ifndef UIC_NO_DEAD_ACCUM             // (A/B builds)
      if (have_len) a.row_len = L.cap_len;     // (steps behind a caption's end add zeros)
#endif
      UIC_TRY(uic_attention_bwd_accum_launch(a, s));
      UIC_TRY(uic_colsum_small_launch(L.dwalpha_part, N, A + 1, A, G->alpha_w, G->alpha_b, s));
    }
    if (part == 1) return UIC_OK;
    // ctx2att (bias gradient as the ones segment's first column where the TN path takes it, else a column-sum pass)
    const bool ones_ok = bias_in_chunks() && A % 128 == 0 && NR % 64 == 0;
    if (ones_ok) {
      const UicGemmTnSeg segs[2] = {{L.attp, H, H}, {L.ones_blk, 128, 128}};
      const WDest dd[2] = {{G->ctx2att_w, H, 0, H}, {G->ctx2att_b, 1, H, 1}};
      UIC_TRY(wgrad_group(L.slab, L.d_patt, A, A, segs, 2, NR, dd, 2, s, false, L.tA, L.tB));
    } else {
      const UicGemmTnSeg seg{L.attp, H, H};
      const WDest d1{G->ctx2att_w, H, 0, H};
      UIC_TRY(wgrad_group(L.slab, L.d_patt, A, A, &seg, 1, NR, &d1, 1, s, false, L.tA, L.tB));
      UIC_TRY(uic_colsum_launch(dt, L.d_patt, NR, A, A, G->ctx2att_b, L.colscratch, L.colscratch_floats, s));
    }
    // d att' = (the attention's own share, in L.d_att) + d p_att W_ctx2att; without BatchNorm behind it and with one feature row
    // per caption row, the backward of att_embed's ReLU + dropout rides on this GEMM's epilogue: it writes L.d_pre (the weight
    // gradient's bf16 operand) directly instead of the f32 sum that a separate pass then masks (53 us of the main stream's tail)
    bool relu_fused = false;
    {
      UicGemmParams g = gemm_base(dt, NR, H);
      add_seg(g, L.d_patt, A, dv.ctx2attT, A, A);
      g.C = L.d_pre; g.ldc = H; g.acc_src = L.d_att; g.ld_acc_src = H; g.mask_act = L.attp; g.ld_mask_act = H; g.mask_scale = inv_keep;
      relu_fused = d.use_bn != 2 && d.seq_per_img <= 1 && dt == UIC_BF16 && uic_gemm_pp_eligible(g);
      if (!relu_fused) {
        g.C = L.d_att; g.flags = UIC_GEMM_OUT_F32 | UIC_GEMM_ACCUM;
        g.acc_src = nullptr; g.mask_act = nullptr;
      }
      UIC_TRY(uic_gemm_launch(g, s));
    }
    // att_embed (padded regions have att' = 0 -> zero gradient, as pack_wrapper never touched them)
    const void* act = L.attp;
    if (d.use_bn == 2) {   // through BatchNorm1d(H): d_att <- d y (in place), grads of its affine parameters
      UIC_REQUIRE(G->att_bn4_w && G->att_bn4_b, "backward: use_bn=2 needs gradient tensors for att_embed.4");
      UIC_TRY(uic_bn_bwd_launch(dt, L.d_att, L.ybn, NR, R, H, b->att_masks ? (d.seq_per_img > 1 ? L.row_len_rep : L.row_len) : nullptr, L.bn_stat4, w->att_bn4_w,
                                training & 1, L.bn_part, L.bn_red, G->att_bn4_w, G->att_bn4_b, s));
      act = L.ybn;
    }
    // seq_per_img > 1 without BN: the S caption rows of an image share the Linear's input, so their gradients are summed
    // first and the weight-gradient GEMM runs over the per-image rows
    const bool fold = d.seq_per_img > 1;
    const int NRa = fold ? NR / d.seq_per_img : NR;
    if (fold)
      UIC_TRY(uic_relu_mask_bwd_fold_launch(dt, L.d_att, act, inv_keep, L.d_pre, N / d.seq_per_img, d.seq_per_img, (size_t)R * H, s));
    else if (!relu_fused)
      UIC_TRY(uic_relu_mask_bwd_launch(dt, L.d_att, act, inv_keep, L.d_pre, (size_t)NR * H, s));
    if (ones_ok && D % 128 == 0 && NRa % 64 == 0) {
      const UicGemmTnSeg segs[2] = {{att_in, D, D}, {L.ones_blk, 128, 128}};
      const WDest dd[2] = {{G->att_w, D, 0, D}, {G->att_b, 1, D, 1}};
      UIC_TRY(wgrad_group(L.slab, L.d_pre, H, H, segs, 2, NRa, dd, 2, s, false, L.tA, L.tB));
    } else {
      const UicGemmTnSeg seg{att_in, D, D};
      const WDest d1{G->att_w, D, 0, D};
      UIC_TRY(wgrad_group(L.slab, L.d_pre, H, H, &seg, 1, NRa, &d1, 1, s, false, L.tA, L.tB));
      UIC_TRY(uic_colsum_launch(dt, L.d_pre, NRa, H, H, G->att_b, L.colscratch, L.colscratch_floats, s));
    }
    if (b->d_att_feats) {  // optional: d att_feats = d_pre W_att  (one row per image region; d_pre is already folded over seq_per_img)
      UIC_TRY(uic_transpose_launch(dt, dv.att_w, H, D, D, L.attwT, H, s));
      UicGemmParams g = gemm_base(dt, NRa, D);
      add_seg(g, L.d_pre, H, L.attwT, H, H);
      g.C = b->d_att_feats; g.ldc = D; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
    if (d.use_bn) {        // G->att_w holds dW' = d_pre^T xhat: unfold the BatchNorm1d(D) affine part (batchnorm.hip)
      UIC_REQUIRE(G->att_bn0_w && G->att_bn0_b, "backward: use_bn needs gradient tensors for att_embed.0");
      UIC_TRY(uic_bn_fold_grad_launch(w->att_w, w->att_bn0_w, w->att_bn0_b, G->att_w, G->att_b, H, D, G->att_bn0_w, G->att_bn0_b, s));
    }
    return UIC_OK;
  }
};

// One decode step of AttModel.get_logprobs_state (:158-165) on the sampling buffers: embedding of L.s_it, both LSTM cells,
// attention, logits into L.s_logits; recurrent state read from slot `cur`, written to slot `nxt`.
int decode_step(const uic_topdown_dims& d, const uic_topdown_weights* w, const Derived& dv, const uic_topdown_batch* b, const Layout& L,
                int cur, int nxt, int t, float drop_p, unsigned seed, hipStream_t s, bool train_mode = false, bool xt_ready = false) {
  const int dt = d.dtype;
  const int N = d.N, H = d.H, E = d.E, V1 = d.V1;
  const int V1p = (int)vpad(V1), H4 = 4 * H, ldih = E + 2 * H;
  // (xt_ready: the sampling kernel of the previous step already wrote this step's embedding rows into L.s_xt)
  if (!xt_ready) UIC_TRY(uic_embed_fwd_t_launch(dt, dv.embed_w, dt, V1, E, L.s_it, 1, N, 1, drop_p, seed, UIC_SITE_EMBED, (size_t)t * N * E, 1, L.s_xt, s));
  {
    UicGemmParams g = gemm_base(dt, N, H4);
    g.lstm = 1; g.H = H;
    add_seg(g, L.s_h_lang[cur], H, dv.att_w_ih, ldih, H);
    add_seg(g, L.fcp, H, off(dv.att_w_ih, H, dt), ldih, H);
    add_seg(g, L.s_xt, E, off(dv.att_w_ih, 2 * H, dt), ldih, E);
    add_seg(g, L.s_h_att[cur], H, dv.att_w_hh, H, H);
    g.bias = w->att_lstm_b_ih; g.bias2 = w->att_lstm_b_hh;
    g.c_prev = L.s_c_att[cur]; g.c_out = L.s_c_att[nxt];
    g.h_out = L.s_h_att[nxt]; g.ldh = H;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  UIC_TRY(attention_step(d, w, dv, b, L, L.s_h_att[nxt], L.s_atth, L.s_alpha, L.s_ctx, s));
  {
    UicGemmParams g = gemm_base(dt, N, H4);
    g.lstm = 1; g.H = H;
    add_seg(g, L.s_ctx, H, dv.lang_w_ih, 2 * H, H);
    add_seg(g, L.s_h_att[nxt], H, off(dv.lang_w_ih, H, dt), 2 * H, H);
    add_seg(g, L.s_h_lang[cur], H, dv.lang_w_hh, H, H);
    g.bias = w->lang_lstm_b_ih; g.bias2 = w->lang_lstm_b_hh;
    g.c_prev = L.s_c_lang[cur]; g.c_out = L.s_c_lang[nxt];
    g.h_out = L.s_h_lang[nxt]; g.ldh = H;
    g.h_drop = L.s_hdrop; g.ldhd = H;        // dropout(h_lang) feeds the logit layer (AttModel.py:443)
    g.drop_p = drop_p; g.seed = seed; g.site = UIC_SITE_OUT0 + (unsigned)t;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  const void* x = L.s_hdrop;
  for (int l = 0; l + 1 < d.logit_layers; ++l) {      // hidden logit blocks (AttModel.py:90-91), same masks as a training pass of step t
    UicGemmParams g = gemm_base(dt, N, H);
    add_seg(g, x, H, dv.logit_h_w[l], H, H);
    g.C = L.s_lh[l]; g.ldc = H; g.bias = w->logit_h_b[l]; g.flags = UIC_GEMM_RELU;
    if (train_mode) { g.drop_p = 0.5f; g.seed = seed; g.site = UIC_SITE_LOGIT_H0 + (unsigned)l; g.drop_row0 = t * N; }
    UIC_TRY(uic_gemm_launch(g, s));
    x = L.s_lh[l];
  }
  UicGemmParams g = gemm_base(dt, N, V1);
  add_seg(g, x, H, dv.logit_w, H, H);
  g.C = L.s_logits; g.ldc = V1p; g.bias = w->logit_b; g.flags = UIC_GEMM_OUT_F32;
  return uic_gemm_launch(g, s);
}

}  // namespace

extern "C" {

int uic_topdown_forward(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                        const uic_topdown_batch* b, int32_t t_run, int32_t training, uint32_t seed,
                        void* workspace, float* logprobs_out, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived && b && workspace, "forward: null pointer");
  UIC_REQUIRE(b->fc_feats && b->att_feats && b->labels, "forward: batch needs fc_feats, att_feats and labels");
  UIC_REQUIRE(t_run >= 1 && t_run <= d->T, "forward: t_run=%d outside [1,%d]", t_run, d->T);
  UIC_REQUIRE(b->ld_labels >= t_run, "forward: labels have %d columns, need %d", b->ld_labels, t_run);
  hipStream_t s = (hipStream_t)stream;
  Step st;
  st.init(d, w, derived, b, t_run, training, seed, workspace, nullptr);
  UIC_TRY(wait_refresh(s));                           // (the embedding table's operand copy is made on the side stream)
  UIC_TRY(st.fwd_prologue(s));
  UIC_TRY(st.fwd_steps(0, t_run, s));
  UIC_TRY(st.logits_rows(0, t_run, s));
  if (logprobs_out) UIC_TRY(st.xe_rows(0, t_run, nullptr, logprobs_out, 0, s));
  return UIC_OK;
}

int uic_topdown_xe_loss(const uic_topdown_dims* d, const uic_topdown_batch* b, int32_t t_run, void* workspace,
                        const float* inv_den, float* loss_out, float* den_out, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(b && workspace && b->labels && b->masks && loss_out, "xe_loss: null pointer");
  UIC_REQUIRE(t_run >= 1 && t_run <= d->T, "xe_loss: t_run=%d outside [1,%d]", t_run, d->T);
  UIC_REQUIRE(b->ld_labels >= d->T + 1 && b->ld_masks >= d->T + 1, "xe_loss: labels/masks need %d columns", d->T + 1);
  hipStream_t s = (hipStream_t)stream;
  uic_topdown_weights none;
  memset(&none, 0, sizeof(none));
  Step st;
  st.init(d, &none, nullptr, b, t_run, 0, 0, workspace, nullptr);
  // denominator: sum of masks[:, 1:T+1] over ALL T columns (criterion.py:146-149), also after an early break
  UIC_TRY(uic_masked_sum_launch(nullptr, b->masks, b->ld_masks, 1, d->N, d->T, st.L.scalars, st.L.scalars + 1, s));
  const float* inv = inv_den ? inv_den : st.L.scalars + 1;
  UIC_TRY(st.xe_rows(0, t_run, inv, nullptr, 1, s));
  UIC_TRY(uic_reduce_sum_launch(st.L.row_loss, (size_t)t_run * d->N, 0.f, inv, loss_out, s));
  if (den_out) UIC_TRY(uic_copy_launch(den_out, st.L.scalars, 4, s));
  return UIC_OK;
}

int uic_topdown_backward(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                         const uic_topdown_batch* b, int32_t t_run, int32_t training, uint32_t seed,
                         void* workspace, const float* dlogprobs, const float* logprobs,
                         const uic_topdown_weights* G, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived && b && workspace && G, "backward: null pointer");
  UIC_REQUIRE(t_run >= 1 && t_run <= d->T, "backward: t_run=%d outside [1,%d]", t_run, d->T);
  UIC_REQUIRE(!dlogprobs || logprobs, "backward: dlogprobs needs the forward log-probs");
  UIC_REQUIRE(!b->d_att_feats || !d->use_bn, "backward: d_att_feats is not available through att_embed's BatchNorm (use_bn=%d)", d->use_bn);
  hipStream_t s = (hipStream_t)stream;
  Step st;
  st.init(d, w, derived, b, t_run, training, seed, workspace, G);
  UIC_TRY(wait_refresh(s));
  if (dlogprobs)
    UIC_TRY(uic_logsoftmax_bwd_launch(st.dt, st.L.dlogits, st.Meff, st.V1, st.V1p, st.N, dlogprobs, (size_t)st.V1,
                                      (size_t)d->T * st.V1, logprobs, s));
  UIC_TRY(st.dh_rows(0, t_run, s));
  UIC_TRY(st.logit_weight_grads(s));
  UIC_TRY(st.bwd_begin(s));
  UIC_TRY(st.bwd_steps(0, t_run, s));
  return st.bwd_epilogue(s);
}

int uic_topdown_xe_train_step(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                              const uic_topdown_batch* b, int32_t t_run, int32_t training, uint32_t seed,
                              void* workspace, const float* inv_den, float* loss_out, float* den_out,
                              const uic_topdown_weights* G, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived && b && workspace && G && loss_out, "xe_train_step: null pointer");
  UIC_REQUIRE(b->fc_feats && b->att_feats && b->labels && (b->masks || b->grad_scale), "xe_train_step: batch needs features, labels and masks (or grad_scale)");
  UIC_REQUIRE(t_run >= 1 && t_run <= d->T, "xe_train_step: t_run=%d outside [1,%d]", t_run, d->T);
  UIC_REQUIRE(b->ld_labels >= d->T + 1 && (!b->masks || b->ld_masks >= d->T + 1), "xe_train_step: labels/masks need %d columns", d->T + 1);
  UIC_REQUIRE(!b->grad_scale || b->ld_grad_scale >= t_run, "xe_train_step: grad_scale needs %d columns", t_run);
  UIC_REQUIRE(!b->d_att_feats || !d->use_bn, "xe_train_step: d_att_feats is not available through att_embed's BatchNorm (use_bn=%d)", d->use_bn);
  hipStream_t s = (hipStream_t)stream;
  SideStream* ss = nullptr;
  UIC_TRY(get_side(&ss));
  hipStream_t s2 = ss->stream;
  hipStream_t s3 = ss->stream3;
  hipStream_t s4 = ss->stream4;
  Step st;
  st.init(d, w, derived, b, t_run, training, seed, workspace, G);
  // The fused step's BPTT is always the launch chain: the persistent BPTT kernel holds every CU, which serialises the side
  // stream's GEMMs behind it (measured 3.81 vs 3.48 ms, profiles/r03_v4_rnn_bwd_probe.txt); UIC_REC_BWD_PERSIST selects it for
  // the single-stream uic_topdown_backward call only, where nothing runs beside it.
  st.d.recurrence &= ~UIC_REC_BWD_PERSIST;
  st.init_live();
  const bool early = st.early_grads();
  ss->embed_recorded = false;
  // UIC_REC_COMM_STREAM: a communication stream of the caller is busy beside this step (data parallel).  The chip dispatches
  // from THREE busy hardware queues at full speed; a fourth non-empty queue -- even one wave that only waits -- makes the BPTT
  // loop's 85 dependent launches 1.5-2.5x slower (tools/queue_probe.py, profiles/r05_*_queue_probe.txt).  So with the caller's
  // queue counted the step keeps to two of its own while the loop runs: the chunks' weight gradients go back to the side stream,
  // behind the logit layer (round 4's order), instead of stream 3.
  const bool comm = (d->recurrence & UIC_REC_COMM_STREAM) != 0;
  const int CH = WG_CHUNK;                            // decode steps per hand-off to the side stream
  // chunk c = decode steps [cb[c], cb[c + 1]).  Default: chunks of CH from step 0 (the last one -- the one the BPTT loop starts
  // from -- is t_run % CH steps long: ONE at the reference's 17 steps).  UIC_KNOB_SHORT_FIRST (measurement knob): the FIRST chunk is
  // one step too -- the BPTT loop ends on it, so that all but 1/t_run of the recurrent weight gradients is done when the loop ends.
  int cb[MAX_CHUNKS + 1];
  int nchunk = 0;
  {
    UIC_REQUIRE((t_run + CH - 1) / CH + 2 <= MAX_CHUNKS, "xe_train_step: too many decode steps (%d)", t_run);
    int t = 0;
    cb[0] = 0;
    const bool short_first = (d->recurrence & UIC_KNOB_SHORT_FIRST) && !early && t_run > CH + 1;
    if (short_first) { cb[++nchunk] = 1; t = 1; }
    while (t < t_run) {
      int len = CH < t_run - t ? CH : t_run - t;
      if (short_first && t_run - t > 1 && t + len == t_run) len = t_run - t - 1;   // keep the loop's first chunk short too
      t += len;
      cb[++nchunk] = t;
    }
  }
#define UIC_HIP(expr) UIC_TRY(uic_check_hip((expr), #expr))
#define UIC_MARK(i, strm) do { if (ss->marks_on) UIC_HIP(hipEventRecord(ss->mark[i], strm)); } while (0)

  UIC_MARK(0, s);
  // main: features, recurrence; hands each finished chunk of steps to the side stream.  ev_den: the step has begun (whatever
  // the caller enqueued on `s` before it is done) -- the side streams start from there.
  const float* inv = inv_den ? inv_den : st.L.scalars + 1;
  UIC_HIP(hipEventRecord(ss->ev_den, s));
  UIC_HIP(hipStreamWaitEvent(s2, ss->ev_den, 0));
  // side: the loss denominator (the criterion's first use of it is on this stream, after the recurrence).  First on its stream:
  // behind the branch below it would be dispatched beside the persistent recurrence and wait for a CU until that ends.
  // (A resumed step has no prologue after which the main stream waits for the side stream, and the loop's first logit chunk --
  // its criterion reads the denominator -- runs on the main stream: there the denominator is made on the main stream.)
  if (b->masks) UIC_TRY(uic_masked_sum_launch(nullptr, b->masks, b->ld_masks, 1, d->N, d->T, st.L.scalars, st.L.scalars + 1,
                                              (training & 4) ? s : s2));
  // (the live list, when the caller left it to the library: on the side stream ahead of its prologue branch, whose event the main
  // stream waits for before the recurrence -- every logit chunk on either stream is ordered behind it)
  // (the whole recurrence as one launch of the weight-stationary kernel: it fills the compact operand itself)
#ifndef UIC_NO_FUSED_GATHER          // (A/B builds: the gather launches)
  st.fused_gather = st.compact && st.build_live && !(training & 4) && st.persist_ok() && st.dt == UIC_BF16;
#endif
  if (st.compact && st.build_live) UIC_TRY(st.live_build((training & 4) ? s : s2));
  else if (b->grad_scale) UIC_TRY(st.len_build((training & 4) ? s : s2));
  // (not for a plain masked step without counts: it stays bit-equal to the three separate API calls, whose backward pass sums over
  // every step -- tests/test_gpu_topdown.py)
  st.have_len = (st.compact && st.build_live) || b->grad_scale;
  // the prologue's two independent branches side by side: att_embed + ctx2att here, fc_embed + embedding + the batched
  // input GEMM on the side stream (idle until the recurrence is through; its part of the weight refresh comes first there)
  // training bit 2: the workspace already holds this forward pass (uic_topdown_sample_train drew b->labels with these
  // weights, this seed and these dims): the step starts at the criterion
  const bool resume = (training & 4) != 0;
  bool bwd_begun = false, live_begun = false;
  UIC_REQUIRE(!resume || !st.ss_on(), "xe_train_step: a resumed step cannot use scheduled sampling");
  if (!resume) {
    // three branches: att_embed + ctx2att (main), embedding + batched input GEMM (side), fc_embed + Gfc + initial state (third
    // stream: two small latency-bound GEMMs that would otherwise sit behind the batched input GEMM on the side stream)
    // Enqueued longest branch first: the host has just come back from the previous step's loss read, so the GPU starts with an
    // empty queue and every launch ahead of att_embed's GEMM is ~5 us of its idle time.
    const bool split3 = st.gfc_separate() && !st.ss_on();
    UIC_TRY(st.fwd_prologue(s, 2));
    UIC_TRY(st.fwd_prologue(s2, split3 ? 3 : 1));
    if (!split3) UIC_TRY(flush_gathered_late(ss, s2));
    UIC_HIP(hipEventRecord(ss->ev_pro, s2));
    if (split3) {
      UIC_HIP(hipStreamWaitEvent(s3, ss->ev_den, 0));
      if (ss->g2_recorded) UIC_HIP(hipStreamWaitEvent(s3, ss->ev_g2, 0));   // (gathered refresh: fc_embed's operand copy was made on the side stream)
      UIC_TRY(st.fwd_prologue(s3, 4));
      // (the BPTT loop's zeroed carries and ones block -- nothing in the forward pass touches them -- and the live list's cleared
      // d hdrop.  On the main stream behind its own, shorter branch instead: no change per-caption, +0.02 ms with per-image features.)
      UIC_TRY(st.bwd_begin(s3, st.persist_ok(), st.compact));
      bwd_begun = true;
      live_begun = st.compact;
      UIC_TRY(flush_gathered_late(ss, s3));           // (gathered refresh: the recurrence's and the logit layer's operand copies, last to arrive)
      UIC_HIP(hipEventRecord(ss->ev_pro3, s3));
    }
    UIC_TRY(flush_transposes(ss));                    // (behind the branch: only the backward pass reads them)
    UIC_HIP(hipStreamWaitEvent(s, ss->ev_pro, 0));
    if (split3) UIC_HIP(hipStreamWaitEvent(s, ss->ev_pro3, 0));
    if (ss->cast_recorded) UIC_HIP(hipStreamWaitEvent(s, ss->ev_cast, 0));   // the recurrence reads copies made on the side stream
  }

  if (st.compact && !live_begun) UIC_TRY(st.live_begin(s));   // (every logit chunk, on either stream, is ordered behind this point of the main stream)
  UIC_MARK(1, s);
  // persistent mode 3: the whole recurrence as ONE launch (it holds every CU, nothing overlaps it); the logit layer follows
  // chunk by chunk on the side stream beside the BPTT loop
  const bool one_launch = resume || st.persist_ok();
  const bool first_on_main = one_launch && nchunk >= 2 && !st.ss_on();
  if (one_launch) { if (!resume) UIC_TRY(st.fwd_steps(0, t_run, s)); UIC_MARK(2, s); }
  for (int i = 0; i < nchunk; ++i) {
    // after a single launch every step is there at once: the logit layer then takes the chunks LAST FIRST, the order the
    // BPTT loop consumes them in, so that loop starts after one chunk instead of after all of them
    const int c = one_launch ? nchunk - 1 - i : i;
    const int t0 = cb[c], t1 = cb[c + 1];
    if (!one_launch) UIC_TRY(st.fwd_steps(t0, t1, s));
    if (first_on_main && i == 0) {
      // the chunk the BPTT loop starts from, on the MAIN stream itself (idle until that chunk is through anyway): no event hop to
      // the side stream and back in front of the loop; the side stream starts with the next chunk once this one is done
      UIC_TRY(wait_refresh(s));                       // (d hdrop = d logits W_logit reads the transposed copy the side stream made)
      if (!resume) UIC_TRY(st.logits_rows(t0, t1, s));
      UIC_TRY(st.xe_rows(t0, t1, inv, nullptr, 1, s));
      UIC_TRY(st.dh_rows(t0, t1, s, false));
      UIC_HIP(hipEventRecord(ss->ev_main[c], s));
      UIC_HIP(hipStreamWaitEvent(s2, ss->ev_main[c], 0));
      continue;
    }
    if (!one_launch || i == 0) UIC_HIP(hipEventRecord(ss->ev_main[c], s));
    // side: logit layer of the chunk, forward and backward-to-h (beside the next chunk's recurrence)
    if (!one_launch || i == 0) UIC_HIP(hipStreamWaitEvent(s2, ss->ev_main[c], 0));
    if (!resume) UIC_TRY(st.logits_rows(t0, t1, s2));
    UIC_TRY(st.xe_rows(t0, t1, inv, nullptr, 1, s2));
    UIC_TRY(st.dh_rows(t0, t1, s2, true));
    UIC_HIP(hipEventRecord(ss->ev_side[c], s2));
  }
  if (!one_launch) UIC_MARK(2, s);
  UIC_MARK(3, s2);                                    // side: the logit layer (forward, loss, d hdrop) is through
  // side: logit-layer weight gradients + loss reduction, beside the BPTT loop
  // (weight-gradient GEMMs that run beside the BPTT loop keep the 2-stage kernel: the 4-stage ring's 128 KB of LDS would
  // keep the loop's 74-KB workgroups off its CUs -- measured 3.89 -> 3.99 ms)
  g_uic_tn_ring_off = 1;
  g_uic_knobs = d->recurrence & UIC_KNOB_MASK;
  // (data-parallel order: the chunks' gradients share the side stream with the logit layer, where the 128 x 128 kernel's short
  // launches serve the stream's latency better than 56-workgroup ones -- 3.84 vs 3.99 ms per step beside the stand-in exchange)
  if (comm) g_uic_knobs |= UIC_KNOB_CHUNK_TN128;
  UIC_TRY(st.logit_weight_grads(s2, true));
  UIC_TRY(uic_reduce_sum_launch(st.L.row_loss, st.compact ? (size_t)st.live_total() : (size_t)t_run * d->N, 0.f, inv, loss_out, s2));
  if (den_out) UIC_TRY(uic_copy_launch(den_out, st.L.scalars, 4, s2));
  UIC_HIP(hipEventRecord(ss->ev_logit, s2));          // gradient group 0 (logit layer) final: its exchange can start now
  UIC_MARK(10, s2);
  // main: BPTT, each step waits for the d hdrop rows of its chunk; side: the recurrent weight gradients of every
  // finished chunk (transposes + accumulating GEMMs), so only the last chunk's share outlives the loop
  UIC_TRY(wait_refresh(s));                           // the BPTT loop reads the transposed weight copies
  if (!bwd_begun) UIC_TRY(st.bwd_begin(s));
  {
    // The embedding gradient's token bucketing needs only the tokens (labels; under scheduled sampling the tokens the forward
    // pass fed, which the side stream has waited for): in the side stream's slack inside the BPTT window.  Two halves: the
    // share of decode steps [CH, t_run) is gathered while the BPTT loop works on its last chunk, only steps [0, CH) afterwards.
    st.embed_split = st.early_grads() && nchunk >= 2 ? cb[1] : 0;
    if (!st.ss_on() || st.early_grads()) {
      UIC_TRY(uic_embed_bwd_sorted_prepare(st.embed_tokens(), st.embed_ldtok(), d->N, t_run, d->V1, d->E, G->embed_w, st.L.embed_scratch, s2,
                                           st.embed_split));
      st.embed_prepared = true;
      UIC_HIP(hipEventRecord(ss->ev_prep, s2));
    }
  }
  // The tail after the BPTT loop is three independent pieces of throughput work: the last chunk's recurrent weight gradients
  // (side stream), the late group (main stream), and what completes att_lstm.weight_ih and the embedding table (sum over steps,
  // fc' columns, d xt GEMM, gather).  In the default order the third piece goes to the THIRD stream beside the other two
  // instead of behind the first (measured: the side stream's tail was the end of the step, 0.1 ms after the main stream's).
  const bool tail3 = !early && st.embed_prepared;
  for (int c = nchunk - 1; c >= 0; --c) {
    const int t0 = cb[c], t1 = cb[c + 1];
    if (!(first_on_main && c == nchunk - 1)) UIC_HIP(hipStreamWaitEvent(s, ss->ev_side[c], 0));
    if (c == nchunk - 1) UIC_MARK(4, s);              // main: BPTT starts
    UIC_TRY(st.bwd_steps(t0, t1, s));
    UIC_HIP(hipEventRecord(ss->ev_main[c], s));       // (the forward's use of ev_main[c] was consumed long ago)
    if (early) {
      UIC_HIP(hipStreamWaitEvent(s2, ss->ev_main[c], 0));
      UIC_HIP(hipStreamWaitEvent(s3, ss->ev_main[c], 0));
      UIC_TRY(st.wgrad_chunk(t0, t1, c == nchunk - 1, s2, s3));
      if (c == 1) UIC_TRY(st.embed_grad(1, s2));      // d xt of steps [CH, t_run) is complete: their share of the embedding table
    } else if (comm) {
      UIC_HIP(hipStreamWaitEvent(s2, ss->ev_main[c], 0));
      UIC_TRY(st.wgrad_chunk(t0, t1, c == nchunk - 1, s2));
    } else {
      // default order: the chunk's two independent shares on streams 3 and 4 (the 256 x 256 weight-gradient kernel runs a share on
      // ~60 CUs: side by side, and beside the side stream's logit layer, they leave the BPTT chain most of the chip), the side
      // stream stays with the logit layer
      // ... one chunk after the other on stream 3 while the loop runs (two streams of them slow the chain down again: 3.15 vs
      // 3.08 ms, profiles/r05_v1_ab_knobs.txt); the LAST chunk's shares, which start when the loop is over, side by side
      const bool two = false;      // (a fourth stream is not worth having: see get_side)
      if (two && !(g_uic_knobs & UIC_KNOB_TWO_WG_STREAMS) && nchunk > 1) {
        // (the att_lstm / h2att share moves to stream 4 for this chunk: it accumulates into what stream 3's chunks wrote)
        UIC_HIP(hipEventRecord(ss->ev_s3, s3));
        UIC_HIP(hipStreamWaitEvent(s4, ss->ev_s3, 0));
      }
      UIC_HIP(hipStreamWaitEvent(s3, ss->ev_main[c], 0));
      if (two) UIC_HIP(hipStreamWaitEvent(s4, ss->ev_main[c], 0));
      if (c == 0 && !(g_uic_knobs & UIC_KNOB_LAST_BESIDE)) g_uic_tn_ring_off = 0;
      UIC_TRY(st.wgrad_chunk(t0, t1, c == nchunk - 1, s3, two ? s4 : s3, true));
    }
  }
  if (early) {
    // side: what completes att_lstm.weight_ih and the embedding table comes first, so that the two big tensors of the early
    // group are final with gradient group 1 (the fc_embed chain and the bias sums -- ~4 MB -- follow)
    UIC_TRY(st.fc_cols_grad(s2, st.L.slab2, st.L.tSA, st.L.tSB));
    UIC_TRY(st.embed_grad(0, s2));
    UIC_HIP(hipEventRecord(ss->ev_embed, s2));
    ss->embed_recorded = true;
    UIC_HIP(hipEventRecord(ss->ev_s3, s3));
    UIC_HIP(hipStreamWaitEvent(s2, ss->ev_s3, 0));    // the third stream's share of the weight gradients
  } else if (comm) {
    if (tail3) {   // (round 4's tail: what completes att_lstm.weight_ih and the embedding table on stream 3 beside the last chunk's gradients)
      UIC_HIP(hipStreamWaitEvent(s3, ss->ev_main[0], 0));
      UIC_HIP(hipStreamWaitEvent(s3, ss->ev_prep, 0));
      UIC_TRY(st.fc_cols_grad(s3, st.L.slab3, st.L.tTA, st.L.tTB));
      UIC_TRY(st.embed_grad(0, s3));
      UIC_HIP(hipEventRecord(ss->ev_s3, s3));
      UIC_HIP(hipEventRecord(ss->ev_embed, s3));
      ss->embed_recorded = true;
    }
  } else {
    // the recurrent weight gradients are complete when both shares of the last chunk are (joined on stream 3)
    UIC_HIP(hipEventRecord(ss->ev_s4, s4));
    UIC_HIP(hipStreamWaitEvent(s3, ss->ev_s4, 0));
    if (tail3) {
      UIC_HIP(hipStreamWaitEvent(s2, ss->ev_main[0], 0));   // the BPTT loop is through: dG1 of every step exists
      UIC_TRY(st.fc_cols_grad(s2, st.L.slab2, st.L.tSA, st.L.tSB));
      UIC_TRY(st.embed_grad(0, s2));
      UIC_HIP(hipEventRecord(ss->ev_embed, s2));
      ss->embed_recorded = true;
    }
  }
  // side: the rest of the early gradient group (LSTM / h2att biases, embedding, fc_embed); main: the late group
  // (attention accumulation, ctx2att, att_embed).  ev_early: the early group, the logit layer and the loss are final.
  g_uic_tn_ring_off = 0;
  UIC_MARK(5, s);                                     // main: BPTT done
  hipStream_t s_lstm = (early || comm) ? s2 : s3;
  UIC_HIP(hipEventRecord(ss->ev_lstm, s_lstm));       // gradient group 1 final (uic_topdown_grad_ready_wait)
  UIC_MARK(6, s_lstm);                                // side: recurrent weight gradients done
  UIC_TRY(st.bwd_epilogue_late(s, true));             // (enqueued first: it is the longer of the two tails)
  UIC_MARK(7, s);
  if (!early && !tail3) UIC_HIP(hipStreamWaitEvent(s2, ss->ev_main[0], 0));
  if (comm && tail3) UIC_HIP(hipStreamWaitEvent(s2, ss->ev_s3, 0));      // (d fc' below reads the dGfc that fc_cols_grad left)
  const bool chunks_elsewhere = !early && !comm;      // the chunks' weight (and bias) gradients were made on stream 3
  UIC_TRY(st.bwd_epilogue_early(s2, true, true, early || tail3, !chunks_elsewhere));
  if (chunks_elsewhere) {
    UIC_HIP(hipStreamWaitEvent(s2, ss->ev_lstm, 0));  // the early group includes them; bias_hh = bias_ih needs the bias columns
    UIC_TRY(st.bias_hh_copies(s2));
  }
  UIC_HIP(hipEventRecord(ss->ev_early, s2));
  ss->early_recorded = true;
  UIC_MARK(8, s2);
  UIC_HIP(hipStreamWaitEvent(s, ss->ev_early, 0));    // join
  UIC_MARK(9, s);
  ss->marks_valid = ss->marks_on;
#undef UIC_MARK
#undef UIC_HIP
  return UIC_OK;
}

int uic_topdown_step_marks(int32_t enable, float* ms_out) {
  SideStream* ss = nullptr;
  UIC_TRY(get_side(&ss));
  if (ms_out) {
    UIC_REQUIRE(ss->marks_valid, "step_marks: no fused step has run with the marks enabled");
    UIC_TRY(uic_check_hip(hipEventSynchronize(ss->mark[9]), "hipEventSynchronize"));   // (joined: every other mark precedes it)
    UIC_TRY(uic_check_hip(hipEventSynchronize(ss->mark[8]), "hipEventSynchronize"));
    ms_out[0] = 0.f;
    for (int i = 1; i < UIC_STEP_MARKS; ++i) UIC_TRY(uic_check_hip(hipEventElapsedTime(&ms_out[i], ss->mark[0], ss->mark[i]), "hipEventElapsedTime"));
  }
  ss->marks_on = enable != 0;
  if (!ss->marks_on) ss->marks_valid = false;
  return UIC_OK;
}

int uic_topdown_grad_ready_wait(void* stream, int32_t group) {
  SideStream* ss = nullptr;
  UIC_TRY(get_side(&ss));
  UIC_REQUIRE(group >= 0 && group <= 4, "grad_ready_wait: group=%d must be 0 (logit layer), 1 (LSTM weights), 2 (early group), 3 (embedding table + att_lstm.weight_ih) or 4 (embedding table)", group);
  UIC_REQUIRE(ss->early_recorded, "grad_ready_wait: no uic_topdown_xe_train_step has run on this device yet");
  if (group == 4)   // embed.0.weight alone: its gather runs beside the last chunk's weight gradients and is through first
    return uic_check_hip(hipStreamWaitEvent((hipStream_t)stream, ss->embed_recorded ? ss->ev_embed : ss->ev_early, 0), "hipStreamWaitEvent");
  if (group == 3) {
    // embed.0.weight and core.att_lstm.weight_ih: the table's gather and the fc' columns (ev_embed) + the matrix's other columns,
    // which come with the recurrent weight gradients (ev_lstm).  A step that made them inside its early epilogue (scheduled
    // sampling without the early order) has no separate event: they are final with the early group
    if (!ss->embed_recorded) return uic_check_hip(hipStreamWaitEvent((hipStream_t)stream, ss->ev_early, 0), "hipStreamWaitEvent");
    UIC_TRY(uic_check_hip(hipStreamWaitEvent((hipStream_t)stream, ss->ev_embed, 0), "hipStreamWaitEvent"));
    return uic_check_hip(hipStreamWaitEvent((hipStream_t)stream, ss->ev_lstm, 0), "hipStreamWaitEvent");
  }
  hipEvent_t ev = group == 0 ? ss->ev_logit : group == 1 ? ss->ev_lstm : ss->ev_early;
  return uic_check_hip(hipStreamWaitEvent((hipStream_t)stream, ev, 0), "hipStreamWaitEvent");
}

int uic_topdown_sample(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                       const uic_topdown_batch* b, int32_t Lsteps, int32_t sample_max, float temperature,
                       int32_t decoding_constraint, uint32_t seed, const int64_t* forced, int32_t training,
                       void* workspace, int64_t* seq, float* seq_logp, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived && b && workspace && seq && seq_logp, "sample: null pointer");
  UIC_REQUIRE(b->fc_feats && b->att_feats, "sample: batch needs fc_feats and att_feats");
  UIC_REQUIRE(Lsteps >= 1 && Lsteps <= d->T, "sample: L=%d outside [1,%d]", Lsteps, d->T);
  UIC_REQUIRE(temperature > 0.f, "sample: temperature must be positive");
  hipStream_t s = (hipStream_t)stream;
  const Layout L = make_layout(*d, workspace);
  const Derived dv = make_derived(*d, w, (void*)derived);
  const int dt = d->dtype;
  const int N = d->N, H = d->H, V1 = d->V1;
  const int V1p = (int)vpad(V1);
  const size_t S = uic_dtype_size(dt);
  const size_t NH = (size_t)N * H;
  const void *fc_in, *att_in;
  const float drop_p = (training & 1) ? d->drop_p : 0.f;
  UIC_TRY(prepare_features(*d, w, dv, b, L, training, drop_p, seed, &fc_in, &att_in, s));
  UIC_TRY(wait_refresh(s));
  {
    Step st;
    st.init(d, w, derived, b, Lsteps, training, seed, workspace, nullptr);
    if (st.decode_persist_ok(sample_max, temperature, decoding_constraint)) {
      UIC_TRY(uic_zero4_launch(L.h_att, NH * S, L.h_lang, NH * S, L.c_att, NH * 4, L.c_lang, NH * 4, s));
      return st.decode_persist(Lsteps, sample_max, forced, false, seq, seq_logp, s);
    }
  }
  UIC_TRY(uic_fill_launch(L.s_h_att[0], 0, NH * S, s));
  UIC_TRY(uic_fill_launch(L.s_h_lang[0], 0, NH * S, s));
  UIC_TRY(uic_fill_launch(L.s_c_att[0], 0, NH * 4, s));
  UIC_TRY(uic_fill_launch(L.s_c_lang[0], 0, NH * 4, s));
  UIC_TRY(uic_fill_launch(L.s_it, 0, (size_t)N * 8, s));       // <bos> = 0 (AttModel.py:214-215)
  UIC_TRY(uic_fill_launch(L.s_unf, 0, (size_t)N * 4, s));
  UIC_TRY(uic_fill_launch(L.s_nunf, 0, (size_t)(d->T + 2) * UIC_NUNF_STRIPES * 4, s));
  for (int t = 0; t < Lsteps; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    UIC_TRY(decode_step(*d, w, dv, b, L, cur, nxt, t, drop_p, seed, s, (training & 1) != 0, t > 0));
    UicSampleParams p;
    memset(&p, 0, sizeof(p));
    p.dtype = dt; p.N = N; p.V1 = V1; p.ldv = V1p; p.t = t; p.L = Lsteps;
    p.logits = L.s_logits; p.sample_max = sample_max; p.temperature = temperature; p.seed = seed;
    p.decoding_constraint = decoding_constraint;
    p.seq = seq; p.seq_logp = seq_logp; p.it = L.s_it; p.unfinished = L.s_unf; p.n_unfinished = L.s_nunf;
    p.forced = forced;
    if (t + 1 < Lsteps) {      // the next step's embedding rows ride in the sampling kernel (one launch less per step)
      p.embed_table = dv.embed_w; p.embed_table_dtype = dt; p.embed_V1 = V1; p.embed_E = d->E; p.embed_drop_p = drop_p; p.embed_site = UIC_SITE_EMBED;
      p.embed_idx_base = (size_t)(t + 1) * N * d->E; p.xt_out = L.s_xt;
    }
    UIC_TRY(uic_sample_step_launch(p, s));
  }
  return UIC_OK;
}


// The multinomial / greedy pass in the TRAINING layout: the per-step chain of uic_topdown_forward with each step's input
// tokens drawn from the previous step's distribution, so that every activation the backward pass reads (gates, states,
// attention weights, contexts, dropped outputs, logits) is left in the workspace -- uic_topdown_xe_train_step with
// training bit 2 (value 4) then starts at the criterion instead of replaying the sampled captions teacher-forced.
int uic_topdown_sample_train(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                             const uic_topdown_batch* b, int32_t Lsteps, int32_t sample_max, float temperature,
                             int32_t decoding_constraint, uint32_t seed, const int64_t* forced, int32_t training,
                             void* workspace, int64_t* seq, float* seq_logp, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived && b && workspace && seq && seq_logp, "sample_train: null pointer");
  UIC_REQUIRE(b->fc_feats && b->att_feats, "sample_train: batch needs fc_feats and att_feats");
  UIC_REQUIRE(Lsteps >= 1 && Lsteps <= d->T, "sample_train: L=%d outside [1,%d]", Lsteps, d->T);
  UIC_REQUIRE(temperature > 0.f, "sample_train: temperature must be positive");
  UIC_REQUIRE(!(b->ss_prob > 0.f), "sample_train: the pass draws every token itself (ss_prob must be 0)");
  hipStream_t s = (hipStream_t)stream;
  Step st;
  st.init(d, w, derived, b, Lsteps, training, seed, workspace, nullptr);
  const Layout& L = st.L;
  const int N = d->N;
  const void *f, *a;
  UIC_TRY(prepare_features(st.d, w, st.dv, b, L, training, st.drop_p, seed, &f, &a, s));
  UIC_TRY(uic_zero4_launch(L.h_att, st.NH * st.S, L.h_lang, st.NH * st.S, L.c_att, st.NH * 4, L.c_lang, st.NH * 4, s));
  UIC_TRY(wait_refresh(s));
  if (st.decode_persist_ok(sample_max, temperature, decoding_constraint))
    return st.decode_persist(Lsteps, sample_max, forced, true, seq, seq_logp, s);
  UIC_TRY(uic_fill_launch(L.s_it, 0, (size_t)N * 8, s));       // <bos> = 0 (AttModel.py:214-215)
  UIC_TRY(uic_fill_launch(L.s_unf, 0, (size_t)N * 4, s));
  UIC_TRY(uic_fill_launch(L.s_nunf, 0, (size_t)(d->T + 2) * UIC_NUNF_STRIPES * 4, s));
  for (int t = 0; t < Lsteps; ++t) {
    if (t == 0)   // (later steps: written by the previous step's sampling kernel)
      UIC_TRY(uic_embed_fwd_t_launch(st.dt, st.dv.embed_w, st.dt, st.V1, st.E, L.s_it, 1, N, 1, st.drop_p, seed, UIC_SITE_EMBED,
                                   (size_t)t * N * st.E, 1, offw(L.xt_all, (size_t)t * N * st.E, st.dt), s));
    UIC_TRY(st.fwd_step(t, s, true));
    UIC_TRY(st.logits_rows_now(t, t + 1, s));
    UicSampleParams p;
    memset(&p, 0, sizeof(p));
    p.dtype = st.dt; p.N = N; p.V1 = st.V1; p.ldv = st.V1p; p.t = t; p.L = Lsteps;
    p.logits = L.logits + (size_t)t * N * st.V1p; p.sample_max = sample_max; p.temperature = temperature; p.seed = seed;
    p.decoding_constraint = decoding_constraint;
    p.seq = seq; p.seq_logp = seq_logp; p.it = L.s_it; p.unfinished = L.s_unf; p.n_unfinished = L.s_nunf;
    p.forced = forced;
    if (t + 1 < Lsteps) {
      p.embed_table = st.dv.embed_w; p.embed_table_dtype = st.dt; p.embed_V1 = st.V1; p.embed_E = st.E; p.embed_drop_p = st.drop_p; p.embed_site = UIC_SITE_EMBED;
      p.embed_idx_base = (size_t)(t + 1) * N * st.E; p.xt_out = offw(L.xt_all, (size_t)(t + 1) * N * st.E, st.dt);
    }
    UIC_TRY(uic_sample_step_launch(p, s));
  }
  return UIC_OK;
}

int uic_topdown_prepare_feature(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                                const uic_topdown_batch* b, int32_t training, uint32_t seed, void* workspace,
                                float* fc_out, float* att_out, float* p_att_out, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived && b && workspace && fc_out && att_out && p_att_out, "prepare_feature: null pointer");
  UIC_REQUIRE(b->fc_feats && b->att_feats, "prepare_feature: batch needs fc_feats and att_feats");
  hipStream_t s = (hipStream_t)stream;
  const Layout L = make_layout(*d, workspace);
  const Derived dv = make_derived(*d, w, (void*)derived);
  const void *fc_in, *att_in;
  const float drop_p = (training & 1) ? d->drop_p : 0.f;
  UIC_TRY(prepare_features(*d, w, dv, b, L, training, drop_p, seed, &fc_in, &att_in, s));
  const size_t N = d->N, R = d->R, H = d->H, A = d->A;
  UIC_TRY(uic_to_f32_launch(d->dtype, L.fcp, fc_out, N * H, s));
  UIC_TRY(uic_to_f32_launch(d->dtype, L.attp, att_out, N * R * H, s));
  return uic_to_f32_launch(d->dtype, L.patt, p_att_out, N * R * A, s);
}

int uic_topdown_logprobs_state(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived, const int64_t* it,
                               const float* fc, const float* att, const float* p_att, const float* att_masks,
                               const float* h_in, const float* c_in, int32_t t, int32_t training, uint32_t seed, void* workspace,
                               float* logprobs, float* h_out, float* c_out, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived && it && fc && att && p_att && h_in && c_in && workspace && logprobs && h_out && c_out, "logprobs_state: null pointer");
  UIC_REQUIRE(d->seq_per_img <= 1, "logprobs_state: the prepared features are per caption row (seq_per_img must be 0 or 1)");
  UIC_REQUIRE(t >= 0, "logprobs_state: t=%d", t);
  hipStream_t s = (hipStream_t)stream;
  const Layout L = make_layout(*d, workspace);
  const Derived dv = make_derived(*d, w, (void*)derived);
  const int dt = d->dtype;
  const size_t N = d->N, R = d->R, H = d->H, A = d->A, V1 = d->V1;
  const size_t S = uic_dtype_size(dt);
  UIC_TRY(wait_refresh(s));
  // the caller's prepared features and state become the step's operands (operand dtype) ...
  UIC_TRY(uic_cast_f32_launch(dt, fc, L.fcp, N * H, s));
  UIC_TRY(uic_cast_f32_launch(dt, att, L.attp, N * R * H, s));
  UIC_TRY(uic_cast_f32_launch(dt, p_att, L.patt, N * R * A, s));
  UIC_TRY(uic_cast_f32_launch(dt, h_in, L.s_h_att[0], N * H, s));
  UIC_TRY(uic_cast_f32_launch(dt, h_in + N * H, L.s_h_lang[0], N * H, s));
  UIC_TRY(uic_copy_launch(L.s_c_att[0], c_in, N * H * 4, s));
  UIC_TRY(uic_copy_launch(L.s_c_lang[0], c_in + N * H, N * H * 4, s));
  UIC_TRY(uic_copy_launch(L.s_it, it, N * 8, s));
  uic_topdown_batch b;
  memset(&b, 0, sizeof(b));
  b.att_masks = att_masks;
  const float drop_p = (training & 1) ? d->drop_p : 0.f;
  // ... one get_logprobs_state (P/models/AttModel.py:158-165) ...
  UIC_TRY(decode_step(*d, w, dv, &b, L, 0, 1, t, drop_p, seed, s, (training & 1) != 0));
  // ... and its results go back out: log_softmax of the logits, the new (h, c) stacks [att_lstm, lang_lstm]
  UicXeParams x;
  memset(&x, 0, sizeof(x));
  x.dtype = dt; x.M = (int)N; x.V1 = (int)V1; x.ldv = (int)vpad(V1); x.N = (int)N;
  x.logits = L.s_logits; x.logprobs = logprobs; x.lp_step_stride = V1; x.lp_row_stride = V1;
  UIC_TRY(uic_xe_launch(x, s));
  UIC_TRY(uic_to_f32_launch(dt, L.s_h_att[1], h_out, N * H, s));
  UIC_TRY(uic_to_f32_launch(dt, L.s_h_lang[1], h_out + N * H, N * H, s));
  UIC_TRY(uic_copy_launch(c_out, L.s_c_att[1], N * H * 4, s));
  UIC_TRY(uic_copy_launch(c_out + N * H, L.s_c_lang[1], N * H * 4, s));
  (void)S;
  return UIC_OK;
}

int uic_topdown_beam_done_lists(const uic_topdown_dims* d, void* workspace, int32_t Lsteps, int32_t beam_size, int32_t* done_count,
                                float* done_p, int64_t* done_seq, float* done_lp, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(workspace && done_count && done_p && done_seq && done_lp, "beam_done_lists: null pointer");
  UIC_REQUIRE(beam_size >= 1 && d->N % beam_size == 0 && Lsteps >= 1 && Lsteps <= d->T, "beam_done_lists: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const Layout L = make_layout(*d, workspace);
  const size_t n_img = (size_t)d->N / beam_size, LB = (size_t)Lsteps * beam_size;
  UIC_TRY(uic_copy_launch(done_count, L.bm_done_count, n_img * 4, s));
  UIC_TRY(uic_copy_launch(done_p, L.bm_done_p, n_img * LB * 4, s));
  UIC_TRY(uic_copy_launch(done_seq, L.bm_done_seq, n_img * LB * Lsteps * 8, s));
  return uic_copy_launch(done_lp, L.bm_done_lp, n_img * LB * Lsteps * 4, s);
}

int uic_topdown_sample_beam(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                            const uic_topdown_batch* b, int32_t Lsteps, int32_t beam_size, int32_t decoding_constraint,
                            int32_t max_ppl, void* workspace, int64_t* seq, float* seq_logp, void* stream) {
  UIC_TRY(check_dims(d));
  UIC_REQUIRE(w && derived && b && workspace && seq && seq_logp, "sample_beam: null pointer");
  UIC_REQUIRE(b->fc_feats && b->att_feats, "sample_beam: batch needs fc_feats and att_feats");
  UIC_REQUIRE(Lsteps >= 1 && Lsteps <= d->T, "sample_beam: L=%d outside [1,%d]", Lsteps, d->T);
  UIC_REQUIRE(beam_size >= 1 && beam_size <= UIC_BEAM_MAX && beam_size <= d->V1, "sample_beam: beam_size=%d outside [1, %d]", beam_size, UIC_BEAM_MAX);
  UIC_REQUIRE(d->N % beam_size == 0, "sample_beam: N=%d rows must be images x beam_size=%d (every image replicated beam_size times)", d->N, beam_size);
  hipStream_t s = (hipStream_t)stream;
  const Layout L = make_layout(*d, workspace);
  const Derived dv = make_derived(*d, w, (void*)derived);
  const int dt = d->dtype;
  const int N = d->N, H = d->H, V1 = d->V1, T = d->T;
  const size_t S = uic_dtype_size(dt), NH = (size_t)N * H;
  const void *fc_in, *att_in;
  UIC_TRY(prepare_features(*d, w, dv, b, L, 0, 0.f, 0, &fc_in, &att_in, s));      // eval mode, like eval_utils.eval_split
  UIC_TRY(wait_refresh(s));
  UIC_TRY(uic_fill_launch(L.s_h_att[0], 0, NH * S, s));
  UIC_TRY(uic_fill_launch(L.s_h_lang[0], 0, NH * S, s));
  UIC_TRY(uic_fill_launch(L.s_c_att[0], 0, NH * 4, s));
  UIC_TRY(uic_fill_launch(L.s_c_lang[0], 0, NH * 4, s));
  UIC_TRY(uic_fill_launch(L.s_it, 0, (size_t)N * 8, s));       // <bos> for every beam row (AttModel.py:187)
  for (int i = 0; i < 2; ++i) {
    UIC_TRY(uic_fill_launch(L.bm_seq[i], 0, (size_t)N * T * 8, s));
    UIC_TRY(uic_fill_launch(L.bm_lp[i], 0, (size_t)N * T * 4, s));
  }
  UIC_TRY(uic_fill_launch(L.bm_sum, 0, (size_t)N * 4, s));
  UIC_TRY(uic_fill_launch(L.bm_done_count, 0, (size_t)N * 4, s));
  UicBeamParams p;
  memset(&p, 0, sizeof(p));
  p.n_img = N / beam_size; p.B = beam_size; p.L = Lsteps; p.V1 = V1; p.ldv = (int)vpad(V1);
  p.decoding_constraint = decoding_constraint; p.max_ppl = max_ppl;
  p.logits = L.s_logits; p.cand_val = L.bm_cand_val; p.cand_idx = L.bm_cand_idx;
  p.beam_seq_hist[0] = L.bm_seq[0]; p.beam_seq_hist[1] = L.bm_seq[1]; p.beam_lp_hist[0] = L.bm_lp[0]; p.beam_lp_hist[1] = L.bm_lp[1];
  p.beam_sum = L.bm_sum; p.parent = L.bm_parent; p.it = L.s_it;
  p.done_count = L.bm_done_count; p.done_p = L.bm_done_p; p.done_seq = L.bm_done_seq; p.done_lp = L.bm_done_lp;
  UIC_TRY(decode_step(*d, w, dv, b, L, 0, 1, 0, 0.f, 0, s));   // logprobs after <bos> (AttModel.py:184-190)
  for (int t = 0; t < Lsteps; ++t) {
    p.t = t;
    UIC_TRY(uic_beam_step_launch(p, s));
    if (t + 1 == Lsteps) break;                                // (the reference's last get_logprobs_state is never used)
    UIC_TRY(uic_beam_gather_launch(dt, L.bm_parent, N, beam_size, H, L.s_h_att[1], L.s_h_att[0], L.s_h_lang[1], L.s_h_lang[0],
                                   L.s_c_att[1], L.s_c_att[0], L.s_c_lang[1], L.s_c_lang[0], s));
    UIC_TRY(decode_step(*d, w, dv, b, L, 0, 1, t + 1, 0.f, 0, s));
  }
  return uic_beam_final_launch(p, seq, seq_logp, s);
}

}  // extern "C"
