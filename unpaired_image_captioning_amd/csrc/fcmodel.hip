// Host-side sequencing of the FC captioner (the `fc` caption model = FCModel_NMT + maxout LSTMCore,
// P/models/FCModel_NMT.py:21-217; BASELINE config 1) on one MI355X.
//
// Same restructuring as the TopDown path: teacher forcing makes every core input known up front
// (step 0 = img_embed(fc), step i = embed[labels[:, i-1]]), so i2h runs once over all S*N rows and only the
// h2h GEMM (K = H) with the fused maxout-cell epilogue stays in the recurrence; the logit GEMM, log-softmax and
// criterion run once over all rows; weight gradients are single GEMMs over the stacked rows.
#include "uic_common.h"
#include "uic_host.h"
#include "../../include/uic_hip.h"
#include <string.h>

namespace {

struct FcLayout {
  // operand-dtype weight copies and transposes (rebuilt by every forward: the model is small)
  const void* img_w; const void* i2h_w; const void* h2h_w; const void* logit_w;
  void* c_img_w; void* c_i2h_w; void* c_h2h_w; void* c_logit_w;
  void* h2hT;      // [H, 5H]
  void* i2hT;      // [E, 5H]
  void* logit_wT;  // [H, V1p]
  void* fcT;
  void* x_all;     // [S, N, E]
  float* gx;       // [S*N, 5H]
  void* h_buf;     // [(S+1), N, H] dropped next_h (slot 0 = zeros)
  float* c_buf;    // [(S+1), N, H]
  void* gates;     // [S*N, 5H]
  float* logits; void* dlogits; float* row_loss; float* scalars;
  float* dh_all;   // [(S-1)*N, H]
  void* ds_all;    // [S*N, 5H]
  float* dhrec; float* dc; float* dx_all; void* dx0;
  void* tA; void* tB; float* colscratch; size_t colscratch_floats; float* slab; size_t slab_bytes;
  // sampling
  void* s_h[2]; float* s_c[2]; void* s_xt; float* s_logits; int64_t* s_it; int* s_unf; int* s_nunf;
  // beam search bookkeeping (rows = (image, beam))
  float* bm_cand_val; int* bm_cand_idx; int64_t* bm_seq[2]; float* bm_lp[2]; float* bm_sum; int* bm_parent;
  int* bm_done_count; float* bm_done_p; int64_t* bm_done_seq; float* bm_done_lp;
  int* embed_scratch;   // uic_embed_bwd_sorted_launch
  size_t total;
};

FcLayout fc_layout(const uic_fc_dims& d, const uic_fc_weights* w, void* ws) {
  FcLayout L;
  memset(&L, 0, sizeof(L));
  Bump b{(char*)ws, 0};
  const size_t Sz = uic_dtype_size(d.dtype);
  const size_t N = d.N, Dfc = d.Dfc, H = d.H, E = d.E, V1 = d.V1, S = d.S, V1p = vpad(V1);
  const size_t M = S * N, Mp = rup8(M), Np = rup8(N);
  const bool bf = d.dtype == UIC_BF16;
  L.c_img_w = b.take(E * Dfc * Sz);
  L.c_i2h_w = b.take(5 * H * E * Sz);
  L.c_h2h_w = b.take(5 * H * H * Sz);
  L.c_logit_w = b.take(V1 * H * Sz);
  L.img_w = bf || !w ? L.c_img_w : (const void*)w->img_embed_w;
  L.i2h_w = bf || !w ? L.c_i2h_w : (const void*)w->i2h_w;
  L.h2h_w = bf || !w ? L.c_h2h_w : (const void*)w->h2h_w;
  L.logit_w = bf || !w ? L.c_logit_w : (const void*)w->logit_w;
  L.h2hT = b.take(H * 5 * H * Sz);
  L.i2hT = b.take(E * 5 * H * Sz);
  L.logit_wT = b.take(H * V1p * Sz);
  L.fcT = b.take(N * Dfc * Sz);
  L.x_all = b.take(M * E * Sz);
  L.gx = (float*)b.take(M * 5 * H * 4);
  L.h_buf = b.take((S + 1) * N * H * Sz);
  L.c_buf = (float*)b.take((S + 1) * N * H * 4);
  L.gates = b.take(M * 5 * H * Sz);
  L.logits = (float*)b.take(M * V1p * 4);
  L.dlogits = b.take(M * V1p * Sz);
  L.row_loss = (float*)b.take(M * 4);
  L.scalars = (float*)b.take(64);
  L.dh_all = (float*)b.take(M * H * 4);
  L.ds_all = b.take(M * 5 * H * Sz);
  L.dhrec = (float*)b.take(N * H * 4);
  L.dc = (float*)b.take(N * H * 4);
  L.dx_all = (float*)b.take(M * E * 4);
  L.dx0 = b.take(N * E * Sz);
  size_t ta = V1 * Mp;
  if (5 * H * Mp > ta) ta = 5 * H * Mp;
  if (E * Np > ta) ta = E * Np;
  size_t tb = (H + E) * Mp;
  if (Dfc * Np > tb) tb = Dfc * Np;
  L.tA = b.take(ta * Sz);
  L.tB = b.take(tb * Sz);
  size_t maxcols = V1p > 5 * H ? V1p : 5 * H;
  L.colscratch_floats = 128 * maxcols;
  L.colscratch = (float*)b.take(L.colscratch_floats * 4);
  L.slab_bytes = 4 * (5 * H) * (H + E) * 4;
  L.slab = (float*)b.take(L.slab_bytes);
  for (int i = 0; i < 2; ++i) {
    L.s_h[i] = b.take(N * H * Sz);
    L.s_c[i] = (float*)b.take(N * H * 4);
  }
  L.s_xt = b.take(N * E * Sz);
  L.s_logits = (float*)b.take(N * V1p * 4);
  L.s_it = (int64_t*)b.take(N * 8);
  L.s_unf = (int*)b.take(N * 4);
  L.s_nunf = (int*)b.take((S + 2) * UIC_NUNF_STRIPES * 4);
  L.bm_cand_val = (float*)b.take(N * UIC_BEAM_MAX * 4);
  L.bm_cand_idx = (int*)b.take(N * UIC_BEAM_MAX * 4);
  for (int i = 0; i < 2; ++i) {
    L.bm_seq[i] = (int64_t*)b.take(N * S * 8);
    L.bm_lp[i] = (float*)b.take(N * S * 4);
  }
  L.bm_sum = (float*)b.take(N * 4);
  L.bm_parent = (int*)b.take(N * 4);
  L.bm_done_count = (int*)b.take(N * 4);
  L.bm_done_p = (float*)b.take(N * S * 4);
  L.bm_done_seq = (int64_t*)b.take(N * S * S * 8);
  L.bm_done_lp = (float*)b.take(N * S * S * 4);
  L.embed_scratch = (int*)b.take(uic_embed_bwd_sorted_scratch_ints((int)N, (int)S, d.V1, d.E) * 4);
  L.total = (b.off + 255) & ~(size_t)255;
  return L;
}

int fc_check(const uic_fc_dims* d) {
  UIC_REQUIRE(d != nullptr, "null dims");
  UIC_REQUIRE(d->dtype == UIC_F32 || d->dtype == UIC_BF16, "bad dtype %d", d->dtype);
  UIC_REQUIRE(d->N > 0 && d->S >= 2 && d->V1 > 1, "bad sizes N=%d S=%d V1=%d", d->N, d->S, d->V1);
  UIC_REQUIRE(d->Dfc % 8 == 0 && d->H % 8 == 0 && d->E % 8 == 0, "Dfc=%d H=%d E=%d must be multiples of 8", d->Dfc, d->H, d->E);
  UIC_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "drop_p=%f outside [0,1)", (double)d->drop_p);
  return UIC_OK;
}

int fc_refresh(const uic_fc_dims& d, const uic_fc_weights* w, const FcLayout& L, hipStream_t s) {
  const int dt = d.dtype, H = d.H, E = d.E, V1 = d.V1, V1p = (int)vpad(V1), H5 = 5 * H;
  if (dt == UIC_BF16) {
    UIC_TRY(uic_cast_f32_launch(dt, w->img_embed_w, L.c_img_w, (size_t)E * d.Dfc, s));
    UIC_TRY(uic_cast_f32_launch(dt, w->i2h_w, L.c_i2h_w, (size_t)H5 * E, s));
    UIC_TRY(uic_cast_f32_launch(dt, w->h2h_w, L.c_h2h_w, (size_t)H5 * H, s));
    UIC_TRY(uic_cast_f32_launch(dt, w->logit_w, L.c_logit_w, (size_t)V1 * H, s));
  }
  UIC_TRY(uic_transpose_launch(dt, L.h2h_w, H5, H, H, L.h2hT, H5, s));
  UIC_TRY(uic_transpose_launch(dt, L.i2h_w, H5, E, E, L.i2hT, H5, s));
  return uic_transpose_launch(dt, L.logit_w, V1, H, H, L.logit_wT, V1p, s);
}

// one core step: gates = pre + [x W_i2h^T] + h W_h2h^T; maxout cell; h_out = dropout(next_h)
int fc_core(const uic_fc_dims& d, const FcLayout& L, const void* x, const uic_fc_weights* w, const float* pre,
            const void* h_prev, const float* c_prev, float* c_out, void* h_out, void* gates_out, float drop_p,
            unsigned seed, int step, hipStream_t s) {
  UicGemmParams g = gemm_base(d.dtype, d.N, 5 * d.H);
  g.lstm = 2; g.H = d.H;
  if (x) {
    add_seg(g, x, d.E, L.i2h_w, d.E, d.E);
    g.bias = w->i2h_b; g.bias2 = w->h2h_b;
  }
  add_seg(g, h_prev, d.H, L.h2h_w, d.H, d.H);
  g.pre1 = pre; g.ldpre1 = 5 * d.H;
  g.c_prev = c_prev; g.c_out = c_out; g.h_out = h_out; g.ldh = d.H; g.gates_out = gates_out;
  g.drop_p = drop_p; g.seed = seed; g.site = UIC_SITE_OUT0 + (unsigned)step;
  return uic_gemm_launch(g, s);
}

}  // namespace

extern "C" {

size_t uic_fc_workspace_bytes(const uic_fc_dims* d) {
  if (fc_check(d)) return 0;
  return fc_layout(*d, nullptr, nullptr).total;
}

int uic_fc_forward(const uic_fc_dims* d, const uic_fc_weights* w, const uic_topdown_batch* b, int32_t s_run,
                   int32_t training, uint32_t seed, void* workspace, float* logprobs_out, void* stream) {
  UIC_TRY(fc_check(d));
  UIC_REQUIRE(w && b && workspace && b->fc_feats && b->labels, "fc_forward: null pointer");
  UIC_REQUIRE(s_run >= 2 && s_run <= d->S, "fc_forward: s_run=%d outside [2,%d]", s_run, d->S);
  UIC_REQUIRE(b->ld_labels >= s_run - 1, "fc_forward: labels have %d columns, need %d", b->ld_labels, s_run - 1);
  hipStream_t s = (hipStream_t)stream;
  const FcLayout L = fc_layout(*d, w, workspace);
  const int dt = d->dtype, N = d->N, H = d->H, E = d->E, V1 = d->V1, V1p = (int)vpad(V1), H5 = 5 * H;
  const size_t Sz = uic_dtype_size(dt), NH = (size_t)N * H;
  const float drop_p = training ? d->drop_p : 0.f;
  UIC_TRY(fc_refresh(*d, w, L, s));
  const void* fc_in = b->fc_feats;
  if (dt == UIC_BF16) {
    UIC_TRY(uic_cast_f32_launch(dt, b->fc_feats, L.fcT, (size_t)N * d->Dfc, s));
    fc_in = L.fcT;
  }
  {  // step 0 input: xt = img_embed(fc_feats) (:96-97)
    UicGemmParams g = gemm_base(dt, N, E);
    add_seg(g, fc_in, d->Dfc, L.img_w, d->Dfc, d->Dfc);
    g.C = L.x_all; g.ldc = E; g.bias = w->img_embed_b;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  // steps i >= 1: xt = embed(labels[:, i-1]) -- a bare nn.Embedding (:79,118)
  UIC_TRY(uic_embed_fwd_launch(dt, w->embed_w, V1, E, b->labels, b->ld_labels, N, s_run - 1, 0.f, 0, 0, 0, 0,
                               offw(L.x_all, (size_t)N * E, dt), s));
  {  // i2h over all steps (+ both biases)
    UicGemmParams g = gemm_base(dt, s_run * N, H5);
    add_seg(g, L.x_all, E, L.i2h_w, E, E);
    g.C = L.gx; g.ldc = H5; g.bias = w->i2h_b; g.bias2 = w->h2h_b; g.flags = UIC_GEMM_OUT_F32;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  UIC_TRY(uic_fill_launch(L.h_buf, 0, NH * Sz, s));
  UIC_TRY(uic_fill_launch(L.c_buf, 0, NH * 4, s));
  for (int t = 0; t < s_run; ++t)
    UIC_TRY(fc_core(*d, L, nullptr, w, L.gx + (size_t)t * N * H5, off(L.h_buf, t * NH, dt), L.c_buf + t * NH,
                    L.c_buf + (t + 1) * NH, offw(L.h_buf, (t + 1) * NH, dt), offw(L.gates, (size_t)t * N * H5, dt),
                    drop_p, seed, t, s));
  {  // logits of steps 1 .. s_run-1 (outputs[:, 1:], :121-124)
    UicGemmParams g = gemm_base(dt, (s_run - 1) * N, V1);
    add_seg(g, off(L.h_buf, 2 * NH, dt), H, L.logit_w, H, H);
    g.C = L.logits; g.ldc = V1p; g.bias = w->logit_b; g.flags = UIC_GEMM_OUT_F32;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  if (logprobs_out) {
    UicXeParams x;
    memset(&x, 0, sizeof(x));
    x.dtype = dt; x.M = (s_run - 1) * N; x.V1 = V1; x.ldv = V1p; x.logits = L.logits; x.N = N;
    x.logprobs = logprobs_out; x.lp_step_stride = V1; x.lp_row_stride = (size_t)(d->S - 1) * V1;
    UIC_TRY(uic_xe_launch(x, s));
  }
  return UIC_OK;
}

int uic_fc_xe_loss(const uic_fc_dims* d, const uic_topdown_batch* b, int32_t s_run, void* workspace, const float* inv_den,
                   float* loss_out, void* stream) {
  UIC_TRY(fc_check(d));
  UIC_REQUIRE(b && workspace && b->labels && b->masks && loss_out, "fc_xe_loss: null pointer");
  UIC_REQUIRE(s_run >= 2 && s_run <= d->S, "fc_xe_loss: s_run=%d outside [2,%d]", s_run, d->S);
  UIC_REQUIRE(b->ld_labels >= d->S && b->ld_masks >= d->S, "fc_xe_loss: labels/masks need %d columns", d->S);
  hipStream_t s = (hipStream_t)stream;
  const FcLayout L = fc_layout(*d, nullptr, workspace);
  const int N = d->N, V1 = d->V1, V1p = (int)vpad(V1), T = d->S - 1;
  UIC_TRY(uic_masked_sum_launch(nullptr, b->masks, b->ld_masks, 1, N, T, L.scalars, L.scalars + 1, s));
  const float* inv = inv_den ? inv_den : L.scalars + 1;
  UicXeParams x;
  memset(&x, 0, sizeof(x));
  x.dtype = d->dtype; x.M = (s_run - 1) * N; x.V1 = V1; x.ldv = V1p; x.logits = L.logits; x.dlogits = L.dlogits; x.N = N;
  x.target = b->labels; x.ldtarget = b->ld_labels; x.target_col0 = 1;
  x.mask = b->masks; x.ldmask = b->ld_masks; x.mask_col0 = 1;
  x.inv_den = inv; x.row_loss = L.row_loss; x.write_grad = 1;
  UIC_TRY(uic_xe_launch(x, s));
  return uic_reduce_sum_launch(L.row_loss, (size_t)(s_run - 1) * N, 0.f, inv, loss_out, s);
}

int uic_fc_backward(const uic_fc_dims* d, const uic_fc_weights* w, const uic_topdown_batch* b, int32_t s_run,
                    int32_t training, uint32_t seed, void* workspace, const float* dlogprobs, const float* logprobs,
                    const uic_fc_weights* G, void* stream) {
  UIC_TRY(fc_check(d));
  UIC_REQUIRE(w && b && workspace && G && b->fc_feats && b->labels, "fc_backward: null pointer");
  UIC_REQUIRE(s_run >= 2 && s_run <= d->S, "fc_backward: s_run=%d outside [2,%d]", s_run, d->S);
  UIC_REQUIRE(!dlogprobs || logprobs, "fc_backward: dlogprobs needs the forward log-probs");
  hipStream_t s = (hipStream_t)stream;
  const FcLayout L = fc_layout(*d, w, workspace);
  const int dt = d->dtype, N = d->N, H = d->H, E = d->E, V1 = d->V1, V1p = (int)vpad(V1), H5 = 5 * H, Dfc = d->Dfc;
  const size_t NH = (size_t)N * H;
  const float drop_p = training ? d->drop_p : 0.f;
  const int Ml = (s_run - 1) * N, Ms = s_run * N;
  const void* fc_in = dt == UIC_BF16 ? L.fcT : (const void*)b->fc_feats;
  if (dlogprobs)
    UIC_TRY(uic_logsoftmax_bwd_launch(dt, L.dlogits, Ml, V1, V1p, N, dlogprobs, (size_t)V1, (size_t)(d->S - 1) * V1, logprobs, s));
  {  // d h (logit path) for steps 1 .. s_run-1
    UicGemmParams g = gemm_base(dt, Ml, H);
    add_seg(g, L.dlogits, V1p, L.logit_wT, V1p, V1p);
    g.C = L.dh_all; g.ldc = H; g.flags = UIC_GEMM_OUT_F32;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  {
    const UicGemmTnSeg seg{off(L.h_buf, 2 * NH, dt), H, H};
    const WDest d1{G->logit_w, H, 0, H};
    UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.dlogits, V1p, V1, &seg, 1, Ml, &d1, 1, s, false, L.tA, L.tB));
  }
  UIC_TRY(uic_colsum_launch(dt, L.dlogits, Ml, V1, V1p, G->logit_b, L.colscratch, L.colscratch_floats, s));
  // BPTT
  UIC_TRY(uic_fill_launch(L.dc, 0, NH * 4, s));
  for (int t = s_run - 1; t >= 0; --t) {
    UicLstmBwdParams p;
    memset(&p, 0, sizeof(p));
    p.dtype = dt; p.M = N; p.H = H;
    if (t >= 1) { p.dh0 = L.dh_all + (size_t)(t - 1) * NH; p.lddh0 = H; }
    if (t < s_run - 1) { p.dh1 = L.dhrec; p.lddh1 = H; }
    p.drop_p = drop_p; p.seed = seed; p.site = UIC_SITE_OUT0 + (unsigned)t;
    p.dc = L.dc; p.gates = off(L.gates, (size_t)t * N * H5, dt);
    p.c_prev = L.c_buf + t * NH; p.c = L.c_buf + (t + 1) * NH;
    p.dgates = offw(L.ds_all, (size_t)t * N * H5, dt);
    UIC_TRY(uic_maxout_lstm_bwd_launch(p, s));
    if (t > 0) {
      UicGemmParams g = gemm_base(dt, N, H);
      add_seg(g, off(L.ds_all, (size_t)t * N * H5, dt), H5, L.h2hT, H5, H5);
      g.C = L.dhrec; g.ldc = H; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
    }
  }
  // core weights: dS^T [5H, S*N] x [h_prev | x]^T in one GEMM
  {
    const UicGemmTnSeg segs[2] = {{L.h_buf, H, H}, {L.x_all, E, E}};
    const WDest dd[2] = {{G->h2h_w, H, 0, H}, {G->i2h_w, E, H, E}};
    UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.ds_all, H5, H5, segs, 2, Ms, dd, 2, s, false, L.tA, L.tB));
  }
  UIC_TRY(uic_colsum_launch(dt, L.ds_all, Ms, H5, H5, G->i2h_b, L.colscratch, L.colscratch_floats, s));
  UIC_TRY(uic_copy_launch(G->h2h_b, G->i2h_b, (size_t)H5 * 4, s));
  {  // d x for all steps
    UicGemmParams g = gemm_base(dt, Ms, E);
    add_seg(g, L.ds_all, H5, L.i2hT, H5, H5);
    g.C = L.dx_all; g.ldc = E; g.flags = UIC_GEMM_OUT_F32;
    UIC_TRY(uic_gemm_launch(g, s));
  }
  // (bucketed by word, one owner per table row: bit-reproducible, no floating-point atomics -- csrc/pointwise.hip)
  UIC_TRY(uic_embed_bwd_sorted_launch(dt, L.dx_all + (size_t)N * E, nullptr, b->labels, b->ld_labels, N, s_run - 1, V1, E, 0.f, -1, G->embed_w,
                                      L.embed_scratch, s));
  // img_embed from d x_0
  UIC_TRY(uic_cast_f32_launch(dt, L.dx_all, L.dx0, (size_t)N * E, s));
  {
    const UicGemmTnSeg seg{fc_in, Dfc, Dfc};
    const WDest d1{G->img_embed_w, Dfc, 0, Dfc};
    UIC_TRY(wgrad_group(L.slab, L.slab_bytes, dt, L.dx0, E, E, &seg, 1, N, &d1, 1, s, false, L.tA, L.tB));
  }
  return uic_colsum_launch(UIC_F32, L.dx_all, N, E, E, G->img_embed_b, L.colscratch, L.colscratch_floats, s);
}

int uic_fc_sample(const uic_fc_dims* d, const uic_fc_weights* w, const uic_topdown_batch* b, int32_t Lsteps,
                  int32_t sample_max, float temperature, uint32_t seed, const int64_t* forced, void* workspace,
                  int64_t* seq, float* seq_logp, void* stream) {
  UIC_TRY(fc_check(d));
  UIC_REQUIRE(w && b && workspace && seq && seq_logp && b->fc_feats, "fc_sample: null pointer");
  UIC_REQUIRE(Lsteps >= 1 && Lsteps + 1 <= d->S, "fc_sample: L=%d needs S >= %d", Lsteps, Lsteps + 1);
  UIC_REQUIRE(temperature > 0.f, "fc_sample: temperature must be positive");
  hipStream_t s = (hipStream_t)stream;
  const FcLayout L = fc_layout(*d, w, workspace);
  const int dt = d->dtype, N = d->N, H = d->H, E = d->E, V1 = d->V1, V1p = (int)vpad(V1);
  const size_t Sz = uic_dtype_size(dt), NH = (size_t)N * H;
  UIC_TRY(fc_refresh(*d, w, L, s));
  const void* fc_in = b->fc_feats;
  if (dt == UIC_BF16) {
    UIC_TRY(uic_cast_f32_launch(dt, b->fc_feats, L.fcT, (size_t)N * d->Dfc, s));
    fc_in = L.fcT;
  }
  UIC_TRY(uic_fill_launch(L.s_h[0], 0, NH * Sz, s));
  UIC_TRY(uic_fill_launch(L.s_c[0], 0, NH * 4, s));
  UIC_TRY(uic_fill_launch(L.s_it, 0, (size_t)N * 8, s));            // <bos> at step 1 (:183-184)
  UIC_TRY(uic_fill_launch(L.s_unf, 0, (size_t)N * 4, s));
  UIC_TRY(uic_fill_launch(L.s_nunf, 0, (size_t)(d->S + 2) * UIC_NUNF_STRIPES * 4, s));
  const int ld = Lsteps + 1;                                         // seq / seqLogprobs are [N, L+1] (:176-177)
  UIC_TRY(uic_fill_launch(seq, 0, (size_t)N * ld * 8, s));
  UIC_TRY(uic_fill_launch(seq_logp, 0, (size_t)N * ld * 4, s));
  for (int t = 0; t <= Lsteps; ++t) {                                // core steps 0 .. L; decisions after steps 1 .. L
    const int cur = t & 1, nxt = cur ^ 1;
    if (t == 0) {
      UicGemmParams g = gemm_base(dt, N, E);
      add_seg(g, fc_in, d->Dfc, L.img_w, d->Dfc, d->Dfc);
      g.C = L.s_xt; g.ldc = E; g.bias = w->img_embed_b;
      UIC_TRY(uic_gemm_launch(g, s));
    } else {
      UIC_TRY(uic_embed_fwd_launch(dt, w->embed_w, V1, E, L.s_it, 1, N, 1, 0.f, 0, 0, 0, 0, L.s_xt, s));
    }
    UIC_TRY(fc_core(*d, L, L.s_xt, w, nullptr, L.s_h[cur], L.s_c[cur], L.s_c[nxt], L.s_h[nxt], nullptr, 0.f, 0, t, s));
    if (t >= 1) {
      UicGemmParams g = gemm_base(dt, N, V1);
      add_seg(g, L.s_h[nxt], H, L.logit_w, H, H);
      g.C = L.s_logits; g.ldc = V1p; g.bias = w->logit_b; g.flags = UIC_GEMM_OUT_F32;
      UIC_TRY(uic_gemm_launch(g, s));
      UicSampleParams p;
      memset(&p, 0, sizeof(p));
      p.dtype = dt; p.N = N; p.V1 = V1; p.ldv = V1p; p.t = t - 1; p.L = Lsteps;
      p.logits = L.s_logits; p.sample_max = sample_max; p.temperature = temperature; p.seed = seed;
      p.seq = seq; p.seq_logp = seq_logp; p.it = L.s_it; p.unfinished = L.s_unf; p.n_unfinished = L.s_nunf;
      p.forced = forced; p.fc_mode = 1; p.ld_out = ld;
      UIC_TRY(uic_sample_step_launch(p, s));
    }
  }
  return uic_sample_fixup_launch(N, Lsteps, ld, L.s_nunf, seq, seq_logp, s);
}


int uic_fc_sample_beam(const uic_fc_dims* d, const uic_fc_weights* w, const uic_topdown_batch* b, int32_t Lsteps, int32_t beam_size,
                       int32_t decoding_constraint, int32_t max_ppl, void* workspace, int64_t* seq, float* seq_logp, void* stream) {
  UIC_TRY(fc_check(d));
  UIC_REQUIRE(w && b && workspace && seq && seq_logp && b->fc_feats, "fc_sample_beam: null pointer");
  UIC_REQUIRE(Lsteps >= 1 && Lsteps <= d->S, "fc_sample_beam: L=%d outside [1,%d]", Lsteps, d->S);
  UIC_REQUIRE(beam_size >= 1 && beam_size <= UIC_BEAM_MAX && beam_size <= d->V1, "fc_sample_beam: beam_size=%d outside [1, %d]", beam_size, UIC_BEAM_MAX);
  UIC_REQUIRE(d->N % beam_size == 0, "fc_sample_beam: N=%d rows must be images x beam_size=%d", d->N, beam_size);
  hipStream_t s = (hipStream_t)stream;
  const FcLayout L = fc_layout(*d, w, workspace);
  const int dt = d->dtype, N = d->N, H = d->H, E = d->E, V1 = d->V1, V1p = (int)vpad(V1), S = d->S;
  const size_t Sz = uic_dtype_size(dt), NH = (size_t)N * H;
  UIC_TRY(fc_refresh(*d, w, L, s));
  const void* fc_in = b->fc_feats;
  if (dt == UIC_BF16) {
    UIC_TRY(uic_cast_f32_launch(dt, b->fc_feats, L.fcT, (size_t)N * d->Dfc, s));
    fc_in = L.fcT;
  }
  UIC_TRY(uic_fill_launch(L.s_h[0], 0, NH * Sz, s));
  UIC_TRY(uic_fill_launch(L.s_c[0], 0, NH * 4, s));
  UIC_TRY(uic_fill_launch(L.s_it, 0, (size_t)N * 8, s));            // <bos> (:151-153)
  for (int i = 0; i < 2; ++i) {
    UIC_TRY(uic_fill_launch(L.bm_seq[i], 0, (size_t)N * S * 8, s));
    UIC_TRY(uic_fill_launch(L.bm_lp[i], 0, (size_t)N * S * 4, s));
  }
  UIC_TRY(uic_fill_launch(L.bm_sum, 0, (size_t)N * 4, s));
  UIC_TRY(uic_fill_launch(L.bm_done_count, 0, (size_t)N * 4, s));
  auto logits = [&](const void* h) -> int {
    UicGemmParams g = gemm_base(dt, N, V1);
    add_seg(g, h, H, L.logit_w, H, H);
    g.C = L.s_logits; g.ldc = V1p; g.bias = w->logit_b; g.flags = UIC_GEMM_OUT_F32;
    return uic_gemm_launch(g, s);
  };
  {  // warm-up step 0: the image embedding (:149-150), state slot 0 -> 1
    UicGemmParams g = gemm_base(dt, N, E);
    add_seg(g, fc_in, d->Dfc, L.img_w, d->Dfc, d->Dfc);
    g.C = L.s_xt; g.ldc = E; g.bias = w->img_embed_b;
    UIC_TRY(uic_gemm_launch(g, s));
    UIC_TRY(fc_core(*d, L, L.s_xt, w, nullptr, L.s_h[0], L.s_c[0], L.s_c[1], L.s_h[1], nullptr, 0.f, 0, 0, s));
  }
  // warm-up step 1: <bos> (:151-156), slot 1 -> 0; its log-probs open the search
  UIC_TRY(uic_embed_fwd_launch(dt, w->embed_w, V1, E, L.s_it, 1, N, 1, 0.f, 0, 0, 0, 0, L.s_xt, s));
  UIC_TRY(fc_core(*d, L, L.s_xt, w, nullptr, L.s_h[1], L.s_c[1], L.s_c[0], L.s_h[0], nullptr, 0.f, 0, 1, s));
  UIC_TRY(logits(L.s_h[0]));
  UicBeamParams p;
  memset(&p, 0, sizeof(p));
  p.n_img = N / beam_size; p.B = beam_size; p.L = Lsteps; p.V1 = V1; p.ldv = V1p;
  p.decoding_constraint = decoding_constraint; p.max_ppl = max_ppl;
  p.logits = L.s_logits; p.cand_val = L.bm_cand_val; p.cand_idx = L.bm_cand_idx;
  p.beam_seq_hist[0] = L.bm_seq[0]; p.beam_seq_hist[1] = L.bm_seq[1]; p.beam_lp_hist[0] = L.bm_lp[0]; p.beam_lp_hist[1] = L.bm_lp[1];
  p.beam_sum = L.bm_sum; p.parent = L.bm_parent; p.it = L.s_it;
  p.done_count = L.bm_done_count; p.done_p = L.bm_done_p; p.done_seq = L.bm_done_seq; p.done_lp = L.bm_done_lp;
  for (int t = 0; t < Lsteps; ++t) {
    p.t = t;
    UIC_TRY(uic_beam_step_launch(p, s));
    if (t + 1 == Lsteps) break;
    UIC_TRY(uic_beam_gather_launch(dt, L.bm_parent, N, beam_size, H, L.s_h[0], L.s_h[1], nullptr, nullptr, L.s_c[0], L.s_c[1], nullptr, nullptr, s));
    // get_logprobs_state (:126-134): embed -> core -> log_softmax(logit), slot 1 -> 0
    UIC_TRY(uic_embed_fwd_launch(dt, w->embed_w, V1, E, L.s_it, 1, N, 1, 0.f, 0, 0, 0, 0, L.s_xt, s));
    UIC_TRY(fc_core(*d, L, L.s_xt, w, nullptr, L.s_h[1], L.s_c[1], L.s_c[0], L.s_h[0], nullptr, 0.f, 0, t + 2, s));
    UIC_TRY(logits(L.s_h[0]));
  }
  return uic_beam_final_launch(p, seq, seq_logp, s);
}

}  // extern "C"
