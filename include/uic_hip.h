/* libuic_hip.so -- C ABI of the MI355X (gfx950) TopDown captioner hot path.
 *
 * The reference (gujiuxiang/unpaired_image_captioning, tree pivot_based_eccv2018/, "P/" below)
 * has no native boundary: its hot path is Python nn.Module code calling stock PyTorch ops.
 * Each entry point therefore names the reference Python call site it replaces.  All pointers
 * are device pointers unless stated; the caller owns every buffer (the library never allocates,
 * frees or retains memory); every call only enqueues work on `stream` (hipStream_t passed as
 * void*), never synchronises, and returns 0 on success, a negative value for an argument error
 * and a positive hipError_t otherwise.  uic_last_error_string() describes the last failure on
 * the calling thread.  dtype: 0 = f32 operands (exact-f32 MFMA, parity path), 1 = bf16 operands
 * (bf16 MFMA, f32 accumulation / state / loss).
 * Threading: the only state the library keeps is one lazily created set of side streams / events per device (mutex
 * guarded creation) used by the fused steps; like any stream-ordered resource it is driven by one host thread per
 * device at a time -- different devices (one process or thread per GPU) are independent.
 */
#ifndef UIC_HIP_H
#define UIC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UIC_DTYPE_F32 0
#define UIC_DTYPE_BF16 1

const char* uic_last_error_string(void);
int uic_version(void);

/* Persistent recurrence (csrc/rnn_persist.hip, csrc/rnn_bwd_persist.hip): with rnn_size = att_hid_size = 512 and at most 40
 * regions the decode loop of AttModel._forward (P/models/AttModel.py:129-154: att_lstm, h2att, attention, lang_lstm per
 * step) runs as ONE launch for a range of decode steps instead of four dependent launches per step (bf16: the recurrent
 * weights stay in LDS and registers for the whole launch; it holds every CU while it runs), in uic_topdown_forward and in
 * uic_topdown_xe_train_step, whose logit layer then follows the recurrence chunk by chunk, last chunk first, beside the BPTT
 * loop.  How a call launches the loop is part of its dims (uic_topdown_dims.recurrence, no library-wide state):
 *   UIC_REC_FWD_CHAIN    the forward recurrence (and the decode loop of uic_topdown_sample / _sample_train) as per-step
 *                        launches.  Two PROCESSES sharing one GPU must both set it: a persistent kernel's workgroups have to
 *                        be resident together.  Inside one process the library makes every persistent launch wait for the
 *                        previous one on its device, whatever streams they are on, so two passes meant to run side by side
 *                        (the self-critical step's sampling pass and greedy baseline) want the chains
 *   UIC_REC_BWD_PERSIST  (uic_topdown_backward only) the BPTT loop as one persistent launch (bf16) instead of five launches
 *                        per step: alone it runs a step in 57 us instead of 82.  uic_topdown_xe_train_step ignores it -- its
 *                        side stream's GEMMs cannot run beside a kernel that holds every CU (3.81 vs 3.48 ms when tried)
 *   UIC_REC_SAFE         the persistent kernels use their placement-independent exchange protocol (device-scope stores, groups
 *                        by arrival order) instead of XCD-local groups -- same results bit for bit; tests
 *   UIC_REC_STAMPS       the persistent kernels write per-phase time stamps into the workspace ("rnn_dbg" / "rnn_bwd_dbg" of
 *                        uic_topdown_workspace_ptr: [256 workgroups][T][16] uint64, 100 MHz); tools/
 *   UIC_REC_EARLY_GRADS  (uic_topdown_xe_train_step) the order of the gradient work that makes most gradient bytes final EARLY:
 *                        the recurrent weight gradients of a chunk run on two side streams, the embedding gradient of decode
 *                        steps >= 4 is gathered during the BPTT loop, the fc' columns of att_lstm.weight_ih and the rest of the
 *                        embedding table come first after it.  48.8 of 78.8 MB are then final 0.17 ms before the step ends and
 *                        10.5 MB in its last 0.1 ms (default order: 16.8 MB at -0.24 ms, 42.5 MB in the last 0.1 ms), for a step
 *                        that is 0.12 ms (4 %) longer on its own -- the chip is busy either way (profiles/LOG.md).  For a
 *                        data-parallel caller to choose; same gradients up to f32 summation order
 *   UIC_REC_NO_F32A      att_embed (P/models/AttModel.py:76-80,111-115) as a cast pass over the f32 region features + the bf16
 *                        GEMM instead of the GEMM that rounds its f32 A operand itself (csrc/gemm_pp.hip): bit-identical
 *                        results, ~45 us more per step.  The fallback for that kernel's hand-counted register loads (its build
 *                        is audited by tools/audit_f32a_asm.py and soaked in the GPU suite)
 *   UIC_REC_COMM_STREAM  (uic_topdown_xe_train_step) the caller keeps a communication stream busy beside the step -- the
 *                        overlapped gradient exchange of a data-parallel run (uic_topdown_grad_ready_wait).  MI355X dispatches
 *                        from three busy hardware queues at full speed; with a fourth non-empty one -- even a single wave that
 *                        only waits for an event -- the BPTT loop's dependent launches take 1.5-2.5x as long (measured:
 *                        profiles/r05_*_queue_probe.txt, r05_*_comm_proxy.txt).  With this flag the step uses two queues of its
 *                        own while the loop runs (the chunks' weight gradients behind the logit layer on the side stream)
 *                        instead of three; without a communication stream the three-queue order is 0.07 ms faster
 * uic_topdown_dims.rnn_status: NULL, or 4 caller-allocated, caller-zeroed uint32 on the device that the persistent kernels
 * update: [0] != 0 after a bounded spin timed out (the results of that call are invalid), [1] / [2] launches that ran with
 * the XCD-local / the SAFE protocol. */
#define UIC_REC_FWD_CHAIN 1
#define UIC_REC_BWD_PERSIST 2
#define UIC_REC_SAFE 4
#define UIC_REC_STAMPS 8
#define UIC_REC_EARLY_GRADS 16
#define UIC_REC_NO_F32A 32
#define UIC_REC_COMM_STREAM 64

/* ---- shapes of one TopDown step (P/models/AttModel.py:56-92,422-428,530-536) ---- */
typedef struct uic_topdown_dims {
  int32_t N;        /* caption rows = batch_size * seq_per_img (P/misc/dataloader/dataloader.py:231) */
  int32_t R;        /* regions per image (att_feats.size(1)) */
  int32_t D;        /* att_feat_size; a multiple of 8 -- for 2048 + 5 box features pad feature rows and the columns of
                     * att_embed's Linear / BatchNorm with zeros (exact; the Python engine does it) */
  int32_t Dfc;      /* fc_feat_size */
  int32_t H;        /* rnn_size */
  int32_t E;        /* input_encoding_size */
  int32_t A;        /* att_hid_size */
  int32_t V1;       /* vocab_size + 1 */
  int32_t T;        /* decode steps the workspace is sized for (labels.size(1) - 1) */
  int32_t dtype;    /* UIC_DTYPE_* */
  float drop_p;     /* drop_prob_lm; applied only when `training` is non-zero */
  int32_t use_bn;   /* opt.use_bn (P/opts.py:52): 0 none, 1 BatchNorm1d(D) in front of att_embed's Linear, 2 also
                     * BatchNorm1d(H) after its Dropout (P/models/AttModel.py:78-84) */
  int32_t seq_per_img; /* 0 or 1: the batch carries one feature row per CAPTION row, replicated by the loader as the
                     * reference does on the host (P/misc/dataloader/dataloader.py:270-277).  S > 1 (must divide N): the
                     * batch carries fc_feats / att_feats / att_masks once per IMAGE ([N/S, ...]); caption row n uses image
                     * n / S.  Results are those of the S-fold replicated batch (same dropout masks per caption row); the
                     * att_embed [BatchNorm +] Linear and its weight gradient then run on N/S * R rows instead of N * R
                     * (batch statistics: every image row counts S times). */
  int32_t logit_layers; /* opt.logit_layers (P/models/AttModel.py:86-91): 0 or 1 = logit is one Linear(H, V1); n > 1 (at most
                     * UIC_MAX_LOGIT_LAYERS) = n - 1 blocks Linear(H, H) + ReLU + Dropout(0.5) in front of it (the 0.5 is
                     * hard-coded in the reference and active in train mode whatever drop_p is). */
  int32_t recurrence;  /* UIC_REC_* flags (above); 0 = default */
  uint32_t* rnn_status; /* status words of the persistent kernels (above) or NULL */
} uic_topdown_dims;
#define UIC_MAX_LOGIT_LAYERS 4

/* Master parameters (f32), one pointer per tensor of TopDownModel.state_dict(), same shapes as the reference
 * (with use_bn >= 1 the Linear of att_embed sits at att_embed.1).  The same struct
 * carries the gradient pointers on the way back (running statistics have no gradient: those pointers are ignored).
 * `training` arguments below: bit 0 = train mode (dropout on, BatchNorm batch statistics); bit 1 = do NOT update the
 * BatchNorm running statistics (the teacher-forced replay of a sampled caption in self-critical training re-runs
 * the forward of an iteration whose sampling pass already updated them); bit 2 (uic_topdown_xe_train_step only) = the
 * workspace already holds this forward pass, left there by uic_topdown_sample_train: start at the criterion. */
typedef struct uic_topdown_weights {
  float* embed_w;         /* embed.0.weight            [V1, E]        */
  float* fc_w;            /* fc_embed.0.weight         [H, Dfc]       */
  float* fc_b;            /* fc_embed.0.bias           [H]            */
  float* att_w;           /* att_embed.0.weight        [H, D]         */
  float* att_b;           /* att_embed.0.bias          [H]            */
  float* logit_w;         /* logit.weight              [V1, H]        */
  float* logit_b;         /* logit.bias                [V1]           */
  float* ctx2att_w;       /* ctx2att.weight            [A, H]         */
  float* ctx2att_b;       /* ctx2att.bias              [A]            */
  float* att_lstm_w_ih;   /* core.att_lstm.weight_ih   [4H, E + 2H]   columns: [h_lang | fc | xt] */
  float* att_lstm_w_hh;   /* core.att_lstm.weight_hh   [4H, H]        */
  float* att_lstm_b_ih;   /* core.att_lstm.bias_ih     [4H]           */
  float* att_lstm_b_hh;   /* core.att_lstm.bias_hh     [4H]           */
  float* lang_lstm_w_ih;  /* core.lang_lstm.weight_ih  [4H, 2H]       columns: [att_res | h_att]  */
  float* lang_lstm_w_hh;  /* core.lang_lstm.weight_hh  [4H, H]        */
  float* lang_lstm_b_ih;  /* core.lang_lstm.bias_ih    [4H]           */
  float* lang_lstm_b_hh;  /* core.lang_lstm.bias_hh    [4H]           */
  float* h2att_w;         /* core.attention.h2att.weight     [A, H]   */
  float* h2att_b;         /* core.attention.h2att.bias       [A]      */
  float* alpha_w;         /* core.attention.alpha_net.weight [1, A]   */
  float* alpha_b;         /* core.attention.alpha_net.bias   [1]      */
  /* use_bn >= 1 (NULL otherwise): att_embed.0 = BatchNorm1d(D), updated in place by a training-mode forward */
  float* att_bn0_w;       /* att_embed.0.weight        [D]            */
  float* att_bn0_b;       /* att_embed.0.bias          [D]            */
  float* att_bn0_rm;      /* att_embed.0.running_mean  [D]            */
  float* att_bn0_rv;      /* att_embed.0.running_var   [D]            */
  /* use_bn == 2 (NULL otherwise): att_embed.4 = BatchNorm1d(H) */
  float* att_bn4_w;       /* att_embed.4.weight        [H]            */
  float* att_bn4_b;       /* att_embed.4.bias          [H]            */
  float* att_bn4_rm;      /* att_embed.4.running_mean  [H]            */
  float* att_bn4_rv;      /* att_embed.4.running_var   [H]            */
  /* logit_layers = n > 1 (NULL otherwise): hidden block l = 0 .. n-2 is logit.{3l} = Linear(H, H); logit_w / logit_b above
   * are then logit.{3(n-1)}, the final Linear(H, V1) */
  float* logit_h_w[UIC_MAX_LOGIT_LAYERS - 1];   /* logit.{3l}.weight [H, H] */
  float* logit_h_b[UIC_MAX_LOGIT_LAYERS - 1];   /* logit.{3l}.bias   [H]    */
} uic_topdown_weights;

/* The batch dict of DataLoader.get_batch as consumed at P/trainer.py:147-149 (device copies). */
typedef struct uic_topdown_batch {
  const float* fc_feats;    /* [N, Dfc]   ([N / seq_per_img, Dfc]  when dims.seq_per_img > 1) */
  const float* att_feats;   /* [N, R, D]  ([N / seq_per_img, R, D]) */
  const float* att_masks;   /* [N, R] or NULL  ([N / seq_per_img, R]) */
  const int64_t* labels;    /* [N, ld_labels]; column 0 is BOS = 0 */
  int32_t ld_labels;
  const float* masks;       /* [N, ld_masks] or NULL (forward only) */
  int32_t ld_masks;
  const float* grad_scale;  /* optional [N, ld_grad_scale]: d loss / d logprob[n, t, labels[n, t+1]] given directly (column t),
                               replacing mask / sum(mask) -- the self-critical step (P/trainer.py:167-171) */
  int32_t ld_grad_scale;
  float ss_prob;            /* scheduled-sampling probability (model.ss_prob, P/models/AttModel.py:130-143; applied in train
                               mode at steps >= 1): with this probability a row's input token is replaced by a draw from
                               exp(previous step's log-probs).  The inputs actually used stay in the workspace
                               ("tok_used", [N, T] int64) for the backward pass and for the tests. 0 = teacher forcing. */
  /* Optional outputs of uic_topdown_backward / uic_topdown_xe_train_step (NULL = not computed, the default of every
     reference call site: the reference's features are data).  Gradients of the loss w.r.t. the input features, for an
     encoder in front of the captioner (BASELINE configs[4]'s scene-graph GCN, uic_gcn_backward's `dout`): */
  float* d_att_feats;       /* [N, R, D]  ([N / seq_per_img, R, D]); padded regions get zeros.  Needs dims.use_bn == 0 */
  float* d_fc_feats;        /* [N, Dfc]   ([N / seq_per_img, Dfc]) */
  /* Optional list of the LIVE positions of the batch, for uic_topdown_xe_train_step (live_count NULL = every position is computed, and
     positions with mask 0 contribute exact zeros -- the same result).  A position (t, n) -- decode step t, row n -- is live
     when masks[n, 1 + t] != 0; positions behind a caption's end (LanguageModelCriterion multiplies them by 0,
     P/misc/utils.py:62-73) are a quarter of the benchmark's batch and a third of COCO's.  With the list the logit layer, the
     criterion and their gradients run over the listed rows only; with the list made by the step itself (live_rows NULL) the
     recurrence also stores the listed rows compactly, and the BPTT loop's attention backward and the deferred attention
     accumulation skip, per row, the decode steps behind the row's last live position.  The loss and the gradients are those of
     the full computation up to floating-point summation order (sums run over fewer, differently grouped terms).
       live_count  HOST int32 [t_run]: the number of live positions of each step -- the step sizes its launches with them
       live_rows   DEVICE int32 [roundup(sum(live_count), 128)] or NULL: t * N + n of every live position, step-major (all of
                   step 0, then step 1, ...; any order within a step); the tail up to the multiple of 128 holds -1.
                   NULL (what Trainer passes): the step compacts `masks` itself with one small launch beside its prologue, so
                   nothing but the counts has to be made on the host
     The caller counts from the masks it already holds on the host (DataLoader.get_batch makes the masks there).  The counts must
     be those of `masks`; a list that omits a position whose mask is not zero drops that position's loss and gradient; listing a
     masked position is harmless.
     Ignored (every position computed) under scheduled sampling, with grad_scale, with logit_layers > 1, without masks, when a row
     of hidden units is not a multiple of 16 bytes, when t_run >
     UIC_MAX_LIVE_STEPS, or when a count is outside [0, N]. */
  const int32_t* live_rows;
  const int32_t* live_count;
} uic_topdown_batch;
#define UIC_MAX_LIVE_STEPS 64

/* Sizes (bytes) of the two caller-allocated arenas. */
size_t uic_topdown_workspace_bytes(const uic_topdown_dims* d);
size_t uic_topdown_derived_bytes(const uic_topdown_dims* d);

/* Rebuild the operand-dtype / transposed weight copies in `derived` from the f32 masters.
 * Must run after every change of the masters (optimizer step, load_state_dict). */
int uic_topdown_refresh_weights(const uic_topdown_dims* d, const uic_topdown_weights* w, void* derived, void* stream);
/* The same, for a caller whose NEXT call on this device is a uic_topdown_* consumer of `derived` (the training loop: refresh,
 * then uic_topdown_xe_train_step): only the operand-dtype copies are enqueued here; the transposed copies, which backward
 * passes alone read, are enqueued by that next call -- the fused step puts its side-stream prologue in front of them.  `w`'s
 * tensors and `derived` must stay alive and unchanged until then (as they must for that call anyway). */
int uic_topdown_refresh_weights_deferred(const uic_topdown_dims* d, const uic_topdown_weights* w, void* derived, void* stream);

/* Sharded data parallelism (replaces DataParallel's per-step parameter broadcast, P/trainer.py:74): each
 * rank owns 1/world of every gradient piece (reduce-scatter), runs Adam on that shard only (uic_adam_step_ranges, which also
 * leaves the shard's updated weights in the operand dtype) and the ranks all-gather the OPERAND-DTYPE weights -- half the bytes of
 * the f32 masters in bf16 runs -- while the next step's feature projection already runs.  This struct names the gathered tensors
 * (operand dtype, same shapes as the masters; with f32 operands they must BE the masters) and, per gather group, the hipEvent_t the
 * caller recorded behind that group's all-gather (NULL: already ordered on `stream`):
 *   ready[3]: att_embed weight (ignored / NULL with use_bn: the folded weight needs the f32 master, which the caller keeps
 *             replicated), ctx2att.weight, core.attention.h2att.weight   -- consumed first (main branch of the prologue)
 *   ready[2]: embed.0.weight, fc_embed.0.weight, core.att_lstm.weight_ih
 *   ready[1]: core.lang_lstm.weight_{ih,hh}, core.att_lstm.weight_hh     -- the recurrence
 *   ready[0]: logit weight(s)                                            -- the logit layer
 * (the groups are the gradient groups of uic_topdown_grad_ready_wait, in reverse).  Biases and the other small f32 tensors are read
 * from the masters in `w`: the caller keeps those replicated (all-reduce + the same Adam on every rank). */
typedef struct uic_topdown_gathered {
  const void* embed_w; const void* fc_w; const void* att_w; const void* logit_w; const void* ctx2att_w;
  const void* att_lstm_w_ih; const void* att_lstm_w_hh; const void* lang_lstm_w_ih; const void* lang_lstm_w_hh; const void* h2att_w;
  const void* logit_h_w[UIC_MAX_LOGIT_LAYERS - 1];
  void* ready[4];
} uic_topdown_gathered;
/* uic_topdown_refresh_weights from the gathered tensors.  deferred != 0: as uic_topdown_refresh_weights_deferred (the caller's next
 * call on this device is uic_topdown_xe_train_step, which orders its own streams behind the groups they read, so that e.g. the
 * att_embed GEMM starts as soon as group 3 has arrived); deferred == 0: everything is ordered in front of `stream`. */
int uic_topdown_refresh_weights_gathered(const uic_topdown_dims* d, const uic_topdown_weights* w, const uic_topdown_gathered* g,
                                         void* derived, int32_t deferred, void* stream);

/* AttModel._forward (P/models/AttModel.py:119-156) with ss_prob = 0: feature projection, the
 * teacher-forced unroll over `t_run` <= d->T steps (t_run < T reproduces the early break at :151)
 * and the logit GEMM.  If `logprobs_out` != NULL it receives log-probs as [N, T, V1] f32 (rows of
 * steps >= t_run are left untouched: pass a zero-filled tensor).  Activations stay in `workspace`
 * for uic_topdown_xe_loss / uic_topdown_backward. */
int uic_topdown_forward(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                        const uic_topdown_batch* batch, int32_t t_run, int32_t training, uint32_t seed,
                        void* workspace, float* logprobs_out, void* stream);

/* LanguageModelCriterion (P/misc/criterion.py:143-150) fused with log_softmax on the logits left by
 * uic_topdown_forward: loss_out[0] = -sum(logp[target] * mask) / sum(mask) over labels[:,1:], masks[:,1:],
 * and d loss / d logits is left in the workspace for uic_topdown_backward(dlogprobs = NULL).
 * `inv_den` (device, optional) overrides 1/sum(mask) -- data-parallel ranks pass the global value. */
int uic_topdown_xe_loss(const uic_topdown_dims* d, const uic_topdown_batch* batch, int32_t t_run, void* workspace,
                        const float* inv_den, float* loss_out, float* den_out, void* stream);

/* Backward of uic_topdown_forward (what autograd does for P/trainer.py:173).  If `dlogprobs` != NULL it
 * is the dense upstream gradient w.r.t. the [N, T, V1] log-probs and `logprobs` must be the forward output;
 * otherwise the gradient prepared by uic_topdown_xe_loss is used.  Every tensor of `grads` is overwritten. */
int uic_topdown_backward(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                         const uic_topdown_batch* batch, int32_t t_run, int32_t training, uint32_t seed,
                         void* workspace, const float* dlogprobs, const float* logprobs,
                         const uic_topdown_weights* grads, void* stream);

/* One fused XE training step = uic_topdown_forward + uic_topdown_xe_loss + uic_topdown_backward (what
 * P/trainer.py:164-165,173 do), scheduled on TWO HIP streams: the recurrence runs on `stream`, the logit layer
 * of finished decode steps (logit GEMM, log-softmax + criterion, dH, dW_logit) on a library-owned side stream,
 * joined back into `stream` before return.  Results are identical to the three separate calls. */
int uic_topdown_xe_train_step(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                              const uic_topdown_batch* batch, int32_t t_run, int32_t training, uint32_t seed,
                              void* workspace, const float* inv_den, float* loss_out, float* den_out,
                              const uic_topdown_weights* grads, void* stream);
/* Data-parallel overlap: makes `stream` wait until the most recent uic_topdown_xe_train_step on the current device has
 * FINAL gradients for a group of tensors, while that call is still computing the rest on its own streams:
 *   group 0: logit.* (and the loss)                       -- final when the BPTT loop STARTS (~25 % of the bytes);
 *   group 1: core.lang_lstm.weight_{ih,hh}, core.att_lstm.weight_hh       -- final right after the BPTT loop (~21 %; the fc'
 *            columns of core.att_lstm.weight_ih still come from the sum over steps, so that matrix belongs to group 2).
 *            With UIC_REC_EARLY_GRADS in dims.recurrence also core.att_lstm.weight_ih and embed.0.weight (~62 % with them);
 *   group 2: everything except the late group {att_embed.*, ctx2att.*, core.attention.h2att.*,
 *            core.attention.alpha_net.*}                   -- final when the embedding / fc_embed / bias gradients are done;
 *   group 3: embed.0.weight and core.att_lstm.weight_ih (a subset of group 2, 32 of its 36 MB at the reference's sizes), final
 *            ~0.1 ms before the rest of it: the table's gather and the fc' columns run beside the last chunk's weight gradients;
 *   group 4: embed.0.weight alone (19 MB) -- does not wait for the recurrent weight gradients, so it is the first of the tail's
 *            pieces to be final.
 * A caller that lays its flat gradient arena out as [logit | group 1 | rest of the early group | late group] can
 * start the RCCL all-reduce of the first three pieces on a communication stream as each becomes final; the tail follows on
 * the step's stream.  Enqueue-only, no host sync: a hipStreamWaitEvent on the group's event.  (Round 5 also had a polling-kernel
 * form of the wait; measured no different -- what slows the step's launch chain beside a communication stream is the NUMBER of
 * busy hardware queues of the process (> 4; profiles/r05_v4_queue_probe.txt), not the packet type -- and removed in round 6.
 * A data-parallel process should run with GPU_MAX_HW_QUEUES=2 in its environment: Trainer sets / checks it.) */
int uic_topdown_grad_ready_wait(void* stream, int32_t group);
/* Timing marks of the last uic_topdown_xe_train_step on this device (diagnostics; off by default, the step records no timing
 * events then).  enable != 0 switches the marks on for the following steps.  ms_out (optional, UIC_STEP_MARKS floats): waits
 * for the last marked step and writes, in ms since its first launch: [1] feature projection + batched input GEMMs done,
 * [2] recurrence done, [3] side stream: logit layer + loss + d hdrop done, [4] BPTT loop starts, [5] BPTT loop done,
 * [6] side stream: recurrent weight gradients done, [7] main tail (attention accumulation, ctx2att, att_embed) done,
 * [8] side tail (biases, embedding, fc_embed) done, [9] joined, [10] side stream: logit-layer gradients and the loss final
 * (gradient group 0 of uic_topdown_grad_ready_wait; group 1 is [6], group 2 is [8]). */
#define UIC_STEP_MARKS 11
int uic_topdown_step_marks(int32_t enable, float* ms_out);

/* AttModel._sample with beam_size = 1 (P/models/AttModel.py:198-253): greedy (sample_max = 1) or
 * multinomial decode of `L` <= d->T tokens.  seq [N, L] int64 and seq_logp [N, L] f32 are fully written.
 * `forced` (optional, [N, L] int64) replaces the multinomial draws (parity tests).  training != 0 applies the
 * dropout masks of (seed) exactly as uic_topdown_forward would (the sampling pass of the self-critical step).
 * bf16 at the default widths (rnn_size = att_hid_size = input_encoding_size = 512, <= 40 regions, vocabulary <= 10 240,
 * temperature 1, no decoding constraint, one logit layer): ALL decode steps run as one persistent launch (csrc/rnn_persist.hip,
 * decode mode -- embedding, recurrence, logit layer and the choice inside the launch; 88 us per step instead of six launches,
 * ~120 us); UIC_REC_FWD_CHAIN in d->recurrence keeps the per-step launches.  The two forms differ by bf16 summation order
 * only: the same tokens replayed (forced) give log-probs within 5e-3 of each other. */
int uic_topdown_sample(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                       const uic_topdown_batch* batch, int32_t L, int32_t sample_max, float temperature,
                       int32_t decoding_constraint, uint32_t seed, const int64_t* forced, int32_t training,
                       void* workspace, int64_t* seq, float* seq_logp, void* stream);

/* The same pass in the TRAINING layout (the sampling pass of the self-critical step, P/trainer.py:167): the per-step
 * chain of uic_topdown_forward with every step's input tokens drawn from the previous step's distribution, so that all
 * activations the backward pass reads (and the logits) stay in the workspace.  A following
 * uic_topdown_xe_train_step(..., training | 4, same seed, same dims, same workspace, labels = [0, seq, 0], grad_scale)
 * then starts at the criterion instead of replaying the sampled captions teacher-forced.  Same arguments and results
 * as uic_topdown_sample. */
int uic_topdown_sample_train(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                             const uic_topdown_batch* b, int32_t Lsteps, int32_t sample_max, float temperature,
                             int32_t decoding_constraint, uint32_t seed, const int64_t* forced, int32_t training,
                             void* workspace, int64_t* seq, float* seq_logp, void* stream);
/* AttModel._sample_beam + CaptionModel.beam_search with group_size = 1 (P/models/AttModel.py:167-196,
 * P/models/CaptionModel.py:33-177), all images at once.  The batch holds every image REPLICATED beam_size times (row =
 * image * beam_size + beam; N = images * beam_size), eval mode.  Outputs: the best finished beam per image, seq
 * [images, L] int64 and its per-step log-probs [images, L] (as recorded by the reference: after the -1000 on the last
 * vocabulary index and, with decoding_constraint, -inf on the previous word).  max_ppl ranks finished beams by p / length. */
int uic_topdown_sample_beam(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                            const uic_topdown_batch* b, int32_t L, int32_t beam_size, int32_t decoding_constraint,
                            int32_t max_ppl, void* workspace, int64_t* seq, float* seq_logp, void* stream);

/* The rest of uic_topdown_sample_beam's done list, read from the workspace right after it: CaptionModel.beam_search's
 * done_beams_table (P/models/CaptionModel.py:147-161,174-176) in insertion order -- done_count [images] int32 entries per
 * image, done_p [images, L * beam_size] (p, or p / length with max_ppl), done_seq [images, L * beam_size, L] int64,
 * done_lp likewise f32.  The reference's done_beams[k] is these entries sorted by -p (stable), first beam_size kept. */
int uic_topdown_beam_done_lists(const uic_topdown_dims* d, void* workspace, int32_t L, int32_t beam_size, int32_t* done_count,
                                float* done_p, int64_t* done_seq, float* done_lp, void* stream);

/* AttModel._prepare_feature (P/models/AttModel.py:107-117: fc_embed, pack_wrapper(att_embed), ctx2att) as its own call,
 * for callers that drive the decoder step by step (CaptionModel.beam_search :172, eval_ensemble.py): fc_out [N, H],
 * att_out [N, R, H], p_att_out [N, R, A], all f32 (the workspace keeps them in the operand dtype). */
int uic_topdown_prepare_feature(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived,
                                const uic_topdown_batch* batch, int32_t training, uint32_t seed, void* workspace,
                                float* fc_out, float* att_out, float* p_att_out, void* stream);
/* AttModel.get_logprobs_state (P/models/AttModel.py:158-165): ONE decode step from caller-held prepared features and
 * state.  it [N] int64; fc [N, H], att [N, R, H], p_att [N, R, A] f32 as returned by uic_topdown_prepare_feature;
 * att_masks [N, R] or NULL; state in / out as the reference stacks it: h [2, N, H] = (h_att, h_lang), c likewise, f32.
 * logprobs [N, V1] = log_softmax(logit(dropout(h_lang))).  `t` picks the dropout masks of decode step t when training != 0.
 * dims->seq_per_img must be 0 or 1 (the prepared features are per row). */
int uic_topdown_logprobs_state(const uic_topdown_dims* d, const uic_topdown_weights* w, const void* derived, const int64_t* it,
                               const float* fc, const float* att, const float* p_att, const float* att_masks,
                               const float* h_in, const float* c_in, int32_t t, int32_t training, uint32_t seed, void* workspace,
                               float* logprobs, float* h_out, float* c_out, void* stream);

/* Address of a named activation inside the workspace (tests / debugging); NULL if unknown.
 * Names: fc_embed att_embed p_att e_att xt gx h_att h_lang c_att c_lang att_h alpha ctx logits dlogits ...  (e_att, bf16 workspaces
 * only: e^{2 p_att} of the rounded p_att, what the persistent training recurrence's attention reads -- tanh(p + h) = 1 - 2 / (1 + e^{2p} e^{2h})) */
void* uic_topdown_workspace_ptr(const uic_topdown_dims* d, void* workspace, const char* name);

/* ---- FC captioner: the `fc` caption model = FCModel_NMT + maxout LSTMCore (P/models/FCModel_NMT.py:21-217,
 * P/models/__init__.py:24-26; BASELINE config 1) ---- */
typedef struct uic_fc_dims {
  int32_t N;        /* caption rows */
  int32_t Dfc;      /* fc_feat_size */
  int32_t H;        /* rnn_size */
  int32_t E;        /* input_encoding_size */
  int32_t V1;       /* vocab_size + 1 */
  int32_t S;        /* core steps the workspace is sized for = labels.size(1) (image step + L+1 token steps) */
  int32_t dtype;
  float drop_p;
} uic_fc_dims;

typedef struct uic_fc_weights {      /* FCModel_NMT.state_dict() order */
  float* img_embed_w;   /* img_embed.weight  [E, Dfc] */
  float* img_embed_b;   /* img_embed.bias    [E]      */
  float* i2h_w;         /* core.i2h.weight   [5H, E]  chunks (in, forget, out, a, b) */
  float* i2h_b;         /* core.i2h.bias     [5H]     */
  float* h2h_w;         /* core.h2h.weight   [5H, H]  */
  float* h2h_b;         /* core.h2h.bias     [5H]     */
  float* embed_w;       /* embed.weight      [V1, E]  */
  float* logit_w;       /* logit.weight      [V1, H]  */
  float* logit_b;       /* logit.bias        [V1]     */
} uic_fc_weights;

size_t uic_fc_workspace_bytes(const uic_fc_dims* d);
/* FCModel_NMT._forward (:89-124): `s_run` <= S core steps (s_run < S = the early break at :115); log-probs of
 * steps 1.. are written to logprobs_out [N, S-1, V1] if given.  batch: fc_feats, labels (att fields ignored). */
int uic_fc_forward(const uic_fc_dims* d, const uic_fc_weights* w, const uic_topdown_batch* batch, int32_t s_run,
                   int32_t training, uint32_t seed, void* workspace, float* logprobs_out, void* stream);
/* LanguageModelCriterion fused with log_softmax on the logits left by uic_fc_forward. */
int uic_fc_xe_loss(const uic_fc_dims* d, const uic_topdown_batch* batch, int32_t s_run, void* workspace,
                   const float* inv_den, float* loss_out, void* stream);
/* Backward of uic_fc_forward (dense upstream gradient, or the one prepared by uic_fc_xe_loss if dlogprobs == NULL). */
int uic_fc_backward(const uic_fc_dims* d, const uic_fc_weights* w, const uic_topdown_batch* batch, int32_t s_run,
                    int32_t training, uint32_t seed, void* workspace, const float* dlogprobs, const float* logprobs,
                    const uic_fc_weights* grads, void* stream);
/* FCModel_NMT._sample (:164-217, beam_size = 1): seq / seq_logp are [N, L+1] (last column stays 0). */
int uic_fc_sample(const uic_fc_dims* d, const uic_fc_weights* w, const uic_topdown_batch* batch, int32_t L,
                  int32_t sample_max, float temperature, uint32_t seed, const int64_t* forced, void* workspace,
                  int64_t* seq, float* seq_logp, void* stream);
/* FCModel_NMT._sample_beam (P/models/FCModel_NMT.py:136-162) over CaptionModel.beam_search, all images at once; the batch
 * holds every image replicated beam_size times (N = images * beam_size).  seq / seq_logp: [images, L]. */
int uic_fc_sample_beam(const uic_fc_dims* d, const uic_fc_weights* w, const uic_topdown_batch* b, int32_t L, int32_t beam_size,
                       int32_t decoding_constraint, int32_t max_ppl, void* workspace, int64_t* seq, float* seq_logp, void* stream);

/* ---- pivot NMT step: NMTModel.forward + generator + NMTCriterion and their backward (P/models/NMT_Models.py:27-295,
 * 414-420; O = misc/OpenNMT-py-dalegebit/onmt: O/modules/StackedRNN.py:20-34, O/modules/GlobalAttention.py:112-167;
 * P/misc/criterion.py:126-136,161-205).  Supported configuration = the reference's defaults: LSTM, brnn, input_feed,
 * dotprod attention with softmax, no context gate / coverage / copy attention. ---- */
#define UIC_NMT_MAX_LAYERS 4
typedef struct uic_nmt_dims {
  int32_t B;        /* batch (sentences) */
  int32_t S;        /* padded source length */
  int32_t T;        /* padded target length incl. BOS/EOS; the decoder runs T-1 steps on tgt[:-1] */
  int32_t H;        /* rnn_size (encoder direction size = H/2) */
  int32_t W;        /* word_vec_size */
  int32_t layers;
  int32_t Vs, Vt;   /* source / target dictionary sizes */
  int32_t dtype;
  float drop_p;     /* opt.dropout */
  /* how the decoder's target-step loop is launched (as uic_topdown_dims.recurrence / .rnn_status): 0 = ONE persistent launch
   * (csrc/nmt_persist.hip) where the shapes allow -- bf16, rnn_size 512, source length <= 64 --, UIC_REC_FWD_CHAIN = layers + 2
   * launches per step, UIC_REC_SAFE = the placement-independent exchange protocol; rnn_status: 4 caller-zeroed uint32 status
   * words on the device (word 0 != 0: a bounded spin gave up, the outputs are invalid) or NULL */
  int32_t recurrence;
  uint32_t* rnn_status;
  /* Optional list of the target positions that are not PAD, for uic_nmt_forward_loss / uic_nmt_backward (NULL: every position is
   * computed; NMTCriterion's weight[PAD] = 0 makes the padded ones exact zeros, P/misc/criterion.py:126-136 -- the same result).
   * Position t * B + b (decoder step t, sentence b) is live when tgt[t + 1, b] != PAD (0).  With the list the generator, the
   * criterion, d outputs and the generator's weight gradient run over the listed rows only (40 % of the positions are padding
   * in a batch of target lengths ~U{7..30}).  tgt_live_count > 0: the number of non-PAD positions of tgt[1:] (the caller counts
   * them where it assembles the batch -- on the host, as the reference's onmt.Dataset does); tgt_live_rows: DEVICE int32
   * [roundup(tgt_live_count, 128)], any order, the tail up to the multiple of 128 holds -1, or NULL: the forward call compacts
   * the targets itself (one small launch) and the backward call reads that list from the workspace.  The same values must be
   * passed to the forward and the backward call of a step.  A list that leaves out a non-PAD position drops its loss and
   * gradient.  tgt_live_count == 0: every position is computed. */
  const int32_t* tgt_live_rows;
  int32_t tgt_live_count;
} uic_nmt_dims;

typedef struct uic_nmt_weights {     /* keys of NMTModel.state_dict() + generator (P/trainer.py:85-89) */
  float* enc_lut;                                   /* encoder.embeddings.word_lut.weight   [Vs, W]   */
  float* enc_lin_w;                                 /* encoder.embeddings.linear.weight     [W, W]    */
  float* enc_lin_b;                                 /* encoder.embeddings.linear.bias       [W]       */
  float* enc_w_ih[UIC_NMT_MAX_LAYERS][2];           /* encoder.rnn.weight_ih_l{k}[_reverse] [4H/2, in]*/
  float* enc_w_hh[UIC_NMT_MAX_LAYERS][2];           /* encoder.rnn.weight_hh_l{k}[_reverse] [4H/2, H/2] */
  float* enc_b_ih[UIC_NMT_MAX_LAYERS][2];
  float* enc_b_hh[UIC_NMT_MAX_LAYERS][2];
  float* dec_lut;                                   /* decoder.embeddings.word_lut.weight   [Vt, W]   */
  float* dec_w_ih[UIC_NMT_MAX_LAYERS];              /* decoder.rnn.layers.{k}.weight_ih     [4H, W+H | H] */
  float* dec_w_hh[UIC_NMT_MAX_LAYERS];              /* decoder.rnn.layers.{k}.weight_hh     [4H, H]   */
  float* dec_b_ih[UIC_NMT_MAX_LAYERS];
  float* dec_b_hh[UIC_NMT_MAX_LAYERS];
  float* attn_in_w;                                 /* decoder.attn.linear_in.weight        [H, H]    */
  float* attn_out_w;                                /* decoder.attn.linear_out.weight       [H, 2H]   */
  float* gen_w;                                     /* generator.0.weight                   [Vt, H]   */
  float* gen_b;                                     /* generator.0.bias                     [Vt]      */
} uic_nmt_weights;

size_t uic_nmt_workspace_bytes(const uic_nmt_dims* d);
/* Named pieces of a workspace, for tools (as uic_topdown_workspace_ptr): "dec_fwd_dbg" / "dec_bwd_dbg" = the persistent decoder
 * forward / BPTT launches' per-phase time stamps when dims.recurrence has UIC_REC_STAMPS ([256 workgroups][T-1][16] uint64,
 * 100 MHz; tools/nmt_bwd_probe.py prints them), "d_cq", "dscore", "d_pre".  NULL for an unknown name. */
void* uic_nmt_workspace_ptr(const uic_nmt_dims* d, void* workspace, const char* name);
/* Forward + loss: src [S,B] int64 (PAD = 0), lengths sorted descending (host AND device copies: the reference moves
 * them to the host for pack_padded_sequence too), tgt [T,B] int64.  Outputs (all optional except loss): loss_out[0] =
 * sum of NLL over non-PAD targets; stats_out = {num_correct, num_words} (int32, device); outputs_out [T-1,B,H] and
 * attn_out [T-1,B,S] f32 as returned by NMTModel.forward; context_out [S,B,H] f32. */
int uic_nmt_forward_loss(const uic_nmt_dims* d, const uic_nmt_weights* w, const int64_t* src, const int32_t* lengths_host,
                         const int32_t* lengths_dev, const int64_t* tgt, int32_t training, uint32_t seed, void* workspace,
                         float* loss_out, int32_t* stats_out, float* outputs_out, float* attn_out, float* context_out,
                         void* stream);
/* Backward of the sum-NLL loss left in the workspace by uic_nmt_forward_loss; every tensor of `grads` is overwritten. */
int uic_nmt_backward(const uic_nmt_dims* d, const uic_nmt_weights* w, const int64_t* src, const int32_t* lengths_host,
                     const int64_t* tgt, int32_t training, uint32_t seed, void* workspace, const uic_nmt_weights* grads,
                     void* stream);

/* Data-parallel callers (replaces DataParallel(nmt_model, dim=1)'s reduce-add, P/trainer.py:88-89): lets `stream` wait until a
 * gradient group of the last uic_nmt_backward on this device is final, so that its exchange runs beside the rest of the
 * backward pass.  group 0: generator.{weight,bias} (30 % of the bytes at the reference's sizes; final before the decoder BPTT
 * starts); group 1: the decoder side (decoder.rnn.*, decoder.embeddings.*, decoder.attn.*; final before the encoder's backward
 * pass).  The encoder's gradients are final when uic_nmt_backward's own stream is through.  The clipped Adam step needs the
 * norm of the SUMMED gradient: uic_grad_sqnorm after all three pieces. */
int uic_nmt_grad_ready_wait(void* stream, int32_t group);

/* NMTModel.translateBatch (P/models/NMT_Models.py:322-395) with the fork's Beam (O/Beam.py): beam search translation of a
 * batch, n_best = 1.  src [S, B] int64 (PAD = 0; the encoder runs without lengths, as the reference's does here).  Outputs:
 * hyp_out [B, max_steps] int64 (first *n_iter_out columns valid: the number of decoder steps taken, identical for all
 * sentences -- a sentence's beam keeps advancing until every sentence is done, :372-378), score_out [B] (best final beam
 * score), attn_out [B, max_steps, S] or NULL (attention of the winning hypothesis, PAD source columns dropped and the rest
 * packed to the left), n_iter_out (HOST int).  The reference's `if not active: break` is decided on the device (the search
 * freezes itself once no sentence is active and counts the steps that ran); unlike the other entry points this one
 * synchronises with the host -- every fourth decoder step, to stop enqueueing, and once at the end for n_iter_out.  The
 * reference hard-codes beam_size 15 and max_steps 100. */
size_t uic_nmt_translate_workspace_bytes(const uic_nmt_dims* d, int32_t beam_size, int32_t max_steps);
int uic_nmt_translate(const uic_nmt_dims* d, const uic_nmt_weights* w, const int64_t* src, int32_t beam_size, int32_t max_steps,
                      void* workspace, int64_t* hyp_out, float* score_out, float* attn_out, int32_t* n_iter_out, void* stream);

/* ---- data-parallel gradient exchange on RCCL (replaces DataParallel's reduce-add, P/trainer.py:74,88-89), for callers
 * that do not use torch.distributed.  librccl is dlopen()ed on first use.  uic_comm_unique_id: rank 0 obtains the 128-byte
 * rendezvous id and hands it to every rank out of band (file, socket, MPI, a torch store); uic_comm_init: collective over all
 * `world` ranks, each on ITS current device (one process per GPU); uic_comm_allreduce: in-place sum of `count` elements
 * (UIC_F32 or UIC_BF16) on `stream`, enqueue only -- e.g. the flat gradient arena after uic_topdown_grad_ready_wait.
 * Errors: 1000 + ncclResult_t. ---- */
#define UIC_COMM_ID_BYTES 128
int uic_comm_unique_id(void* id_out);
int uic_comm_init(int32_t rank, int32_t world, const void* id, void** comm_out);
int uic_comm_allreduce(void* comm, void* buf, size_t count, int32_t dtype, void* stream);
int uic_comm_destroy(void* comm);
/* The two halves of the sharded exchange: uic_comm_reduce_scatter sums `world * recvcount` elements at
 * sendbuf over the ranks and leaves elements [rank * recvcount, (rank + 1) * recvcount) of the sum at recvbuf (in place when
 * recvbuf == sendbuf + rank * recvcount); uic_comm_allgather sends `sendcount` elements from every rank and leaves the ranks'
 * blocks in rank order at recvbuf (in place when sendbuf == recvbuf + rank * sendcount).  uic_comm_group_start / _end bracket
 * several of them so that RCCL launches them as one kernel (ncclGroupStart / ncclGroupEnd). */
int uic_comm_reduce_scatter(void* comm, const void* sendbuf, void* recvbuf, size_t recvcount, int32_t dtype, void* stream);
int uic_comm_allgather(void* comm, const void* sendbuf, void* recvbuf, size_t sendcount, int32_t dtype, void* stream);
int uic_comm_group_start(void);
int uic_comm_group_end(void);
/* Measurement aids (tools/comm_proxy.py), NOT collectives: stand-ins on a single GPU.  uic_comm_proxy: one all-reduce of `bytes`
 * bytes (a multiple of 16) -- `workgroups` 256-thread workgroups stream buf -> scratch -> buf on `stream`; buf is unchanged
 * afterwards.  uic_comm_proxy_oneway: a reduce-scatter or an all-gather of `bytes` bytes -- one pass, src -> dst. */
int uic_comm_proxy(void* buf, void* scratch, size_t bytes, int32_t workgroups, void* stream);
int uic_comm_proxy_oneway(const void* src, void* dst, size_t bytes, int32_t workgroups, void* stream);

/* ---- single operators (also used by the parity tests) ---- */

/* nn.Linear as C[M,N] = A[M,K] B[N,K]^T (+bias)(+ReLU); flags: 1 ReLU, 2 accumulate into C, 4 C is f32. */
int uic_linear(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* B, int32_t ldb,
               void* C, int32_t ldc, const float* bias, int32_t flags, void* stream);

/* nn.Linear on an f32 input with bf16 weights (att_embed on the loader's f32 region features, P/models/AttModel.py:76-80 as
 * prepared by :111-115): A[M,K] f32 is rounded to bf16 inside the GEMM -- bit for bit what uic_cast_from_f32 followed by uic_linear
 * gives -- and, with a_bf16 != NULL, that bf16 image [M, ld_a_bf16] is stored as well (the weight gradient's operand), so the
 * separate cast pass over the features disappears.  B[N,K] bf16, C bf16 (or f32 with flag 4), flags as uic_linear.
 * Needs K % 128 == 0, N % 4 == 0, lda % 4 == 0, ld_a_bf16 % 8 == 0, 16-byte aligned operands, M * lda * 4 < 4 GB. */
int uic_linear_f32a(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const void* B, int32_t ldb,
                    void* C, int32_t ldc, const float* bias, int32_t flags, void* a_bf16, int32_t ld_a_bf16, void* stream);

/* The same product split over K: slice z of `splitk` (each a whole number of 128-byte K rounds) leaves its raw partial tile in
 * slab[z][M][N] (f32, dense); the caller sums the slices -- what the BPTT loop's d x GEMMs do, whose consumers (cell backward,
 * attention backward) add the slices while they read them.  One K segment, K % 64 == 0 (bf16) / % 32 (f32). */
int uic_linear_partials(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* B, int32_t ldb,
                        float* slab, int32_t splitk, void* stream);

/* Weight gradient of nn.Linear without transposed copies: dW[M,N] (f32) = dY[K,M]^T X[K,N] (+= if accumulate), dY / X bf16
 * row-major with the reduction index (caption rows / decode steps) as the ROW index, exactly as the backward pass holds
 * them.  gfx950 transposing LDS reads (csrc/gemm_tn.hip).  Needs M >= 128, M % 8 == 0, N % 128 == 0, K % 64 == 0;
 * workspace: at least 4*M*N bytes (more lets it split K over workgroups: deterministic f32 partial slabs, summed in slice order).
 * accumulate: bit 0 = add to dW; for measurements and tests bits 8-9 pick the kernel (0x100: 128 x 128 tiles, csrc/gemm_tn.hip;
 * 0x200: 256 x 256 ping-pong tiles, csrc/gemm_tn_pp.hip, needs K % 128 == 0) and bits 16-23 the number of K slices (0: the
 * library's choice). */
int uic_linear_wgrad(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* dY, int32_t ldy, const void* X, int32_t ldx,
                     float* dW, int32_t ldw, void* workspace, size_t workspace_bytes, int32_t accumulate, void* stream);

/* nn.LSTMCell (P/models/AttModel.py:426-427,434,441) on the concatenation of up to three inputs:
 * gates = sum_i x_i W_i^T + h W_hh^T + b_ih + b_hh.  x_i [M,K_i] (operand dtype), W_i = weight_ih column block
 * (ldw = row stride of weight_ih).  Outputs: c_out f32 [M,H], h_out operand dtype [M,H], gates_out (optional). */
int uic_lstm_cell_fwd(int32_t dtype, int32_t M, int32_t H, int32_t nx, const void* const* x, const int32_t* Kx,
                      const void* const* Wx, const int32_t* ldw, const void* h, const void* W_hh,
                      const float* b_ih, const float* b_hh, const float* c_prev, float* c_out, void* h_out,
                      void* gates_out, void* stream);
/* Pointwise backward of the cell: dh [M,H] f32, dc [M,H] f32 (in: from the future, out: for the past),
 * gates (activated), c_prev, c  ->  dgates [M,4H] operand dtype. */
int uic_lstm_cell_bwd(int32_t dtype, int32_t M, int32_t H, const float* dh, float* dc, const void* gates,
                      const float* c_prev, const float* c, void* dgates, void* stream);

/* Attention.forward after h2att (P/models/AttModel.py:544-556). */
int uic_attention_fwd(int32_t dtype, int32_t N, int32_t R, int32_t A, int32_t H, const float* att_h, const void* p_att,
                      const void* att, const float* w_alpha, const float* b_alpha, const float* mask, float* alpha,
                      void* ctx, void* stream);
int uic_attention_bwd_step(int32_t dtype, int32_t N, int32_t R, int32_t A, int32_t H, const float* att_h,
                           const void* p_att, const void* att, const float* w_alpha, const float* alpha,
                           const float* dctx, float* de, void* d_att_h, void* stream);
int uic_attention_bwd_accum(int32_t dtype, int32_t N, int32_t R, int32_t A, int32_t H, int32_t T,
                            const float* att_h_all, const float* alpha_all, const float* de_all, const float* dctx_all,
                            const void* p_att, const float* w_alpha, float* d_att, void* d_p_att, float* d_walpha_part,
                            void* stream);

/* torch.optim.Adam step (P/misc/optimizer.py:70,93) on one flat f32 arena; `step` is 1-based. */
int uic_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                  float eps, int32_t step, float grad_scale, void* stream);
/* The same step, skipped ON THE DEVICE when skip_if_nonzero[0] has any bit set (NULL: never skipped): the update of a training
 * step whose persistent recurrence launch timed out (uic_topdown_dims.rnn_status[0] != 0, its gradients are invalid) must not
 * reach the weights or the Adam moments.  The word may be rnn_status itself or, in a data-parallel run, the all-reduced sum of
 * the ranks' words as a float (every rank then skips together; replaces nothing in the reference -- P/trainer.py:173 has no
 * persistent kernels to time out). */
int uic_adam_step_guarded(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                          float eps, int32_t step, float grad_scale, const int32_t* skip_if_nonzero, void* stream);

/* The same step (optionally clipped, optionally guarded -- max_norm = 0 / sqnorm = NULL / skip_if_nonzero = NULL switch those off)
 * on `n_ranges` (at most 24) index ranges [lo[i], hi[i]) of the arena in ONE launch: the ranges a data-parallel rank owns after the
 * reduce-scatter of the gradient pieces, plus the replicated tail.  lo / hi are HOST arrays.  w_out (optional): the updated
 * parameters of those ranges are also written there in w_dtype (UIC_DTYPE_BF16; element index = arena index, so w_out must span
 * every listed range -- in practice the whole arena's length) -- the rank's contribution to the all-gather of the operand-dtype weights.  Element for element the arithmetic of uic_adam_step. */
int uic_adam_step_ranges(float* p, const float* g, float* m, float* v, int32_t n_ranges, const uint64_t* lo, const uint64_t* hi,
                         float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale, float max_norm,
                         const float* sqnorm, const int32_t* skip_if_nonzero, void* w_out, int32_t w_dtype, void* stream);

/* torch.nn.utils.clip_grad_norm + Adam as Optim.step applies them to the NMT model (P/misc/optimizer.py:93-100,
 * --nmt_max_grad_norm 5): uic_grad_sqnorm leaves sum(g^2) of the flat gradient arena in out[0] (deterministic
 * two-stage sum; scratch >= 1024 floats); uic_adam_step_clip scales the gradient by grad_scale and, if
 * max_norm / (|grad_scale| * sqrt(sqnorm[0]) + 1e-6) < 1, by that coefficient too, read on the device. */
int uic_grad_sqnorm(const float* g, size_t n, float* scratch, float* out, void* stream);
int uic_adam_step_clip(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                       float eps, int32_t step, float grad_scale, float max_norm, const float* sqnorm, void* stream);
/* ... skipped on the device when skip_if_nonzero[0] has any bit set (as uic_adam_step_guarded: the pivot NMT step's persistent
 * launches, uic_nmt_dims.rnn_status) */
int uic_adam_step_clip_guarded(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                               float eps, int32_t step, float grad_scale, float max_norm, const float* sqnorm,
                               const int32_t* skip_if_nonzero, void* stream);

/* LanguageModelCriterion on materialised log-probs [N, T, V1] (API-compatible path): loss_out[0] and,
 * if dlogp != NULL, the dense gradient scaled by grad_out (host scalar). */
int uic_lm_criterion(int32_t N, int32_t T, int32_t V1, const float* logp, const int64_t* target, int32_t ld_target,
                     const float* mask, int32_t ld_mask, float* loss_out, float* scratch, float* dlogp, float grad_out,
                     void* stream);

/* RewardCriterion (P/misc/criterion.py:117-124): loss_out[0] = -sum(logp * reward * m) / sum(m), m = (seq > 0) shifted
 * right by one with a leading 1; if dlogp != NULL it receives d loss / d logp [N, L]. */
int uic_reward_criterion(int32_t N, int32_t L, const float* logp, const int64_t* seq, const float* reward, float* loss_out,
                         float* dlogp, void* stream);

/* utilities */
int uic_cast_from_f32(int32_t dtype, const float* src, void* dst, size_t n, void* stream);
int uic_cast_to_f32(int32_t dtype, const void* src, float* dst, size_t n, void* stream);
int uic_transpose(int32_t dtype, const void* src, int32_t rows, int32_t cols, int32_t ld_src, void* dst, int32_t ld_dst, void* stream);
/* the multiplicative dropout mask (0 or 1/(1-p)) the kernels use at `site` for elements base..base+n-1 */
int uic_dropout_mask(float* out, size_t n, float p, uint32_t seed, uint32_t site, size_t base, void* stream);
#define UIC_SITE_FC 1u
#define UIC_SITE_ATT 2u
#define UIC_SITE_EMBED 3u
#define UIC_SITE_OUT0 16u   /* + decode step */
#define UIC_SITE_LOGIT_H0 8u   /* + l: Dropout(0.5) of hidden logit block l (logit_layers > 1), element (t*N + n)*H + j */
#define UIC_SITE_SS_MASK0 512u  /* + decode step: row n is re-sampled iff u(seed, site, n) < ss_prob                  */
#define UIC_SITE_SS_DRAW0 768u  /* + decode step: u(seed, site, n) drives the inverse-CDF draw of row n              */
#define UIC_SITE_NMT_ENC0 1000u  /* + l: nn.LSTM dropout after encoder layer l, element (s*B + b)*H + j           */
#define UIC_SITE_NMT_DEC0 2000u  /* + l*256 + t: StackedLSTM dropout after decoder layer l at step t, element b*H+j */
#define UIC_SITE_NMT_OUT0 4000u  /* + t: Decoder.dropout on the attentional output of step t, element b*H + j      */

#define UIC_SITE_DISC 6000u      /* Dropout on the sentence discriminator's highway output, element n*Ft + a             */

/* ---- CNN sentence discriminator (BASELINE configs[3]; north_star "the sentence-discriminator forward/backward").
 * THE REFERENCE TREE HOLDS NO DISCRIMINATOR CODE: this is the package's own statement of the usual text-CNN critic and its
 * parity is UNPINNED (csrc/discriminator.hip states the architecture; oracle/discriminator.py restates it on the CPU).
 *   x = relu(Emb[tok]) [N, L, E];  y_w = relu(conv1d_w(x) + b_w), right zero padding;  p = concat_w max_t y_w  [N, nw*F];
 *   g, h = sigmoid / relu of hw_w p + hw_b (hw_w = [W_gate; W_transform], [2 Ft, Ft]);  z = g h + (1 - g) p;
 *   logit = out_w . dropout(z) + out_b;  D = sigmoid(logit).
 * conv_w[i] is [F, widths[i], E] row-major (= nn.Conv1d weight [F, E, w] with the last two axes swapped).  Rows are the
 * captioner's int64 token rows [N, ld_tokens], first L columns.  Gradient struct = weight struct, f32, overwritten. ---- */
#define UIC_DISC_MAX_WIDTHS 4
#define UIC_DISC_MAX_WIDTH 4
typedef struct {
  int32_t dtype, N, L, V1, E, F, nw;
  int32_t widths[UIC_DISC_MAX_WIDTHS];
  float drop_p;
} uic_disc_dims;
typedef struct {
  float* embed_w;                        /* [V1, E] */
  float* conv_w[UIC_DISC_MAX_WIDTHS];    /* [F, w, E] */
  float* conv_b[UIC_DISC_MAX_WIDTHS];    /* [F] */
  float* hw_w; float* hw_b;              /* [2 Ft, Ft], [2 Ft] */
  float* out_w; float* out_b;            /* [Ft], [1] */
} uic_disc_weights;
size_t uic_disc_workspace_bytes(const uic_disc_dims* d);
/* forward: logits_out / prob_out (optional, [N] f32).  training != 0: dropout with `seed`, and the workspace keeps what
 * uic_disc_backward needs. */
int uic_disc_forward(const uic_disc_dims* d, const uic_disc_weights* w, const int64_t* tokens, int32_t ld_tokens, int32_t training,
                     uint32_t seed, void* workspace, float* logits_out, float* prob_out, void* stream);
/* BCEWithLogitsLoss (mean): loss_out[0]; dlogits_out (optional) = (sigmoid(logit) - label) / N */
int uic_disc_bce(const float* logits, const float* labels, int32_t N, float* loss_out, float* dlogits_out, void* stream);
/* backward of the forward that last ran on `workspace` (same tokens, training, seed): every gradient of G from dlogits [N] */
int uic_disc_backward(const uic_disc_dims* d, const uic_disc_weights* w, const int64_t* tokens, int32_t ld_tokens, int32_t training,
                      uint32_t seed, void* workspace, const float* dlogits, const uic_disc_weights* G, void* stream);

/* ---- Scene-graph GCN encoder (BASELINE configs[4]: "GCN over 36 objects + relations -> attention-LSTM decoder").  NO REFERENCE
 * CODE EXISTS: like the discriminator this is the package's own statement of the usual graph convolution, parity UNPINNED
 * (csrc/gcn.hip, oracle/gcn.py).  X_0 = x [N, R, D];  X_{l+1} = relu(A_hat (X_l W_l^T) + b_l), W_l [H, D_l] (nn.Linear
 * layout), l < layers <= UIC_GCN_MAX_LAYERS;  adj = A_hat [N, R, R] f32, the caller's normalised relation graph (data, no
 * gradient; R <= 64).  out [N, R, H] f32 = the att_feats the captioner takes (att_feat_size = H). ---- */
#define UIC_GCN_MAX_LAYERS 3
typedef struct { int32_t dtype, N, R, D, H, layers; } uic_gcn_dims;
typedef struct { float* w[UIC_GCN_MAX_LAYERS]; float* b[UIC_GCN_MAX_LAYERS]; } uic_gcn_weights;
size_t uic_gcn_workspace_bytes(const uic_gcn_dims* d);
int uic_gcn_forward(const uic_gcn_dims* d, const uic_gcn_weights* w, const float* x, const float* adj, void* workspace, float* out,
                    void* stream);
/* backward of the forward that last ran on `workspace`: G (f32, overwritten) from dout [N, R, H]; dx (optional) [N, R, D] */
int uic_gcn_backward(const uic_gcn_dims* d, const uic_gcn_weights* w, const float* adj, void* workspace, const float* dout,
                     const uic_gcn_weights* G, float* dx, void* stream);

/* ---- CIDEr-D reward of the self-critical step ----
 * get_self_critical_reward (P/misc/rewards.py:37-81) over the CIDEr-D scorer
 * (P/misc/cider/pyciderevalcap/ciderD/ciderD_scorer.py:116-209, ciderD.py:26-50) on integer token rows, so that the sampled
 * and greedy captions stay on the device.  A caption's words are its tokens up to and including the first 0
 * (array_to_str, rewards.py:29-35).
 *
 * Document frequencies live in an open-addressing hash table keyed by the n-gram's tokens:
 *   uic_ciderd_table_slots(n)  capacity (power of two) for n n-grams;
 *   uic_ciderd_table_build     HOST arrays in, HOST arrays out (the caller uploads slot_keys [slots, 4] int32 and
 *                              slot_vals [slots] f64): keys [n, 4] int32 (unused positions -1), values [n] =
 *                              log(max(1, df)) as the scorer's counts2vec uses it (:128). */
int64_t uic_ciderd_table_slots(int64_t n_entries);
int uic_ciderd_table_build(const int32_t* keys, const double* values, int64_t n, int32_t* slot_keys, double* slot_vals,
                           int64_t slots);
/* CiderD.compute_score for n_hyp hypotheses hyp [n_hyp, L] (int64): hypothesis h is scored against the references of
 * image (h % batch_size) / seq_per_img (rewards.py:59); references of image i are rows ref_start[i] .. ref_start[i+1] - 1
 * of ref_tok [*, Lr] (int64, the zero-padded label rows of data['gts']).  ref_len as the scorer holds it (the value stored
 * in the cached-tokens pickle, or log(n_hyp) in 'corpus' mode); penalty[d + pen_half] = e^(-d^2 / (2 sigma^2)) for
 * d = -pen_half .. pen_half (pen_half >= max(L, Lr)), made on the host.  slots = 0: no table, every df is 0.
 * scores [n_hyp] f64 = 10 * mean over n-gram orders and references, as :181-197.  L, Lr <= 64. */
int uic_ciderd_scores(const int64_t* hyp, int32_t n_hyp, int32_t L, int32_t batch_size, int32_t seq_per_img,
                      const int64_t* ref_tok, int32_t Lr, const int32_t* ref_start, int32_t n_img,
                      const int32_t* slot_keys, const double* slot_vals, int64_t slots, double ref_len,
                      const double* penalty, int32_t pen_half, double* scores, void* stream);
/* Per-sentence BLEU-4 of the same hypotheses against the same references (the `bleu_reward_weight` term of
 * get_self_critical_reward, rewards.py:70-75: Bleu(4).compute_score -> bleu_scores[3]); BleuScorer.compute_score with
 * option='closest' (coco_caption/pycxevalcap/bleu/bleu_scorer.py:199-240): clipped n-gram matches, the reference length
 * closest to the hypothesis (ties: the shorter), (correct + 1e-15) / (guess + 1e-9) products, ^(1/4), brevity penalty
 * exp(1 - 1/ratio).  scores [n_hyp] f64 (pow / exp of the device math library: within 1e-14 relative of the host's). */
int uic_bleu_scores(const int64_t* hyp, int32_t n_hyp, int32_t L, int32_t batch_size, int32_t seq_per_img,
                    const int64_t* ref_tok, int32_t Lr, const int32_t* ref_start, int32_t n_img, double* scores, void* stream);
/* reward [N, L] f32 = weight * (scores[n] - scores[N + n]) repeated over the L positions (rewards.py:74-79): rows 0..N-1 of
 * `scores` are the sampled captions, rows N..2N-1 the greedy baseline. */
int uic_ciderd_reward(const double* scores, int32_t N, int32_t L, float weight, float* reward, void* stream);

/* ---- input pipeline: batch assembly on the device --------------------------------------------------------------------
 * Replaces the per-image numpy work of DataLoader.__getitem__ (P/misc/dataloader/dataloader.py:302-331) and the padding /
 * masking of DataLoader.get_batch (:277-283).  The host packs the RAW per-image files back to back, once per image (no
 * seq_per_img replication -- that is uic_topdown_dims.seq_per_img):
 *   feat_pack    [n_regions, D] f32: the `feat` arrays of the att .npz files, images in loader order;
 *   box_pack     [n_regions, 4] f32 (x1, y1, x2, y2) of the box .npy files, or NULL for use_box = 0;
 *   region_start [n_img + 1] i32: image i owns regions region_start[i] .. region_start[i+1] - 1;
 *   img_hw       [n_img, 3] f32 (height, width, width * height) from the info json (needed with box_pack);
 *   img_slot     [n_img] i32: output position of image i (get_batch's stable sort by region count, descending, :264-265).
 * Output att_feats [n_img, Rmax, ld_out] f32 (ld_out >= D + 5 with boxes; columns past the data are zero-filled so a row
 * padded for the captioner's GEMMs needs no second pass) and att_masks [n_img, Rmax] f32.  Per region: x / ||x||_2 when
 * norm_att_feat (:310-311), box features (x1/w, y1/h, x2/w, y2/h, area/(wh)) [/ their L2 norm when norm_box_feat] appended
 * (:318-325), regions sorted by the last column, descending, stable (:327).  Every f32 operation is done in numpy's order
 * (pairwise summation included): the batch is bit-identical to the reference's.  D <= 16384, Rmax <= 2048. */
int uic_att_batch_assemble(const float* feat_pack, const float* box_pack, const int32_t* region_start, const float* img_hw,
                           const int32_t* img_slot, int32_t n_img, int32_t D, int32_t norm_att_feat, int32_t norm_box_feat,
                           int32_t Rmax, int32_t ld_out, float* att_feats, float* att_masks, void* stream);
/* Host side of the input pipeline (no device work): the per-image feature files read by a thread team straight into
 * the caller's (pinned) staging buffer -- replaces np.load in the reference's DataLoader worker processes
 * (P/misc/dataloader/dataloader.py:309,319,331,351-356).  Files: .npy, or .npz whose FIRST member is `<member>.npy`, stored
 * or deflated (np.savez / np.savez_compressed, scripts/make_bu_data.py:55-57); float32, C order, 1 or 2 dimensions.
 * uic_loader_scan: info [n, 6] i64 per file = (ndim, d0, d1, offset of the data inside the member, zip method or -1,
 * offset of the member in the file).  uic_loader_read: the d0 * d1 floats of file i are written at dst[i]; `info` as
 * returned by the scan.  `member` may be NULL for plain .npy files.  A file that cannot be read, or is not float32, fails the
 * whole call (UIC_EARG, uic_last_error_string() names the file).  Thread-safe; n_threads <= 128. */
int uic_loader_scan(const char* const* paths, int32_t n, const char* member, int64_t* info, int32_t n_threads);
int uic_loader_read(const char* const* paths, int32_t n, const int64_t* info, void* const* dst, int32_t n_threads);
/* The decoder uic_loader_read uses for deflated members, on one raw deflate stream (RFC 1951; what a zip member of
 * np.savez_compressed holds): src[0..n) -> exactly m bytes at dst.  fast != 0: the library's own table-driven decoder
 * (csrc/inflate_fast.h: 64-bit bit buffer, one 11-bit lookup per symbol, up to three literals per refill), returns 1 when it
 * declines the stream (malformed, or not exactly m bytes long) -- uic_loader_read then retries with zlib; fast == 0: zlib.  Host
 * only; a test hook (tests/test_dataloader_host.py compares the two decoders on stored, fixed and dynamic blocks).
 * The worker threads of the reader team are pinned to CPUs of their own unless UIC_LOADER_NO_PIN=1 is in the environment;
 * UIC_LOADER_ZLIB=1 keeps uic_loader_read on zlib (both read once, at first use: the library's only environment reads, host side). */
int uic_loader_inflate(const void* src, size_t n, void* dst, size_t m, int32_t fast);
/* The same decoder on TWO streams in lock-step on the calling thread -- how uic_loader_read takes deflated members (a deflate
 * stream is one chain of dependent table lookups; two chains share a core's issue slots).  Returns a bit mask: bit k set = stream k
 * declined. */
int uic_loader_inflate_pair(const void* src0, size_t n0, void* dst0, size_t m0, const void* src1, size_t n1, void* dst1, size_t m1);

#ifdef __cplusplus
}
#endif
#endif /* UIC_HIP_H */
