"""ctypes binding of libuic_hip.so (include/uic_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the HIP library is missing,
or a call fails, this module raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuic_hip.so")

F32, BF16 = 0, 1
DTYPE_IDS = {"f32": F32, "fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16}
TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16}

SITE_FC, SITE_ATT, SITE_EMBED, SITE_OUT0 = 1, 2, 3, 16
SITE_SS_MASK0, SITE_SS_DRAW0 = 512, 768          # + decode step


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "R", "D", "Dfc", "H", "E", "A", "V1", "T", "dtype")] + [("drop_p", C.c_float), ("use_bn", C.c_int32), ("seq_per_img", C.c_int32), ("logit_layers", C.c_int32),
                                                                                                  ("recurrence", C.c_int32), ("rnn_status", C.c_void_p)]


# uic_topdown_dims.recurrence (include/uic_hip.h)
REC_FWD_CHAIN, REC_BWD_PERSIST, REC_SAFE, REC_STAMPS, REC_EARLY_GRADS, REC_NO_F32A, REC_COMM_STREAM = 1, 2, 4, 8, 16, 32, 64
STEP_MARKS = 11          # UIC_STEP_MARKS


MAX_LOGIT_LAYERS = 4
SITE_LOGIT_H0 = 8       # + hidden logit block


WEIGHT_FIELDS = [
    # (struct field, reference state_dict key)
    ("embed_w", "embed.0.weight"),
    ("fc_w", "fc_embed.0.weight"),
    ("fc_b", "fc_embed.0.bias"),
    ("att_w", "att_embed.0.weight"),
    ("att_b", "att_embed.0.bias"),
    ("logit_w", "logit.weight"),
    ("logit_b", "logit.bias"),
    ("ctx2att_w", "ctx2att.weight"),
    ("ctx2att_b", "ctx2att.bias"),
    ("att_lstm_w_ih", "core.att_lstm.weight_ih"),
    ("att_lstm_w_hh", "core.att_lstm.weight_hh"),
    ("att_lstm_b_ih", "core.att_lstm.bias_ih"),
    ("att_lstm_b_hh", "core.att_lstm.bias_hh"),
    ("lang_lstm_w_ih", "core.lang_lstm.weight_ih"),
    ("lang_lstm_w_hh", "core.lang_lstm.weight_hh"),
    ("lang_lstm_b_ih", "core.lang_lstm.bias_ih"),
    ("lang_lstm_b_hh", "core.lang_lstm.bias_hh"),
    ("h2att_w", "core.attention.h2att.weight"),
    ("h2att_b", "core.attention.h2att.bias"),
    ("alpha_w", "core.attention.alpha_net.weight"),
    ("alpha_b", "core.attention.alpha_net.bias"),
]


# BatchNorm1d tensors of att_embed with opt.use_bn >= 1 (att_embed.0) / == 2 (att_embed.4); (field, key, is_parameter)
BN_FIELDS = [
    ("att_bn0_w", "att_embed.0.weight", True), ("att_bn0_b", "att_embed.0.bias", True),
    ("att_bn0_rm", "att_embed.0.running_mean", False), ("att_bn0_rv", "att_embed.0.running_var", False),
    ("att_bn4_w", "att_embed.4.weight", True), ("att_bn4_b", "att_embed.4.bias", True),
    ("att_bn4_rm", "att_embed.4.running_mean", False), ("att_bn4_rv", "att_embed.4.running_var", False),
]


def weight_fields(use_bn=0, logit_layers=1):
    """[(struct field, reference state_dict key, is_parameter)] for opt.use_bn / opt.logit_layers, parameters in
    named_parameters() order.  With use_bn >= 1 the Linear of att_embed moves to index 1 (P/models/AttModel.py:78-84); with
    logit_layers = n > 1 `logit` is a Sequential of n - 1 blocks [Linear(H, H), ReLU, Dropout] and the vocabulary layer, so
    hidden block l is `logit.{3l}` and the final Linear `logit.{3(n-1)}` (:90-91).  Struct fields "logit_h_w:l" / "logit_h_b:l"
    index the pointer arrays of uic_topdown_weights."""
    out = []
    for f, k in WEIGHT_FIELDS:
        if f == "att_w" and use_bn:
            out += [b for b in BN_FIELDS[:4]]
        if f in ("att_w", "att_b") and use_bn:
            k = k.replace("att_embed.0.", "att_embed.1.")
        if f == "logit_w" and logit_layers > 1:
            for l in range(logit_layers - 1):
                out.append(("logit_h_w:%d" % l, "logit.%d.weight" % (3 * l), True))
                out.append(("logit_h_b:%d" % l, "logit.%d.bias" % (3 * l), True))
        if f in ("logit_w", "logit_b") and logit_layers > 1:
            k = k.replace("logit.", "logit.%d." % (3 * (logit_layers - 1)))
        out.append((f, k, True))
        if f == "att_b" and use_bn == 2:
            out += [b for b in BN_FIELDS[4:]]
    return out


class Weights(C.Structure):
    _fields_ = [(f, C.c_void_p) for f, _ in WEIGHT_FIELDS] + [(f, C.c_void_p) for f, _, _ in BN_FIELDS] + \
        [("logit_h_w", C.c_void_p * (MAX_LOGIT_LAYERS - 1)), ("logit_h_b", C.c_void_p * (MAX_LOGIT_LAYERS - 1))]


class FcDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "Dfc", "H", "E", "V1", "S", "dtype")] + [("drop_p", C.c_float)]


FC_WEIGHT_FIELDS = [
    ("img_embed_w", "img_embed.weight"), ("img_embed_b", "img_embed.bias"),
    ("i2h_w", "core.i2h.weight"), ("i2h_b", "core.i2h.bias"),
    ("h2h_w", "core.h2h.weight"), ("h2h_b", "core.h2h.bias"),
    ("embed_w", "embed.weight"), ("logit_w", "logit.weight"), ("logit_b", "logit.bias"),
]


class FcWeights(C.Structure):
    _fields_ = [(f, C.c_void_p) for f, _ in FC_WEIGHT_FIELDS]


DISC_MAX_WIDTHS = 4
SITE_DISC = 6000


class DiscDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "N", "L", "V1", "E", "F", "nw")] + [("widths", C.c_int32 * DISC_MAX_WIDTHS), ("drop_p", C.c_float)]


class DiscWeights(C.Structure):
    _fields_ = [("embed_w", C.c_void_p), ("conv_w", C.c_void_p * DISC_MAX_WIDTHS), ("conv_b", C.c_void_p * DISC_MAX_WIDTHS),
                ("hw_w", C.c_void_p), ("hw_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p)]


GCN_MAX_LAYERS = 3


class GcnDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "N", "R", "D", "H", "layers")]


class GcnWeights(C.Structure):
    _fields_ = [("w", C.c_void_p * GCN_MAX_LAYERS), ("b", C.c_void_p * GCN_MAX_LAYERS)]


NMT_MAX_LAYERS = 4
SITE_NMT_ENC0, SITE_NMT_DEC0, SITE_NMT_OUT0 = 1000, 2000, 4000   # + layer ; + layer*256 + step ; + step


class NmtDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "S", "T", "H", "W", "layers", "Vs", "Vt", "dtype")] + [("drop_p", C.c_float),
                ("recurrence", C.c_int32), ("rnn_status", C.c_void_p), ("tgt_live_rows", C.c_void_p), ("tgt_live_count", C.c_int32)]


class NmtWeights(C.Structure):
    _fields_ = [("enc_lut", C.c_void_p), ("enc_lin_w", C.c_void_p), ("enc_lin_b", C.c_void_p)] + \
               [(f, (C.c_void_p * 2) * NMT_MAX_LAYERS) for f in ("enc_w_ih", "enc_w_hh", "enc_b_ih", "enc_b_hh")] + \
               [("dec_lut", C.c_void_p)] + \
               [(f, C.c_void_p * NMT_MAX_LAYERS) for f in ("dec_w_ih", "dec_w_hh", "dec_b_ih", "dec_b_hh")] + \
               [("attn_in_w", C.c_void_p), ("attn_out_w", C.c_void_p), ("gen_w", C.c_void_p), ("gen_b", C.c_void_p)]


def nmt_weight_keys(layers):
    """(struct field, layer, direction, reference state_dict key) in NMTModel.state_dict() order, generator last."""
    out = [("enc_lut", None, None, "encoder.embeddings.word_lut.weight"),
           ("enc_lin_w", None, None, "encoder.embeddings.linear.weight"),
           ("enc_lin_b", None, None, "encoder.embeddings.linear.bias")]
    for l in range(layers):
        for d, suf in ((0, ""), (1, "_reverse")):
            for f, k in (("enc_w_ih", "weight_ih"), ("enc_w_hh", "weight_hh"), ("enc_b_ih", "bias_ih"), ("enc_b_hh", "bias_hh")):
                out.append((f, l, d, "encoder.rnn.%s_l%d%s" % (k, l, suf)))
    out.append(("dec_lut", None, None, "decoder.embeddings.word_lut.weight"))
    for l in range(layers):
        for f, k in (("dec_w_ih", "weight_ih"), ("dec_w_hh", "weight_hh"), ("dec_b_ih", "bias_ih"), ("dec_b_hh", "bias_hh")):
            out.append((f, l, None, "decoder.rnn.layers.%d.%s" % (l, k)))
    out += [("attn_in_w", None, None, "decoder.attn.linear_in.weight"),
            ("attn_out_w", None, None, "decoder.attn.linear_out.weight"),
            ("gen_w", None, None, "generator.0.weight"), ("gen_b", None, None, "generator.0.bias")]
    return out


def nmt_weights(tensors, layers):
    """Fill a uic_nmt_weights from {reference key: contiguous f32 device tensor}."""
    w = NmtWeights()
    for f, l, d, k in nmt_weight_keys(layers):
        t = tensors[k]
        assert t.dtype == torch.float32 and t.is_contiguous(), k
        if l is None:
            setattr(w, f, t.data_ptr())
        elif d is None:
            getattr(w, f)[l] = t.data_ptr()
        else:
            getattr(w, f)[l][d] = t.data_ptr()
    return w


GATHERED_FIELDS = [
    # (struct field of uic_topdown_gathered, struct field of uic_topdown_weights it shadows, gather group)
    ("embed_w", "embed_w", 2), ("fc_w", "fc_w", 2), ("att_w", "att_w", 3), ("logit_w", "logit_w", 0), ("ctx2att_w", "ctx2att_w", 3),
    ("att_lstm_w_ih", "att_lstm_w_ih", 2), ("att_lstm_w_hh", "att_lstm_w_hh", 1), ("lang_lstm_w_ih", "lang_lstm_w_ih", 1),
    ("lang_lstm_w_hh", "lang_lstm_w_hh", 1), ("h2att_w", "h2att_w", 3),
]


class Gathered(C.Structure):
    """uic_topdown_gathered: the all-gathered operand-dtype weights of a sharded data-parallel rank + one event per gather group."""
    _fields_ = [(f, C.c_void_p) for f, _, _ in GATHERED_FIELDS] + [("logit_h_w", C.c_void_p * (MAX_LOGIT_LAYERS - 1)), ("ready", C.c_void_p * 4)]


class Batch(C.Structure):
    _fields_ = [("fc_feats", C.c_void_p), ("att_feats", C.c_void_p), ("att_masks", C.c_void_p),
                ("labels", C.c_void_p), ("ld_labels", C.c_int32),
                ("masks", C.c_void_p), ("ld_masks", C.c_int32),
                ("grad_scale", C.c_void_p), ("ld_grad_scale", C.c_int32), ("ss_prob", C.c_float),
                ("d_att_feats", C.c_void_p), ("d_fc_feats", C.c_void_p),
                ("live_rows", C.c_void_p), ("live_count", C.POINTER(C.c_int32))]


_SIGS = {
    "uic_last_error_string": (C.c_char_p, []),
    "uic_version": (C.c_int, []),
    "uic_comm_unique_id": (C.c_int, [C.c_void_p]),
    "uic_gcn_workspace_bytes": (C.c_size_t, [C.POINTER(GcnDims)]),
    "uic_gcn_forward": (C.c_int, [C.POINTER(GcnDims), C.POINTER(GcnWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_gcn_backward": (C.c_int, [C.POINTER(GcnDims), C.POINTER(GcnWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GcnWeights),
                                   C.c_void_p, C.c_void_p]),
    "uic_disc_workspace_bytes": (C.c_size_t, [C.POINTER(DiscDims)]),
    "uic_disc_forward": (C.c_int, [C.POINTER(DiscDims), C.POINTER(DiscWeights), C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_disc_bce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_disc_backward": (C.c_int, [C.POINTER(DiscDims), C.POINTER(DiscWeights), C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.POINTER(DiscWeights), C.c_void_p]),
    "uic_comm_init": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "uic_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "uic_comm_destroy": (C.c_int, [C.c_void_p]),
    "uic_comm_proxy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "uic_comm_proxy_oneway": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "uic_comm_reduce_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "uic_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "uic_comm_group_start": (C.c_int, []),
    "uic_comm_group_end": (C.c_int, []),
    "uic_topdown_refresh_weights_gathered": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.POINTER(Gathered), C.c_void_p, C.c_int32, C.c_void_p]),
    "uic_adam_step_ranges": (C.c_int, [C.c_void_p] * 4 + [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)] + [C.c_float] * 4 +
                             [C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "uic_topdown_step_marks": (C.c_int, [C.c_int32, C.POINTER(C.c_float)]),
    "uic_topdown_workspace_bytes": (C.c_size_t, [C.POINTER(Dims)]),
    "uic_topdown_derived_bytes": (C.c_size_t, [C.POINTER(Dims)]),
    "uic_topdown_refresh_weights": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.c_void_p]),
    "uic_topdown_refresh_weights_deferred": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.c_void_p]),
    "uic_topdown_forward": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.POINTER(Batch), C.c_int32,
                                      C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_topdown_xe_loss": (C.c_int, [C.POINTER(Dims), C.POINTER(Batch), C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_topdown_backward": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.POINTER(Batch), C.c_int32,
                                       C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Weights),
                                       C.c_void_p]),
    "uic_topdown_xe_train_step": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.POINTER(Batch), C.c_int32,
                                            C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(Weights), C.c_void_p]),
    "uic_topdown_sample_beam": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.POINTER(Batch), C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_topdown_grad_ready_wait": (C.c_int, [C.c_void_p, C.c_int32]),
    "uic_nmt_grad_ready_wait": (C.c_int, [C.c_void_p, C.c_int32]),
    "uic_topdown_beam_done_lists": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5),
    "uic_topdown_prepare_feature": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.POINTER(Batch), C.c_int32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_topdown_logprobs_state": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p] + [C.c_void_p] * 7 +
                                   [C.c_int32, C.c_int32, C.c_uint32] + [C.c_void_p] * 5),
    "uic_topdown_sample": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.POINTER(Batch), C.c_int32,
                                     C.c_int32, C.c_float, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_topdown_sample_train": (C.c_int, [C.POINTER(Dims), C.POINTER(Weights), C.c_void_p, C.POINTER(Batch), C.c_int32,
                                     C.c_int32, C.c_float, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_fc_workspace_bytes": (C.c_size_t, [C.POINTER(FcDims)]),
    "uic_fc_forward": (C.c_int, [C.POINTER(FcDims), C.POINTER(FcWeights), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_fc_xe_loss": (C.c_int, [C.POINTER(FcDims), C.POINTER(Batch), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_fc_backward": (C.c_int, [C.POINTER(FcDims), C.POINTER(FcWeights), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_uint32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FcWeights), C.c_void_p]),
    "uic_fc_sample": (C.c_int, [C.POINTER(FcDims), C.POINTER(FcWeights), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_float,
                                C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_nmt_workspace_bytes": (C.c_size_t, [C.POINTER(NmtDims)]),
    "uic_nmt_workspace_ptr": (C.c_void_p, [C.POINTER(NmtDims), C.c_void_p, C.c_char_p]),
    "uic_nmt_forward_loss": (C.c_int, [C.POINTER(NmtDims), C.POINTER(NmtWeights), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_nmt_backward": (C.c_int, [C.POINTER(NmtDims), C.POINTER(NmtWeights), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p,
                                   C.c_int32, C.c_uint32, C.c_void_p, C.POINTER(NmtWeights), C.c_void_p]),
    "uic_fc_sample_beam": (C.c_int, [C.POINTER(FcDims), C.POINTER(FcWeights), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_nmt_translate_workspace_bytes": (C.c_size_t, [C.POINTER(NmtDims), C.c_int32, C.c_int32]),
    "uic_nmt_translate": (C.c_int, [C.POINTER(NmtDims), C.POINTER(NmtWeights), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "uic_topdown_workspace_ptr": (C.c_void_p, [C.POINTER(Dims), C.c_void_p, C.c_char_p]),
    "uic_linear": (C.c_int, [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                             C.c_void_p, C.c_int32, C.c_void_p]),
    "uic_linear_f32a": (C.c_int, [C.c_int32] * 3 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "uic_linear_partials": (C.c_int, [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "uic_ciderd_table_slots": (C.c_int64, [C.c_int64]),
    "uic_ciderd_table_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]),
    "uic_ciderd_scores": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_void_p]),
    "uic_bleu_scores": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_void_p]),
    "uic_ciderd_reward": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "uic_loader_scan": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_void_p, C.c_int32]),
    "uic_loader_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "uic_loader_inflate": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32]),
    "uic_loader_inflate_pair": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "uic_att_batch_assemble": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p] * 3),
    "uic_linear_wgrad": (C.c_int, [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "uic_lstm_cell_fwd": (C.c_int, [C.c_int32] * 4 + [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_int32)] + [C.c_void_p] * 9),
    "uic_lstm_cell_bwd": (C.c_int, [C.c_int32] * 3 + [C.c_void_p] * 7),
    "uic_attention_fwd": (C.c_int, [C.c_int32] * 5 + [C.c_void_p] * 9),
    "uic_attention_bwd_step": (C.c_int, [C.c_int32] * 5 + [C.c_void_p] * 9),
    "uic_attention_bwd_accum": (C.c_int, [C.c_int32] * 6 + [C.c_void_p] * 10),
    "uic_adam_step": (C.c_int, [C.c_void_p] * 4 + [C.c_size_t] + [C.c_float] * 4 + [C.c_int32, C.c_float, C.c_void_p]),
    "uic_adam_step_guarded": (C.c_int, [C.c_void_p] * 4 + [C.c_size_t] + [C.c_float] * 4 + [C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "uic_grad_sqnorm": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_adam_step_clip": (C.c_int, [C.c_void_p] * 4 + [C.c_size_t] + [C.c_float] * 4 + [C.c_int32, C.c_float, C.c_float,
                                                                                      C.c_void_p, C.c_void_p]),
    "uic_adam_step_clip_guarded": (C.c_int, [C.c_void_p] * 4 + [C.c_size_t] + [C.c_float] * 4 + [C.c_int32, C.c_float, C.c_float,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "uic_lm_criterion": (C.c_int, [C.c_int32] * 3 + [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "uic_reward_criterion": (C.c_int, [C.c_int32, C.c_int32] + [C.c_void_p] * 6),
    "uic_cast_from_f32": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uic_cast_to_f32": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uic_transpose": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "uic_dropout_mask": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load libuic_hip.so; raise (never fall back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libuic_hip.so is not built (%s). Run `python -m unpaired_image_captioning_amd.build`; "
                "there is no CPU fallback for the captioner hot path." % LIB_PATH)
        lib = C.CDLL(os.environ.get("UIC_LIB", LIB_PATH))      # UIC_LIB: a variant build to A/B (tools/build_variant.sh)
        for name, (res, args) in _SIGS.items():
            if "UIC_LIB" in os.environ and not hasattr(lib, name):
                continue                                        # (an A/B variant of an older source tree)
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_status = {}


class PersistentTimeout(RuntimeError):
    """A bounded spin of a persistent recurrence kernel gave up (another process on the GPU, a long kernel on another stream):
    the results of that call are invalid and the optimizer step that followed was skipped on the device."""


def status_words(device=None):
    """The 4 status words of the persistent recurrence kernels on `device` (a device tensor, allocated on first use): what
    TopDownEngine.dims() hands the library as uic_topdown_dims.rnn_status."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _status:
        with torch.cuda.device(idx):
            _status[idx] = torch.zeros(4, dtype=torch.int32, device="cuda:%d" % idx)
    return _status[idx]


def persistent_status(device=None):
    """[timeout code, XCD-local launches, SAFE launches] of the persistent recurrence kernels on `device`; synchronises.
    A non-zero timeout code raises (and is cleared)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    status_words(dev)
    vals = [int(v) for v in _status[idx].cpu().tolist()]
    if vals[0] != 0:
        _status[idx].zero_()
        raise PersistentTimeout("persistent recurrence kernel timed out (code 0x%x): results of the last calls are invalid" % vals[0])
    return vals[:3]


def check(rc, what=""):
    if rc != 0:
        msg = load().uic_last_error_string()
        raise RuntimeError("libuic_hip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libuic_hip needs device tensors; got a %s tensor" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("libuic_hip needs contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def dtype_id(name):
    if isinstance(name, int):
        return name
    return DTYPE_IDS[str(name).lower()]
