"""FCModel_NMT (the `fc` caption model, P/models/__init__.py:24-26) behind the reference's constructor, state_dict and
call contract (P/models/FCModel_NMT.py:54-217), computed by libuic_hip.so.

Reference bug kept visible (SURVEY.md finding 5): the reference's `_forward(fc_feats, att_feats, seq, att_masks)` cannot
take the trainer's 5-argument call (P/trainer.py:164) and raises TypeError there; this class accepts BOTH conventions --
`(fc, att, seq[, att_masks])` as the reference defines it and `(fc, attri, att, seq, att_masks)` as the trainer calls
it -- and ignores attri / att feats, which is what BASELINE config 1 ("plumbing") needs.
"""
import ctypes as C

import torch
import torch.nn as nn

from .CaptionModel import CaptionModel
from .. import _lib
from .._lib import Batch, FcDims, FcWeights, FC_WEIGHT_FIELDS, check, ptr, stream


class LSTMCore(nn.Module):
    def __init__(self, opt):
        super(LSTMCore, self).__init__()
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.drop_prob_lm = opt.drop_prob_lm
        self.i2h = nn.Linear(self.input_encoding_size, 5 * self.rnn_size)
        self.h2h = nn.Linear(self.rnn_size, 5 * self.rnn_size)
        self.dropout = nn.Dropout(self.drop_prob_lm)


class _FcEngine(object):
    def __init__(self, model):
        self.lib = _lib.load()
        self.m = model
        self.dtype = _lib.dtype_id(model.compute_dtype)
        self._pool = {}

    def dims(self, N, S):
        m = self.m
        return FcDims(N=N, Dfc=m.fc_feat_size, H=m.rnn_size, E=m.input_encoding_size, V1=m.vocab_size + 1, S=S,
                      dtype=self.dtype, drop_p=m.drop_prob_lm)

    def workspace(self, d, device):
        key = (d.N, d.S, d.dtype)
        free = self._pool.setdefault(key, [])
        if free:
            return free.pop()
        nbytes = self.lib.uic_fc_workspace_bytes(C.byref(d))
        if nbytes == 0:
            check(-1, "uic_fc_workspace_bytes")
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def release(self, d, ws):
        self._pool.setdefault((d.N, d.S, d.dtype), []).append(ws)

    @staticmethod
    def weights(tensors):
        w = FcWeights()
        for f, k in FC_WEIGHT_FIELDS:
            setattr(w, f, ptr(tensors[k]))
        return w

    @staticmethod
    def batch(fc, labels=None, masks=None):
        b = Batch()
        b.fc_feats = ptr(fc)
        b.labels = ptr(labels)
        b.ld_labels = labels.shape[1] if labels is not None else 0
        b.masks = ptr(masks)
        b.ld_masks = masks.shape[1] if masks is not None else 0
        return b


class _FcForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, fc, seq, s_run, *params):
        eng = model.engine
        pd = dict(zip(model.param_names, params))
        N, S = fc.shape[0], seq.shape[1]
        d = eng.dims(N, S)
        ws = eng.workspace(d, fc.device)
        seed = model.next_seed()
        logp = torch.zeros(N, S - 1, d.V1, dtype=torch.float32, device=fc.device)
        w = eng.weights(pd)
        b = eng.batch(fc, seq)
        check(eng.lib.uic_fc_forward(C.byref(d), C.byref(w), C.byref(b), s_run, int(model.training), seed, ptr(ws), ptr(logp),
                                     stream()), "fc_forward")
        ctx.model, ctx.ws, ctx.d = model, ws, d
        ctx.call = (s_run, model.training, seed)
        ctx.inputs = (fc, seq)
        ctx.params = params
        ctx.save_for_backward(logp)
        return logp

    @staticmethod
    def backward(ctx, g):
        model = ctx.model
        eng = model.engine
        (logp,) = ctx.saved_tensors
        s_run, training, seed = ctx.call
        fc, seq = ctx.inputs
        pd = dict(zip(model.param_names, ctx.params))
        grads = {k: torch.empty_like(v) for k, v in pd.items()}
        w, gw, b = eng.weights(pd), eng.weights(grads), eng.batch(fc, seq)
        check(eng.lib.uic_fc_backward(C.byref(ctx.d), C.byref(w), C.byref(b), s_run, int(training), seed, ptr(ctx.ws),
                                      ptr(g.contiguous()), ptr(logp), C.byref(gw), stream()), "fc_backward")
        eng.release(ctx.d, ctx.ws)
        ctx.ws = None
        return (None, None, None, None) + tuple(grads[k] for k in model.param_names)


class FCModel_NMT(CaptionModel):
    def __init__(self, opt):
        super(FCModel_NMT, self).__init__()
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_type = getattr(opt, 'rnn_type', 'LSTM')
        self.rnn_size = opt.rnn_size
        self.num_layers = opt.num_layers
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = opt.seq_length
        self.fc_feat_size = opt.fc_feat_size
        self.ss_prob = 0.0
        self.img_embed = nn.Linear(self.fc_feat_size, self.input_encoding_size)
        self.core = LSTMCore(opt)
        self.embed = nn.Embedding(self.vocab_size + 1, self.input_encoding_size)
        self.logit = nn.Linear(self.rnn_size, self.vocab_size + 1)
        self.init_weights()
        self.compute_dtype = getattr(opt, 'compute_dtype', 'bf16')
        self._engine = None
        self._seed_counter = int(getattr(opt, 'seed', 0) or 0) & 0x7FFFFFFF

    def init_weights(self):
        initrange = 0.1
        self.embed.weight.data.uniform_(-initrange, initrange)
        self.logit.bias.data.fill_(0)
        self.logit.weight.data.uniform_(-initrange, initrange)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = _FcEngine(self)
        return self._engine

    @property
    def param_names(self):
        return [k for _, k in FC_WEIGHT_FIELDS]

    def param_dict(self):
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in self.param_names}

    def next_seed(self):
        self._seed_counter = (self._seed_counter * 1103515245 + 12345) & 0x7FFFFFFF
        return self._seed_counter

    @staticmethod
    def _steps_to_run(seq):
        """Early break of _forward (:115-116): stop at the first i >= 2 with seq[:, i-1] all zero."""
        S = seq.size(1)
        zero_cols = (seq[:, 1:S - 1].sum(0) == 0).nonzero()
        return int(zero_cols[0].item()) + 2 if zero_cols.numel() else S

    def _forward(self, fc_feats, *rest):
        # reference signature (fc, att, seq[, att_masks]) or trainer call (fc, attri, att, seq, att_masks)
        seq = next(r for r in rest if torch.is_tensor(r) and r.dtype == torch.int64)
        if self.training and self.ss_prob > 0.0:
            # P/models/FCModel_NMT.py:108 samples from exp(outputs[-1]) -- the [S, V+1] log-prob slab of the LAST BATCH ROW, not
            # the previous step's [N, V+1] -- and then indexes those S draws with batch-row indices (:109-110): an index error
            # as soon as a selected row index reaches S.  There is no behaviour to reproduce.
            raise NotImplementedError("FCModel_NMT scheduled sampling is broken in the reference (outputs[-1] indexes the "
                                      "batch, P/models/FCModel_NMT.py:108); not provided")
        s_run = self._steps_to_run(seq)
        params = [self.param_dict()[k] for k in self.param_names]
        return _FcForward.apply(self, fc_feats.contiguous().float(), seq.contiguous(), s_run, *params)

    def _sample_beam(self, fc_feats, att_feats, att_masks=None, opt={}):
        """FCModel_NMT._sample_beam (P/models/FCModel_NMT.py:136-162) over CaptionModel.beam_search, all images in one
        device pass; returns [N, L] tensors.  `self.done_beams[k]` holds the winning beam of image k."""
        beam_size = opt.get('beam_size', 10)
        group_size = opt.get('group_size', 1)
        if group_size > 1:
            # diverse groups (CaptionModel.py:100-177): the caller only ever receives done_beams[k][0] (FCModel_NMT.py:159-160),
            # the best beam of group 0, which never sees a diversity penalty -- a plain search over its bdash beams
            beam_size = beam_size // group_size
        if self.training:
            raise NotImplementedError("beam search runs in eval mode")
        assert beam_size <= self.vocab_size + 1
        eng = self.engine
        fc = fc_feats.contiguous().float()
        n_img, L = fc.shape[0], self.seq_length
        fcb = fc.repeat_interleave(beam_size, 0).contiguous()
        d = eng.dims(n_img * beam_size, L + 2)
        ws = eng.workspace(d, fc.device)
        seq = torch.zeros(n_img, L, dtype=torch.int64, device=fc.device)
        lp = torch.zeros(n_img, L, dtype=torch.float32, device=fc.device)
        with torch.no_grad():
            w = eng.weights({k: v.detach() for k, v in self.param_dict().items()})
            b = eng.batch(fcb)
            check(eng.lib.uic_fc_sample_beam(C.byref(d), C.byref(w), C.byref(b), L, int(beam_size), int(opt.get('decoding_constraint', 0)),
                                             int(opt.get('max_ppl', 0)), ptr(ws), ptr(seq), ptr(lp), stream()), "fc_sample_beam")
        eng.release(d, ws)
        self.done_beams = [[{'seq': seq[k], 'logps': lp[k]}] for k in range(n_img)]
        return seq, lp

    def _sample(self, fc_feats, *rest, **kw):
        opt = kw.get('opt', None)
        if opt is None:
            opt = next((r for r in rest if isinstance(r, dict)), {})
        if opt.get('beam_size', 1) > 1:
            # P/models/FCModel_NMT.py:168 calls `self._sample_beam(fc_feats, att_feats, opt)`: `opt` lands in the att_masks
            # slot, so the reference searches with the DEFAULT options (beam_size 10, no constraint, no max_ppl) whatever the
            # caller asked for.  Kept, so that eval scripts decode the same captions.
            att_feats = rest[0] if rest else None
            return self._sample_beam(fc_feats, att_feats, opt)
        eng = self.engine
        fc = fc_feats.contiguous().float()
        N, L = fc.shape[0], self.seq_length
        d = eng.dims(N, L + 2)
        ws = eng.workspace(d, fc.device)
        seq = torch.zeros(N, L + 1, dtype=torch.int64, device=fc.device)
        lp = torch.zeros(N, L + 1, dtype=torch.float32, device=fc.device)
        with torch.no_grad():
            w = eng.weights({k: v.detach() for k, v in self.param_dict().items()})
            b = eng.batch(fc)
            check(eng.lib.uic_fc_sample(C.byref(d), C.byref(w), C.byref(b), L, int(opt.get('sample_max', 1)),
                                        float(opt.get('temperature', 1.0)), self.next_seed(), ptr(opt.get('forced_tokens')),
                                        ptr(ws), ptr(seq), ptr(lp), stream()), "fc_sample")
        eng.release(d, ws)
        return seq, lp
