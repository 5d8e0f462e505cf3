"""TopDownModel behind the reference's constructor / call / state_dict contract
(P/models/AttModel.py:55-253, 421-446, 529-558, 686-690), computed by libuic_hip.so.

The nn.Module tree below exists only to own parameters under the reference's names
(`embed.0.weight`, `core.att_lstm.weight_ih`, ...), so `state_dict()` / `load_state_dict()`
round-trip the reference's `model_i2t-best.pth` unchanged; none of these sub-modules'
`forward` is ever called.  There is no eager fallback: without the HIP library, or on a
CPU tensor, every call raises.
"""
import torch
import torch.nn as nn

from .CaptionModel import CaptionModel
from .. import _lib
from ..topdown_engine import TopDownEngine


class Attention(nn.Module):
    def __init__(self, opt):
        super(Attention, self).__init__()
        self.rnn_size = opt.rnn_size
        self.att_hid_size = opt.att_hid_size
        self.h2att = nn.Linear(self.rnn_size, self.att_hid_size)
        self.alpha_net = nn.Linear(self.att_hid_size, 1)


class TopDownCore(nn.Module):
    def __init__(self, opt, use_maxout=False):
        super(TopDownCore, self).__init__()
        self.drop_prob_lm = opt.drop_prob_lm
        self.att_lstm = nn.LSTMCell(opt.input_encoding_size + opt.rnn_size * 2, opt.rnn_size)
        self.lang_lstm = nn.LSTMCell(opt.rnn_size * 2, opt.rnn_size)
        self.attention = Attention(opt)


class _TopDownForward(torch.autograd.Function):
    """log-probs = AttModel._forward(...); backward = the library's BPTT."""

    @staticmethod
    def forward(ctx, model, fc, att, seq, att_masks, t_run, *params):
        eng = model.engine
        names = model.param_names
        pd = dict(zip(names, params))
        seed = model.next_seed()
        training = model.training
        ss_prob = float(model.ss_prob) if training else 0.0          # AttModel.py:130
        logp, ws, (d, w, b) = eng.forward(pd, fc, att, att_masks, seq, t_run, training, seed, want_logprobs=True, ss_prob=ss_prob)
        ctx.model = model
        ctx.ws = ws
        ctx.call = (d, t_run, training, seed, ss_prob)
        ctx.inputs = (fc, att, att_masks, seq)
        ctx.params = params
        ctx.save_for_backward(logp)
        return logp

    @staticmethod
    def backward(ctx, g):
        model = ctx.model
        eng = model.engine
        if ctx.ws is None:
            # the reference calls loss.backward(retain_graph=True) (P/trainer.py:173) but never walks the graph twice; here the
            # first backward hands the forward's workspace (activations of every decode step) back to the engine's pool
            raise RuntimeError("TopDownModel: backward ran a second time over the same forward pass; its workspace was released "
                               "after the first one (retain_graph=True keeps the autograd graph, not the HIP workspace) -- run the "
                               "forward again")
        (logp,) = ctx.saved_tensors
        d, t_run, training, seed, ss_prob = ctx.call
        fc, att, att_masks, seq = ctx.inputs
        pd = dict(zip(model.param_names, ctx.params))
        w = eng.refresh(pd, d)
        # features that came out of an encoder (the scene-graph GCN of configs[4]) ask for their gradient; the reference's
        # own features are data and pay nothing
        d_fc, d_att = eng.input_grad_buffers(fc, att, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        b = eng.batch_struct(fc, att, att_masks, seq, ss_prob=ss_prob, d_fc=d_fc, d_att=d_att)
        grads = {k: torch.empty_like(v) for k, v in pd.items()}
        eng.backward(ctx.ws, d, w, b, t_run, training, seed, grads, dlogprobs=g.contiguous(), logprobs=logp)
        eng.release(ctx.ws)
        ctx.ws = None
        if d_att is not None:
            d_att = d_att[..., :att.shape[-1]]
        return (None, d_fc, d_att, None, None, None) + tuple(grads[k] for k in model.param_names)


class _HeldWorkspace(object):
    """A checked-out engine workspace that goes back to the pool when backward has used it -- or when the graph that
    holds it is dropped without a backward pass."""

    def __init__(self, eng, ws):
        self.eng, self.ws = eng, ws

    def take(self):
        ws, self.ws = self.ws, None
        return ws

    def __del__(self):
        if self.ws is not None:
            try:
                self.eng.release(self.ws)
            except Exception:
                pass
            self.ws = None


class _TopDownSample(torch.autograd.Function):
    """Multinomial sampling pass of the self-critical step (P/trainer.py:167): returns (seq, seqLogprobs) with
    seqLogprobs differentiable.  Backward replays the sampled sequence teacher-forced with the SAME dropout seed
    (the masks are a pure function of seed/site/index) and feeds d loss / d logprob of the sampled tokens
    straight into the fused training step."""

    @staticmethod
    def forward(ctx, model, fc, att, att_masks, sample_kw, *params):
        eng = model.engine
        pd = dict(zip(model.param_names, params))
        seed = model.next_seed()
        # in train mode the pass keeps its forward (training layout): backward starts at the criterion instead of replaying
        keep = bool(model.training) and getattr(model, 'scst_keep_forward', True)
        if keep:
            seq, lp, ws = eng.sample(pd, fc, att, att_masks, model.seq_length, seed=seed, training=True, keep_forward=True, **sample_kw)
            ctx.ws = _HeldWorkspace(eng, ws)
        else:
            seq, lp = eng.sample(pd, fc, att, att_masks, model.seq_length, seed=seed, training=model.training, **sample_kw)
            ctx.ws = None
        ctx.model = model
        ctx.call = (seed, model.training)
        ctx.inputs = (fc, att, att_masks, seq)
        ctx.params = params
        ctx.mark_non_differentiable(seq)
        return seq, lp

    @staticmethod
    def backward(ctx, g_seq, g_lp):
        model = ctx.model
        eng = model.engine
        seed, training = ctx.call
        fc, att, att_masks, seq = ctx.inputs
        pd = dict(zip(model.param_names, ctx.params))
        N, L = seq.shape
        labels = torch.zeros(N, L + 2, dtype=torch.int64, device=seq.device)
        labels[:, 1:L + 1] = seq
        # direct: every p.grad IS its view of the optimizer's flat gradient arena (Trainer.train_self_critical set both up):
        # the kernels write there and autograd gets nothing to copy or accumulate; otherwise fresh tensors go back to autograd
        sink = getattr(model, '_grad_sink', None)
        direct = sink is not None and all(p.grad is sink[k] for k, p in zip(model.param_names, ctx.params))
        grads = sink if direct else {k: torch.empty_like(v) for k, v in pd.items()}
        # d loss / d logits = -g * (softmax - onehot)  ->  gradient weight of position (n, t) is -g[n, t]
        # bit 1: the sampling pass of this iteration already updated the BatchNorm running statistics
        held = ctx.ws.take() if ctx.ws is not None else None
        if held is not None:
            # the sampling pass left its whole forward in `held` (dims T = L + 1: the labels keep their L + 2 columns)
            eng.xe_train_step(pd, fc, att, att_masks, labels, None, L, int(training) | 2, seed, grads,
                              grad_scale=(-g_lp).contiguous(), resume_ws=held)
        else:
            eng.xe_train_step(pd, fc, att, att_masks, labels[:, :L + 1].contiguous(), None, L, int(training) | 2, seed, grads,
                              grad_scale=(-g_lp).contiguous())
        if direct:
            return (None, None, None, None, None) + (None,) * len(model.param_names)
        return (None, None, None, None, None) + tuple(grads[k] for k in model.param_names)


class AttModel(CaptionModel):
    supports_grad_sink = True      # _TopDownSample.backward can write straight into Trainer's flat gradient arena

    def __init__(self, opt):
        super(AttModel, self).__init__()
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.num_layers = opt.num_layers
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = opt.seq_length
        self.fc_feat_size = opt.fc_feat_size
        self.att_feat_size = opt.att_feat_size
        self.att_hid_size = opt.att_hid_size
        self.use_bn = getattr(opt, 'use_bn', 0)
        self.ss_prob = 0.0  # Schedule sampling probability
        if self.use_bn not in (0, 1, 2):
            raise ValueError("use_bn=%r must be 0, 1 or 2" % (self.use_bn,))
        self.logit_layers = getattr(opt, 'logit_layers', 1)
        if not 1 <= self.logit_layers <= _lib.MAX_LOGIT_LAYERS:
            raise NotImplementedError("logit_layers=%r: the MI355X path supports 1..%d" % (self.logit_layers, _lib.MAX_LOGIT_LAYERS))

        self.embed = nn.Sequential(nn.Embedding(self.vocab_size + 1, self.input_encoding_size),
                                   nn.ReLU(),
                                   nn.Dropout(self.drop_prob_lm))
        self.fc_embed = nn.Sequential(nn.Linear(self.fc_feat_size, self.rnn_size),
                                      nn.ReLU(),
                                      nn.Dropout(self.drop_prob_lm))
        self.att_embed = nn.Sequential(*(
            ((nn.BatchNorm1d(self.att_feat_size),) if self.use_bn else ()) +
            (nn.Linear(self.att_feat_size, self.rnn_size),
             nn.ReLU(),
             nn.Dropout(self.drop_prob_lm)) +
            ((nn.BatchNorm1d(self.rnn_size),) if self.use_bn == 2 else ())))
        if self.logit_layers == 1:
            self.logit = nn.Linear(self.rnn_size, self.vocab_size + 1)
        else:       # P/models/AttModel.py:90-91: blocks [Linear(H, H), ReLU, Dropout(0.5)] in front of the vocabulary layer
            blocks = []
            for _ in range(self.logit_layers - 1):
                blocks += [nn.Linear(self.rnn_size, self.rnn_size), nn.ReLU(), nn.Dropout(0.5)]
            self.logit = nn.Sequential(*(blocks + [nn.Linear(self.rnn_size, self.vocab_size + 1)]))
        self.ctx2att = nn.Linear(self.rnn_size, self.att_hid_size)

        # MI355X engine state (not part of the checkpoint)
        self.compute_dtype = getattr(opt, 'compute_dtype', 'bf16')
        self._engine = None
        self._seed_counter = int(getattr(opt, 'seed', 0) or 0) & 0x7FFFFFFF

    # ------------------------------------------------------------------ engine plumbing
    @property
    def engine(self):
        if self._engine is None:
            self._engine = TopDownEngine(dict(V1=self.vocab_size + 1, E=self.input_encoding_size, H=self.rnn_size,
                                              A=self.att_hid_size, D=self.att_feat_size, Dfc=self.fc_feat_size),
                                         dtype=self.compute_dtype, drop_p=self.drop_prob_lm, use_bn=self.use_bn,
                                         logit_layers=self.logit_layers)
        if self.use_bn:      # buffers may have been re-homed by .cuda()/.to(): always hand the live tensors over
            self._engine.buffers = {k: v for k, v in self.named_buffers() if k.endswith(("running_mean", "running_var"))}
        return self._engine

    @property
    def param_names(self):
        return [k for _, k, is_param in _lib.weight_fields(self.use_bn, self.logit_layers) if is_param]

    def _bn_count_batch(self):
        """num_batches_tracked += 1, as a train-mode nn.BatchNorm1d forward does."""
        if self.use_bn and self.training:
            for m in self.att_embed:
                if isinstance(m, nn.BatchNorm1d):
                    m.num_batches_tracked += 1

    def state_dict(self, *args, **kwargs):
        """The reference's checkpoint contract.  Under the sharded data-parallel exchange with bf16 operands a rank's f32 masters
        are current only inside its own shard between steps: Trainer.gather_masters() (a collective) makes them whole again, and
        this call refuses to hand out a stale mixture."""
        stale = getattr(self, '_stale_masters', None)
        if stale is not None and stale():
            raise RuntimeError("state_dict(): this rank's f32 master weights are current only inside its own shard (sharded bf16 exchange); "
                               "call Trainer.gather_masters() -- or save_models() -- on EVERY rank first")
        return super(AttModel, self).state_dict(*args, **kwargs)

    def param_dict(self):
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in self.param_names}

    def next_seed(self):
        self._seed_counter = (self._seed_counter * 1103515245 + 12345) & 0x7FFFFFFF
        return self._seed_counter

    def init_hidden(self, bsz):
        weight = next(self.parameters())
        return (weight.new_zeros(self.num_layers, bsz, self.rnn_size),
                weight.new_zeros(self.num_layers, bsz, self.rnn_size))

    @staticmethod
    def _steps_to_run(seq):
        """Early break of AttModel._forward (:148-151): stop at the first all-zero column i >= 1."""
        T = seq.size(1) - 1
        zero_cols = (seq[:, 1:T].sum(0) == 0).nonzero()
        return int(zero_cols[0].item()) + 1 if zero_cols.numel() else T

    # ------------------------------------------------------------------ reference call surface
    def _forward(self, fc_feats, attri_feats, att_feats, seq, att_masks=None):
        t_run = self._steps_to_run(seq)
        self._bn_count_batch()
        fc = fc_feats.contiguous().float()
        att = att_feats.contiguous().float()
        am = att_masks.contiguous().float() if att_masks is not None else None
        params = [self.param_dict()[k] for k in self.param_names]
        return _TopDownForward.apply(self, fc, att, seq.contiguous(), am, t_run, *params)

    def clip_att(self, att_feats, att_masks):
        """P/models/AttModel.py:99-105: clip the region axis to the longest row of att_masks."""
        if att_masks is not None:
            max_len = int(att_masks.long().sum(1).max())
            att_feats = att_feats[:, :max_len].contiguous()
            att_masks = att_masks[:, :max_len].contiguous()
        return att_feats, att_masks

    def _prepare_feature(self, fc_feats, att_feats, att_masks):
        """P/models/AttModel.py:107-117 -> (fc', att', p_att, att_masks), f32 device tensors (no autograd: this is the
        decode-time call of beam search / ensembles; training goes through _forward)."""
        att_feats, att_masks = self.clip_att(att_feats, att_masks)
        fc = fc_feats.contiguous().float()
        att = att_feats.contiguous().float()
        am = att_masks.contiguous().float() if att_masks is not None else None
        with torch.no_grad():
            pd = {k: v.detach() for k, v in self.param_dict().items()}
            self._bn_count_batch()
            fc_p, att_p, p_att = self.engine.prepare_feature(pd, fc, att, am, training=self.training,
                                                             seed=self.next_seed() if self.training else 0)
        return fc_p, att_p, p_att, am

    def get_logprobs_state(self, it, fc_feats, att_feats, p_att_feats, att_masks, state, t=0):
        """P/models/AttModel.py:158-165: one decode step from prepared features -> (log-probs [N, V+1], new state);
        state = (h [2, N, H], c [2, N, H]) stacked (att_lstm, lang_lstm) as TopDownCore returns it (:445)."""
        with torch.no_grad():
            pd = {k: v.detach() for k, v in self.param_dict().items()}
            am = att_masks.contiguous().float() if att_masks is not None else None
            logp, h, c = self.engine.logprobs_state(pd, it.contiguous().long(), fc_feats.contiguous().float(), att_feats.contiguous().float(),
                                                    p_att_feats.contiguous().float(), am, state[0].contiguous().float(),
                                                    state[1].contiguous().float(), t=t, training=self.training,
                                                    seed=self._seed_counter if self.training else 0)
        return logp, (h, c)

    def _sample_beam(self, fc_feats, att_feats, att_masks=None, opt={}):
        """AttModel._sample_beam (P/models/AttModel.py:167-196) over CaptionModel.beam_search (P/models/CaptionModel.py:
        33-177): all images in one device pass.  `self.done_beams[k]` is the reference's list for image k: its finished
        beams sorted by -p (stable), the first beam_size of them, each {'seq', 'logps', 'unaug_p', 'p'} (:147-161,174-176)."""
        beam_size = opt.get('beam_size', 10)
        group_size = opt.get('group_size', 1)
        if group_size > 1:
            # diverse groups (CaptionModel.py:100-177): the caller only ever receives done_beams[k][0] (AttModel.py:193-194),
            # the best beam of group 0, which never sees a diversity penalty -- a plain search over its bdash beams
            beam_size = beam_size // group_size
        if self.training:
            raise NotImplementedError("beam search runs in eval mode (eval_utils.eval_split calls model.eval() first)")
        assert beam_size <= self.vocab_size + 1
        fc = fc_feats.contiguous().float()
        att = att_feats.contiguous().float()
        am = att_masks.contiguous().float() if att_masks is not None else None
        with torch.no_grad():
            pd = {k: v.detach() for k, v in self.param_dict().items()}
            seq, lp, (cnt, dp, dseq, dlp) = self.engine.sample_beam(pd, fc, att, am, self.seq_length, beam_size,
                                                                   opt.get('decoding_constraint', 0), opt.get('max_ppl', 0), done_lists=True)
        # the per-image lists are built on first access to `self.done_beams` (one D2H copy per array then): a caller that only wants
        # the best captions -- eval_split without `verbose_beam`, the pivot decode -- pays nothing for them
        self._done_raw = (cnt, dp, dseq, dlp, beam_size)
        self._done_beams = None
        return seq, lp

    @property
    def done_beams(self):
        if getattr(self, '_done_beams', None) is None and getattr(self, '_done_raw', None) is not None:
            cnt, dp, dseq, dlp, beam_size = self._done_raw
            cnt_h, dp_h, unaug = cnt.cpu().tolist(), dp.cpu(), dlp.sum(-1).cpu()
            out = []
            for k in range(len(cnt_h)):
                order = sorted(range(cnt_h[k]), key=lambda i: -float(dp_h[k, i]))[:beam_size]     # sorted() is stable, like the reference's
                out.append([{'seq': dseq[k, i], 'logps': dlp[k, i], 'unaug_p': float(unaug[k, i]), 'p': float(dp_h[k, i])} for i in order])
            self._done_beams = out
        return getattr(self, '_done_beams', None)

    @done_beams.setter
    def done_beams(self, value):
        self._done_beams, self._done_raw = value, None

    def _sample(self, fc_feats, attri_feats, att_feats, att_masks=None, opt={}):
        sample_max = opt.get('sample_max', 1)
        beam_size = opt.get('beam_size', 1)
        temperature = opt.get('temperature', 1.0)
        decoding_constraint = opt.get('decoding_constraint', 0)
        if beam_size > 1:
            return self._sample_beam(fc_feats, att_feats, att_masks, opt)
        self._bn_count_batch()
        fc = fc_feats.contiguous().float()
        att = att_feats.contiguous().float()
        am = att_masks.contiguous().float() if att_masks is not None else None
        # opt['captions_per_image'] = S > 1 (extension; not 'seq_per_img': callers pass all of vars(opt) as eval_kwargs, P/train.py): features come once per image, S captions are decoded per image
        kw = dict(sample_max=sample_max, temperature=temperature, decoding_constraint=decoding_constraint,
                  forced=opt.get('forced_tokens'), seq_per_img=int(opt.get('captions_per_image', 1) or 1))
        if torch.is_grad_enabled() and not sample_max:
            params = [self.param_dict()[k] for k in self.param_names]
            return _TopDownSample.apply(self, fc, att, am, kw, *params)
        with torch.no_grad():
            pd = {k: v.detach() for k, v in self.param_dict().items()}
            out = self.engine.sample(pd, fc, att, am, self.seq_length, seed=self.next_seed(), training=self.training, **kw)
        # A lone decode pass is ONE persistent launch (csrc/rnn_persist.hip) whose bounded spin gives up if its workgroups
        # cannot all become resident (another process on the GPU): the library then poisons the captions (token -1, log-prob
        # NaN) and sets the status word.  Callers of this path read the captions back at once (eval_utils.eval_split), so the
        # word is checked HERE -- one 16-byte read-back -- and a time-out raises at the call that suffered it.
        # `defer_status_check = True`: the caller checks at its own host sync (Trainer's self-critical step, timing loops).
        if not getattr(self, 'defer_status_check', False):
            _lib.persistent_status(out[0].device)
        return out


class TopDownModel(AttModel):
    def __init__(self, opt):
        super(TopDownModel, self).__init__(opt)
        self.num_layers = 2
        self.core = TopDownCore(opt)
