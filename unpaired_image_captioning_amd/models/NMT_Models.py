"""The pivot NMT teacher (zh -> en) behind the reference's constructors, state_dict keys and call contract
(P/models/NMT_Models.py:27-135 Embeddings/Encoder, :137-271 Decoder, :272-420 NMTModel), computed by libuic_hip.so
(`uic_nmt_forward_loss` / `uic_nmt_backward`, csrc/nmt.hip).

The reference runs the model and the loss as two python calls (P/trainer.py:178-179):

    outputs, attn, dec_state, upper_bounds = nmt_model(src, tgt, lengths, dec_state)
    loss = nmt_crit(loader, batch, outputs, attn)          # generator + NLLLoss(weight[PAD]=0, size_average=False)
    loss.backward()

The HIP path computes the generator, the loss and the `score` counters in the same device pass as the model (the
generator is attached to the model as `model.generator`, exactly where P/trainer.py:89 puts it).  `forward` therefore
returns `outputs` carrying the fused loss (`outputs.uic_loss`, `outputs.uic_stats`), and `misc.criterion.NMT_loss`
picks those up instead of re-running the generator; `loss.backward()` runs `uic_nmt_backward` and delivers the
gradients of every parameter (generator included).  Options outside the reference's training configuration for this
path (GRU cells, unidirectional encoder, coverage/fertility/copy attention, context gates, positional encodings,
a carried decoder state) raise NotImplementedError.

There is no eager fallback: the nn.LSTM / nn.LSTMCell / nn.Linear modules below are parameter containers that give the
reference's state_dict keys and initialisation; they are never called.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import NmtDims, check, ptr, stream

PAD = 0   # onmt.Constants.PAD


def _dict_size(dicts):
    return int(dicts.size()) if hasattr(dicts, 'size') else int(dicts)


def _unsupported(opt, name, allowed):
    v = getattr(opt, name, allowed[0])
    if v not in allowed:
        raise NotImplementedError("opt.%s=%r is outside the MI355X NMT hot path (supported: %r)" % (name, v, allowed))


class Embeddings(nn.Module):
    """P/models/NMT_Models.py:27-71: word_lut (+ ReLU(linear) for the encoder, which passes feature_dicts=[])."""

    def __init__(self, opt, dicts, feature_dicts=[]):
        super(Embeddings, self).__init__()
        _unsupported(opt, 'position_encoding', (False, 0, None))
        self.word_vec_size = opt.word_vec_size
        self.word_lut = nn.Embedding(_dict_size(dicts), opt.word_vec_size, padding_idx=PAD)
        self.feature_dicts = feature_dicts
        if feature_dicts is not None:
            self.linear = nn.Linear(opt.word_vec_size, opt.word_vec_size)

    def load_pretrained_vectors(self, emb_file):
        if emb_file is not None:
            self.word_lut.weight.data.copy_(torch.load(emb_file))


class Encoder(nn.Module):
    """P/models/NMT_Models.py:74-135."""

    def __init__(self, opt, dicts, feature_dicts=None):
        super(Encoder, self).__init__()
        _unsupported(opt, 'rnn_type', ('LSTM',))
        _unsupported(opt, 'brnn', (True, 1))
        self.layers = opt.layers
        self.num_directions = 2
        assert opt.rnn_size % self.num_directions == 0
        self.hidden_size = opt.rnn_size // self.num_directions
        self.embeddings = Embeddings(opt, dicts)
        self.rnn = nn.LSTM(opt.word_vec_size, self.hidden_size, num_layers=opt.layers, dropout=opt.dropout, bidirectional=True)


class StackedLSTM(nn.Module):
    """O/modules/StackedRNN.py:5-34."""

    def __init__(self, num_layers, input_size, rnn_size, dropout):
        super(StackedLSTM, self).__init__()
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        for _ in range(num_layers):
            self.layers.append(nn.LSTMCell(input_size, rnn_size))
            input_size = rnn_size


class GlobalAttention(nn.Module):
    """O/modules/GlobalAttention.py:40-167 with attn_type='dot', softmax transform, no coverage."""

    def __init__(self, dim):
        super(GlobalAttention, self).__init__()
        self.linear_in = nn.Linear(dim, dim, bias=False)
        self.linear_out = nn.Linear(dim * 2, dim, bias=False)
        self.mask = None

    def applyMask(self, mask):
        raise NotImplementedError("GlobalAttention.applyMask is only used by the beam-search translator (out of the hot path)")

    def applyMaskNone(self):
        self.mask = None


class Decoder(nn.Module):
    """P/models/NMT_Models.py:137-271 (input feed, dot attention)."""

    def __init__(self, opt, dicts):
        super(Decoder, self).__init__()
        _unsupported(opt, 'rnn_type', ('LSTM',))
        _unsupported(opt, 'input_feed', (1, True))
        _unsupported(opt, 'coverage_attn', (False, 0, None))
        _unsupported(opt, 'copy_attn', (False, 0, None))
        _unsupported(opt, 'context_gate', (None,))
        _unsupported(opt, 'attention_type', ('dot',))
        _unsupported(opt, 'attn_transform', ('softmax',))
        for flag in ('fertility', 'predict_fertility', 'guided_fertility', 'supervised_fertility'):
            _unsupported(opt, flag, (None, False, 0, 0.0))
        self.layers = opt.layers
        self.hidden_size = opt.rnn_size
        self.input_feed = 1
        self.embeddings = Embeddings(opt, dicts, None)
        self.rnn = StackedLSTM(opt.layers, opt.word_vec_size + opt.rnn_size, opt.rnn_size, opt.dropout)
        self.attn = GlobalAttention(opt.rnn_size)


class _NmtEngine(object):
    def __init__(self, model):
        self.lib = _lib.load()
        self.m = model
        self.dtype = _lib.dtype_id(model.compute_dtype)
        self._pool = {}

    def dims(self, B, S, T):
        m = self.m
        return NmtDims(B=B, S=S, T=T, H=m.decoder.hidden_size, W=m.encoder.embeddings.word_vec_size, layers=m.encoder.layers,
                       Vs=m.encoder.embeddings.word_lut.num_embeddings, Vt=m.decoder.embeddings.word_lut.num_embeddings,
                       dtype=self.dtype, drop_p=float(m.opt.dropout), recurrence=int(getattr(self, 'recurrence', 0)),
                       rnn_status=_lib.status_words(self.device()).data_ptr() if torch.cuda.is_available() else None)

    def device(self):
        """The device the model lives on (its status words -- written by the persistent kernels, read by Optim.step's guarded
        Adam and cleared by Trainer.train_nmt -- are that device's, whatever torch.cuda.current_device() is)."""
        w = self.m.decoder.embeddings.word_lut.weight
        return w.device if w.is_cuda else None

    def workspace(self, d, device):
        key = (d.B, d.S, d.T, d.dtype)
        free = self._pool.setdefault(key, [])
        if free:
            return free.pop()
        nbytes = self.lib.uic_nmt_workspace_bytes(C.byref(d))
        if nbytes == 0:
            check(-1, "uic_nmt_workspace_bytes")
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def release(self, d, ws):
        self._pool.setdefault((d.B, d.S, d.T, d.dtype), []).append(ws)

    def weights(self, tensors):
        for t in tensors.values():
            ptr(t)   # device + contiguity check
        return _lib.nmt_weights(tensors, self.m.encoder.layers)


def tgt_live_count(tgt):
    """The number of target positions that are not PAD (uic_nmt_dims.tgt_live_count): tgt [T, B] host array / tensor.  Attach
    `(None, count)` to the device batch as `batch.tgt.uic_live`: the step then compacts the targets on the device."""
    t = tgt.detach().cpu().numpy() if torch.is_tensor(tgt) else np.asarray(tgt)
    return int(np.count_nonzero(t[1:]))


def tgt_live_positions(tgt, device=None):
    """The list of the target positions that are not PAD (uic_nmt_dims.tgt_live_rows, include/uic_hip.h): position t * B + b is
    live when tgt[t + 1, b] != PAD -- NMTCriterion weighs every other one with zero (P/misc/criterion.py:126-136).
    tgt: [T, B] host array / tensor, as the Dataset assembles it.  Returns (int32 tensor of roundup(count, 128) positions, the tail
    -1, on `device`; count).  Attach it to the device batch as `batch.tgt.uic_live`: NMTModel.forward picks it up there."""
    t = tgt.detach().cpu().numpy() if torch.is_tensor(tgt) else np.asarray(tgt)
    flat = np.flatnonzero(np.ascontiguousarray(t[1:]) != 0).astype(np.int32)
    rows = np.full(((flat.size + 127) // 128) * 128, -1, dtype=np.int32)
    rows[:flat.size] = flat
    out = torch.from_numpy(rows)
    if device is not None and torch.device(device).type != "cpu":
        out = out.pin_memory().to(device, non_blocking=True)
    return out, int(flat.size)


class _NmtStep(torch.autograd.Function):
    """(loss, outputs, attn, stats) = fused NMTModel.forward + NMT_loss; only `loss` is differentiable."""

    @staticmethod
    def forward(ctx, model, src, tgt, lengths_host, lengths_dev, live, *params):
        eng = model.engine
        pd = dict(zip(model.param_names, params))
        S, B = src.shape
        T = tgt.shape[0]
        d = eng.dims(B, S, T)
        if live is not None:                       # tgt_live_positions(tgt): the generator runs over the non-PAD positions only
            rows, count = live
            if rows is not None:
                if rows.dtype != torch.int32 or not rows.is_contiguous() or rows.device != src.device or rows.numel() != ((int(count) + 127) // 128) * 128:
                    raise ValueError("tgt.uic_live must be (contiguous int32 device tensor of roundup(count, 128) positions or None, count)")
                d.tgt_live_rows = rows.data_ptr()
            d.tgt_live_count = int(count)
            ctx.live = live                        # (the backward call reads the same list through ctx.d)
        ws = eng.workspace(d, src.device)
        seed = model.next_seed()
        dev = src.device
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        stats = torch.zeros(2, dtype=torch.int32, device=dev)
        outputs = torch.empty(T - 1, B, d.H, dtype=torch.float32, device=dev)
        attn = torch.empty(T - 1, B, S, dtype=torch.float32, device=dev)
        lens = (C.c_int32 * B)(*lengths_host)
        w = eng.weights(pd)
        check(eng.lib.uic_nmt_forward_loss(C.byref(d), C.byref(w), ptr(src), lens, ptr(lengths_dev), ptr(tgt), int(model.training), seed,
                                           ptr(ws), ptr(loss), ptr(stats), ptr(outputs), ptr(attn), None, stream()), "nmt_forward_loss")
        ctx.model, ctx.ws, ctx.d = model, ws, d
        ctx.call = (lens, model.training, seed)
        ctx.inputs = (src, tgt)
        ctx.params = params
        ctx.mark_non_differentiable(outputs, attn, stats)
        return loss.view(()), outputs, attn, stats

    @staticmethod
    def backward(ctx, g_loss, *_):
        model = ctx.model
        eng = model.engine
        if ctx.ws is None:
            raise RuntimeError("NMT backward called twice: the workspace was recycled (retain_graph is not supported)")
        lens, training, seed = ctx.call
        src, tgt = ctx.inputs
        pd = dict(zip(model.param_names, ctx.params))
        sink = model.grad_sink
        direct = (sink is not None and model.unit_loss_gradient and all(pd[k].grad is sink[k] for k in model.param_names))
        # direct: every p.grad IS its view of the optimizer's flat gradient arena (Optim.zero_grad just zeroed it) and the
        # caller has DECLARED (model.unit_loss_gradient, set by Trainer.train_nmt) that the loss is backpropagated with
        # weight 1 (a plain loss.backward()), so the kernels write the arena in place -- no temporaries, no scale, no
        # accumulate pass over the 86 M parameters.  Any other caller (scaled losses, loss.backward(gradient=...)) takes
        # the general path below, which honours g_loss and lets autograd accumulate.
        lazy = getattr(model, '_lazy_zero', None)
        if not direct and lazy is not None:
            lazy()                       # (Optim.zero_grad(nmt_direct=True) left the arena as it was: clear it before autograd accumulates)
        model._lazy_zero = None
        if direct:
            if model._sink_written:
                raise RuntimeError("NMT backward ran twice into the optimizer's gradient arena without Optim.zero_grad() in between: "
                                   "the in-place path overwrites, it does not accumulate (clear model.unit_loss_gradient to accumulate)")
            model._sink_written = True
        grads = sink if direct else {k: torch.empty_like(v) for k, v in pd.items()}
        w, gw = eng.weights(pd), eng.weights(grads)
        check(eng.lib.uic_nmt_backward(C.byref(ctx.d), C.byref(w), ptr(src), lens, ptr(tgt), int(training), seed, ptr(ctx.ws),
                                       C.byref(gw), stream()), "nmt_backward")
        eng.release(ctx.d, ctx.ws)
        ctx.ws = None
        if direct:
            return (None,) * (6 + len(model.param_names))
        out = []
        for k in model.param_names:
            out.append(grads[k] * g_loss if model.scale_grads else grads[k])
        return (None, None, None, None, None, None) + tuple(out)


class NMTModel(nn.Module):
    """P/models/NMT_Models.py:272-420.  `generator` must be attached (P/trainer.py:85,89) before the first call."""

    def __init__(self, opt, encoder, decoder, src_dict, tgt_dict, multigpu=False):
        if multigpu:
            raise NotImplementedError("nn.DataParallel is not how this framework scales: run one process per GPU "
                                      "(parallel_exchange.GradientExchange) and pass multigpu=False")
        self.multigpu = multigpu
        super(NMTModel, self).__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.src_dict = src_dict
        self.tgt_dict = tgt_dict
        self.opt = opt
        self.compute_dtype = getattr(opt, 'compute_dtype', 'bf16')
        self.scale_grads = True
        self._engine = None
        self._seed_counter = int(getattr(opt, 'seed', 0) or 0) & 0x7FFFFFFF
        self._last_seed = None
        self.generator = None
        self.grad_sink = None      # set by misc.optimizer.Optim: {key: view of the flat gradient arena}
        self.unit_loss_gradient = False   # set by Trainer.train_nmt: the loss is backpropagated as is (weight 1)
        self._sink_written = False        # cleared by Optim.zero_grad

    # ---- engine plumbing
    @property
    def engine(self):
        if self._engine is None:
            self._engine = _NmtEngine(self)
        return self._engine

    def next_seed(self):
        self._seed_counter = (self._seed_counter * 1103515245 + 12345) & 0x7FFFFFFF
        self._last_seed = self._seed_counter
        return self._seed_counter

    @property
    def param_names(self):
        return [k for _, _, _, k in _lib.nmt_weight_keys(self.encoder.layers)]

    def _param_dict(self):
        if self.generator is None:
            raise RuntimeError("attach the generator first (model.generator = nn.Sequential(nn.Linear(rnn_size, V), nn.LogSoftmax()))")
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in self.param_names}

    # ---- reference surface
    def translateBatch(self, batch, beam_size=15, max_steps=100):
        """NMTModel.translateBatch (P/models/NMT_Models.py:322-395): beam-search translation of `batch.src` [S, B, 1] with
        the reference's hard-coded beam 15 / 100 steps.  Returns (allHyp, allScores, allAttn, goldScores) shaped like the
        reference's: per sentence a 1-best list of token-id lists (as long as the number of decoder steps taken -- cut
        them at EOS like buildTargetTokens does), its score, and its attention [steps, valid source positions]."""
        if self.training:
            raise NotImplementedError("translateBatch runs in eval mode")
        src = batch.src
        if src.dim() == 3:
            src = src[:, :, 0]
        src = src.contiguous()
        S, B = src.shape
        eng = self.engine
        d = eng.dims(B, S, 2)
        nbytes = eng.lib.uic_nmt_translate_workspace_bytes(C.byref(d), beam_size, max_steps)
        if nbytes == 0:
            check(-1, "uic_nmt_translate_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=src.device)
        hyp = torch.zeros(B, max_steps, dtype=torch.int64, device=src.device)
        scores = torch.zeros(B, dtype=torch.float32, device=src.device)
        attn = torch.zeros(B, max_steps, S, dtype=torch.float32, device=src.device)
        n_iter = C.c_int32(0)
        with torch.no_grad():
            w = eng.weights({k: v.detach() for k, v in self._param_dict().items()})
            check(eng.lib.uic_nmt_translate(C.byref(d), C.byref(w), ptr(src), beam_size, max_steps, ptr(ws), ptr(hyp), ptr(scores),
                                            ptr(attn), C.byref(n_iter), stream()), "nmt_translate")
        n = n_iter.value
        hyp_h, valid = hyp[:, :n].cpu(), (src != PAD).sum(0).cpu()
        allHyp = [[[int(t) for t in hyp_h[b]]] for b in range(B)]
        allScores = [scores[b:b + 1] for b in range(B)]
        allAttn = [[attn[b, :n, :int(valid[b])]] for b in range(B)]
        return allHyp, allScores, allAttn, scores.new_zeros(B)

    def forward(self, src, tgt, lengths, dec_state=None):
        """src [S,B,1] int64, tgt [T,B] int64, lengths [1,B] (sorted descending).  Returns (outputs [T-1,B,H],
        attns {'std': [T-1,B,S]}, dec_state=None, upper_bounds=None) like P/models/NMT_Models.py:414-420."""
        if dec_state is not None:
            raise NotImplementedError("a carried decoder state is never passed by the reference trainer (P/trainer.py:142,178)")
        if src.dim() == 3:
            if src.shape[2] != 1:
                raise NotImplementedError("source word features (nfeat > 1) are outside the hot path")
            src = src[:, :, 0]
        live = getattr(tgt, 'uic_live', None) if getattr(self.opt, 'live_positions', 1) else None
        src = src.contiguous()
        tgt = tgt.contiguous()
        lengths_host = [int(x) for x in lengths.reshape(-1).tolist()]
        lengths_dev = lengths.reshape(-1).to(device=src.device, dtype=torch.int32).contiguous()
        pd = self._param_dict()
        loss, outputs, attn, stats = _NmtStep.apply(self, src, tgt, lengths_host, lengths_dev, live, *[pd[k] for k in self.param_names])
        outputs.uic_loss = loss
        outputs.uic_stats = stats
        return outputs, {'std': attn}, None, None
