"""CNN sentence discriminator of the unpaired / adversarial configuration (BASELINE configs[3]): D(caption) in (0, 1), trained
with BCE on real vs generated token rows; its score is a reward of the self-critical step.  The reference tree holds NO
discriminator code -- the architecture is this package's own statement (csrc/discriminator.hip, include/uic_hip.h) and its
parity is unpinned.  All arithmetic runs in libuic_hip.so (no CPU / eager fallback).

    D = SentenceDiscriminator(opt).cuda()
    logits = D(tokens)                      # [N] f32, differentiable w.r.t. D's parameters; tokens [N, >= L] int64
    loss = D.bce(logits, labels)            # BCEWithLogitsLoss (mean) on the device
    r = D.scores(tokens)                    # sigmoid(logit), eval mode, no grad

opt: vocab_size, seq_length, input_encoding_size (or disc_embed_size), disc_num_filters (128), disc_filter_sizes ((1, 2, 3, 4)),
disc_dropout (0.25), compute_dtype ('bf16' | 'f32').
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, ptr, stream


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, tokens, training, seed, *params):
        lib = _lib.load()
        d = module._dims(tokens.shape[0])
        w = module._weights(params)
        ws = module._checkout(d, tokens.device)
        logits = torch.empty(tokens.shape[0], dtype=torch.float32, device=tokens.device)
        check(lib.uic_disc_forward(C.byref(d), C.byref(w), ptr(tokens), tokens.shape[1], int(training), seed, ptr(ws), ptr(logits), None,
                                   stream()), "disc_forward")
        ctx.module, ctx.tokens, ctx.training, ctx.seed, ctx.ws, ctx.d = module, tokens, int(training), seed, ws, d
        ctx.save_for_backward(*params)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        if ctx.ws is None:
            # the activations went back to the workspace pool after the first backward (they may already belong to another forward)
            raise RuntimeError("the discriminator's activations are released by its first backward: a second backward over the same "
                               "graph (retain_graph=True) needs a new forward" )
        params = ctx.saved_tensors
        module = ctx.module
        grads = [torch.empty_like(p) for p in params]
        w, g = module._weights(params), module._weights(grads)
        dl = dlogits.contiguous().float()
        check(lib.uic_disc_backward(C.byref(ctx.d), C.byref(w), ptr(ctx.tokens), ctx.tokens.shape[1], ctx.training, ctx.seed, ptr(ctx.ws),
                                    ptr(dl), C.byref(g), stream()), "disc_backward")
        module._release(ctx.d, ctx.ws)
        ctx.ws = None
        return (None, None, None, None) + tuple(grads)


class SentenceDiscriminator(nn.Module):
    def __init__(self, opt):
        super(SentenceDiscriminator, self).__init__()
        g = lambda k, dflt: getattr(opt, k, dflt) if getattr(opt, k, None) is not None else dflt
        self.V1 = opt.vocab_size + 1
        self.L = opt.seq_length
        self.E = g("disc_embed_size", g("input_encoding_size", 512))
        self.F = g("disc_num_filters", 128)
        self.widths = tuple(int(w) for w in g("disc_filter_sizes", (1, 2, 3, 4)))
        self.drop_p = float(g("disc_dropout", 0.25))
        self.dtype_id = _lib.dtype_id(g("compute_dtype", "bf16"))
        self.seed = int(g("seed", 0)) & 0x7FFFFFFF
        if not (1 <= len(self.widths) <= _lib.DISC_MAX_WIDTHS and all(1 <= w <= 4 and w <= self.L for w in self.widths)):
            raise NotImplementedError("disc_filter_sizes %r: 1 to %d widths in [1, min(4, seq_length)]" % (self.widths, _lib.DISC_MAX_WIDTHS))
        Ft = self.F * len(self.widths)
        self.Ft = Ft
        self.embed = nn.Embedding(self.V1, self.E)
        for w in self.widths:
            conv = nn.Module()
            conv.weight = nn.Parameter(torch.empty(self.F, w, self.E).uniform_(-1, 1) * (1.0 / (w * self.E)) ** 0.5)
            conv.bias = nn.Parameter(torch.zeros(self.F))
            setattr(self, "conv%d" % w, conv)
        self.highway = nn.Linear(Ft, 2 * Ft)
        self.out = nn.Module()
        self.out.weight = nn.Parameter(torch.empty(Ft).uniform_(-1, 1) * (1.0 / Ft) ** 0.5)
        self.out.bias = nn.Parameter(torch.zeros(1))
        self._ws = {}
        self._calls = 0

    # ---- plumbing
    def _param_list(self):
        ps = [self.embed.weight]
        for w in self.widths:
            ps += [getattr(self, "conv%d" % w).weight, getattr(self, "conv%d" % w).bias]
        return ps + [self.highway.weight, self.highway.bias, self.out.weight, self.out.bias]

    def _dims(self, N):
        d = _lib.DiscDims()
        d.dtype, d.N, d.L, d.V1, d.E, d.F, d.nw = self.dtype_id, N, self.L, self.V1, self.E, self.F, len(self.widths)
        for i, w in enumerate(self.widths):
            d.widths[i] = w
        d.drop_p = self.drop_p
        return d

    def _weights(self, tensors):
        w = _lib.DiscWeights()
        it = iter(tensors)
        w.embed_w = ptr(next(it))
        for i in range(len(self.widths)):
            w.conv_w[i] = ptr(next(it))
            w.conv_b[i] = ptr(next(it))
        w.hw_w, w.hw_b, w.out_w, w.out_b = ptr(next(it)), ptr(next(it)), ptr(next(it)), ptr(next(it))
        return w

    # Workspaces are checked out per forward pass and go back to the pool when its backward has run (or at once for a
    # no-grad call): two forward passes of the same size before a backward -- D(real) and D(fake) summed into one loss --
    # must not share the activations they saved.  A pass whose backward never runs simply keeps its buffer out of the pool.
    def _checkout(self, d, device):
        free = self._ws.setdefault((d.N, str(device)), [])
        if free:
            return free.pop()
        nbytes = _lib.load().uic_disc_workspace_bytes(C.byref(d))
        if nbytes == 0:
            check(-1, "uic_disc_workspace_bytes")
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def _release(self, d, ws):
        if ws is not None:
            self._ws.setdefault((d.N, str(ws.device)), []).append(ws)

    def _check(self, tokens):
        if not tokens.is_cuda:
            raise RuntimeError("SentenceDiscriminator runs on the MI355X only: there is no CPU fallback")
        if tokens.dtype != torch.int64 or tokens.dim() != 2 or tokens.shape[1] < self.L:
            raise ValueError("tokens must be int64 [N, >= %d]" % self.L)
        return tokens.contiguous()

    # ---- surface
    def forward(self, tokens, seed=None):
        tokens = self._check(tokens)
        ps = [p if p.is_contiguous() else p.contiguous() for p in self._param_list()]
        if seed is None:
            self._calls += 1
            seed = (self.seed * 0x9E3779B1 + self._calls) & 0x7FFFFFFF
        return _DiscFn.apply(self, tokens, self.training, int(seed), *ps)

    def scores(self, tokens):
        """sigmoid(logit) without dropout and without autograd (the reward of the generator's self-critical step)."""
        tokens = self._check(tokens)
        lib = _lib.load()
        with torch.no_grad():
            d = self._dims(tokens.shape[0])
            w = self._weights([p.contiguous() for p in self._param_list()])
            ws = self._checkout(d, tokens.device)
            prob = torch.empty(tokens.shape[0], dtype=torch.float32, device=tokens.device)
            check(lib.uic_disc_forward(C.byref(d), C.byref(w), ptr(tokens), tokens.shape[1], 0, 0, ptr(ws), None, ptr(prob), stream()),
                  "disc_forward")
            self._release(d, ws)               # (stream order: whoever takes it next runs after this call on the stream)
        return prob

    @staticmethod
    def bce(logits, labels):
        """BCEWithLogitsLoss(mean) of [N] logits against [N] float labels, loss and gradient from one device kernel."""
        return _BceFn.apply(logits, labels)


class _BceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        lib = _lib.load()
        lg, lb = logits.contiguous().float(), labels.contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=lg.device)
        dl = torch.empty_like(lg)
        check(lib.uic_disc_bce(ptr(lg), ptr(lb), lg.numel(), ptr(loss), ptr(dl), stream()), "disc_bce")
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None
