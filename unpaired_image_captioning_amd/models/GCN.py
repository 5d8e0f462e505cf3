"""Scene-graph GCN encoder of BASELINE configs[4]: object features + relation graph -> node features the attention-LSTM
captioner takes as `att_feats` (att_feat_size = gcn_hidden_size).  The reference tree holds NO GCN code -- the architecture is
this package's own statement (csrc/gcn.hip, include/uic_hip.h) and its parity is unpinned.  All arithmetic runs in
libuic_hip.so (no CPU / eager fallback).

    enc = SceneGraphEncoder(opt).cuda()
    nodes = enc(obj_feats, adj)         # [N, R, D] f32, [N, R, R] f32 (normalised relation graph, data) -> [N, R, H] f32
    seq_logp = captioner(nodes.mean(1)..., att_feats=nodes, ...)

opt: att_feat_size (D), gcn_hidden_size (H, default rnn_size), gcn_layers (1..3, default 2), compute_dtype.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, ptr, stream


class _GcnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, adj, *params):
        lib = _lib.load()
        d = module._dims(x.shape[0], x.shape[1])
        w = module._weights(params)
        ws = module._checkout(d, x.device)
        out = torch.empty(x.shape[0], x.shape[1], module.H, dtype=torch.float32, device=x.device)
        check(lib.uic_gcn_forward(C.byref(d), C.byref(w), ptr(x), ptr(adj), ptr(ws), ptr(out), stream()), "gcn_forward")
        ctx.module, ctx.d, ctx.ws, ctx.adj, ctx.need_dx = module, d, ws, adj, x.requires_grad
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        if ctx.ws is None:
            # the activations went back to the workspace pool after the first backward (they may already belong to another forward)
            raise RuntimeError("the scene-graph encoder's activations are released by its first backward: a second backward over the same "
                               "graph (retain_graph=True) needs a new forward" )
        params = ctx.saved_tensors
        module = ctx.module
        grads = [torch.empty_like(p) for p in params]
        w, g = module._weights(params), module._weights(grads)
        dx = torch.empty(ctx.d.N, ctx.d.R, ctx.d.D, dtype=torch.float32, device=dout.device) if ctx.need_dx else None
        check(lib.uic_gcn_backward(C.byref(ctx.d), C.byref(w), ptr(ctx.adj), ptr(ctx.ws), ptr(dout.contiguous().float()), C.byref(g),
                                   ptr(dx) if dx is not None else None, stream()), "gcn_backward")
        module._release(ctx.d, ctx.ws)
        ctx.ws = None
        return (None, dx, None) + tuple(grads)


class SceneGraphEncoder(nn.Module):
    def __init__(self, opt):
        super(SceneGraphEncoder, self).__init__()
        g = lambda k, dflt: getattr(opt, k, dflt) if getattr(opt, k, None) is not None else dflt
        self.D = int(g("att_feat_size", 2048))
        self.H = int(g("gcn_hidden_size", g("rnn_size", 512)))
        self.layers = int(g("gcn_layers", 2))
        self.dtype_id = _lib.dtype_id(g("compute_dtype", "bf16"))
        if not 1 <= self.layers <= _lib.GCN_MAX_LAYERS:
            raise NotImplementedError("gcn_layers=%d: 1..%d" % (self.layers, _lib.GCN_MAX_LAYERS))
        self.gcn = nn.ModuleList([nn.Linear(self.D if l == 0 else self.H, self.H) for l in range(self.layers)])
        self._ws = {}

    def _dims(self, N, R):
        d = _lib.GcnDims()
        d.dtype, d.N, d.R, d.D, d.H, d.layers = self.dtype_id, N, R, self.D, self.H, self.layers
        return d

    def _weights(self, tensors):
        w = _lib.GcnWeights()
        for l in range(self.layers):
            w.w[l] = ptr(tensors[2 * l])
            w.b[l] = ptr(tensors[2 * l + 1])
        return w

    # Workspaces are checked out per forward pass and go back to the pool when its backward has run (or at once for a
    # no-grad call): two forward passes of the same size before a backward -- D(real) and D(fake) summed into one loss --
    # must not share the activations they saved.  A pass whose backward never runs simply keeps its buffer out of the pool.
    def _checkout(self, d, device):
        free = self._ws.setdefault((d.N, d.R, str(device)), [])
        if free:
            return free.pop()
        nbytes = _lib.load().uic_gcn_workspace_bytes(C.byref(d))
        if nbytes == 0:
            check(-1, "uic_gcn_workspace_bytes")
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def _release(self, d, ws):
        if ws is not None:
            self._ws.setdefault((d.N, d.R, str(ws.device)), []).append(ws)

    def forward(self, obj_feats, adj):
        if not (obj_feats.is_cuda and adj.is_cuda):
            raise RuntimeError("SceneGraphEncoder runs on the MI355X only: there is no CPU fallback")
        if obj_feats.dim() != 3 or obj_feats.shape[2] != self.D or adj.shape != (obj_feats.shape[0], obj_feats.shape[1], obj_feats.shape[1]):
            raise ValueError("obj_feats [N, R, %d] and adj [N, R, R] expected" % self.D)
        ps = []
        for lin in self.gcn:
            ps += [lin.weight.contiguous(), lin.bias.contiguous()]
        return _GcnFn.apply(self, obj_feats.contiguous().float(), adj.contiguous().float(), *ps)
