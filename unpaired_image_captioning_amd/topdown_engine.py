"""Device-side driver of the TopDown captioner: owns the caller-allocated arenas
(workspace, derived weights) the C ABI asks for and turns torch tensors into the
pointer structs of include/uic_hip.h.  torch is plumbing only (device memory,
streams); all arithmetic happens in libuic_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Batch, Dims, Gathered, Weights, check, ptr, stream, weight_fields


class Workspace(object):
    """One forward's activations; held until its backward has run."""

    def __init__(self, dims, buf):
        self.dims = dims
        self.buf = buf
        self.released = (None, None)     # (event, stream) of the last release: see TopDownEngine.checkout


class TopDownEngine(object):
    def __init__(self, sizes, dtype="bf16", drop_p=0.5, use_bn=0, logit_layers=1):
        """sizes: dict with V1, E, H, A, D, Dfc.  use_bn: opt.use_bn; the BatchNorm running statistics (buffers, not
        parameters) are looked up in `self.buffers`, which the owning model keeps pointed at its live tensors."""
        self.use_bn = int(use_bn)
        self.logit_layers = int(logit_layers)
        self.buffers = {}
        self.lib = _lib.load()
        self.sizes = dict(sizes)
        # att_feat_size that is not a multiple of 8 (2048 + 5 box features with the reference's default use_box = 1): the
        # library works on zero-padded feature rows / weight columns, which is exact (see _pad_for_library)
        self.D = int(self.sizes["D"])
        # (padded to a multiple of 128, not just 8, so that the LDS-DMA and transposing-read GEMMs stay eligible)
        self.Dp = self.D if self.D % 8 == 0 else (self.D + 127) // 128 * 128
        self.sizes["D"] = self.Dp
        self.dtype = _lib.dtype_id(dtype)
        self.drop_p = float(drop_p)
        self._derived = None
        self._derived_key = None
        self._pool = {}
        self.seed = 0x5EED
        # uic_topdown_dims.recurrence (_lib.REC_*): how the decode loop and its BPTT are launched; 0 = the library's default
        # (persistent forward recurrence where the shapes allow, per-step BPTT launches)
        self.recurrence = 0
        # sharded data parallelism (trainer._GatheredWeights): the operand-dtype weights come from the all-gathered arena, each
        # gather group behind its event (uic_topdown_refresh_weights_gathered); None = cast from the f32 masters
        self.gathered = None

    # ------------------------------------------------------------------ arenas
    def dims(self, N, R, T, seq_per_img=1):
        s = self.sizes
        return Dims(N=N, R=R, D=s["D"], Dfc=s["Dfc"], H=s["H"], E=s["E"], A=s["A"], V1=s["V1"], T=T,
                    dtype=self.dtype, drop_p=self.drop_p, use_bn=self.use_bn, seq_per_img=seq_per_img,
                    logit_layers=self.logit_layers, recurrence=int(self.recurrence),
                    rnn_status=_lib.status_words().data_ptr() if torch.cuda.is_available() else None)

    @staticmethod
    def _key(d):
        return (d.N, d.R, d.T, d.dtype, d.seq_per_img)

    @staticmethod
    def _rows(att, labels):
        """(caption rows N, seq_per_img S): features may come once per image ([N / S, ...]) with labels per caption row --
        the loader's S-fold replication (P/misc/dataloader/dataloader.py:270-277) then happens on the device."""
        N = labels.shape[0]
        n_feat = att.shape[0]
        if N % n_feat != 0:
            raise ValueError("labels have %d rows, features %d: not a whole number of captions per image" % (N, n_feat))
        return N, N // n_feat

    def checkout(self, d, device):
        free = self._pool.setdefault(self._key(d), [])
        # release() happens at ENQUEUE time: the kernels that use a pooled buffer may still be running on the stream that
        # released it (the self-critical step's greedy baseline runs on a second stream beside the sampling pass, and with
        # one caption per image both passes ask for the same dims).  A buffer is handed out again only to the stream that
        # released it (stream order protects it) or once that stream has passed the release point; otherwise the pool grows.
        cur = torch.cuda.current_stream().cuda_stream
        for i, ws in enumerate(free):
            ev, sid = ws.released
            if ev is None or sid == cur or ev.query():
                free.pop(i)
                ws.released = (None, None)
                return ws
        nbytes = self.lib.uic_topdown_workspace_bytes(C.byref(d))
        if nbytes == 0:
            check(-1, "uic_topdown_workspace_bytes")
        return Workspace(d, torch.empty(nbytes, dtype=torch.uint8, device=device))

    def release(self, ws):
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        ws.released = (ev, cur.cuda_stream)
        self._pool.setdefault(self._key(ws.dims), []).append(ws)

    # ---- att_feat_size % 8 != 0: zero feature columns meet zero weight columns, a BatchNorm over an all-zero column gives
    # xhat = 0 (mean 0, var 0), so padded results and gradients equal the unpadded ones; the padded columns of every
    # output (gradients, running statistics) are dropped on the way back
    _PAD_FILL = {"weight": 1.0, "bias": 0.0, "running_mean": 0.0, "running_var": 1.0}

    def _pad_for_library(self, key, t, as_output):
        """Padded stand-in of tensor `key` when it has an att_feat_size axis, else None."""
        if self.Dp == self.D:
            return None
        lin = "att_embed.1.weight" if self.use_bn else "att_embed.0.weight"
        if key == lin:
            tmp = t.new_zeros(t.shape[0], self.Dp)
        elif self.use_bn and key.startswith("att_embed.0."):
            tmp = t.new_full((self.Dp,), 0.0 if as_output else self._PAD_FILL[key.rsplit(".", 1)[1]])
        else:
            return None
        if not as_output:
            tmp[..., :self.D].copy_(t)
        return tmp

    def _pad_att(self, att):
        if att is None or self.Dp == self.D:
            return att
        if att.shape[-1] != self.D:
            raise ValueError("att_feats have %d columns, the model was built for att_feat_size=%d" % (att.shape[-1], self.D))
        if (getattr(att, "_uic_zero_padded_ld", 0) == self.Dp and att.dim() == 3 and
                att.stride() == (att.shape[1] * self.Dp, self.Dp, 1)):
            # the loader's assembly kernel wrote the rows at the padded stride already, padding zero-filled
            return att.as_strided((att.shape[0], att.shape[1], self.Dp), att.stride())
        return torch.nn.functional.pad(att, (0, self.Dp - self.D)).contiguous()

    def _write_back(self, w):
        """Copy what the library wrote into padded stand-ins (gradients, BatchNorm running statistics) to the real tensors."""
        for tmp, orig in getattr(w, "_writeback", ()):
            orig.copy_(tmp[..., :self.D])

    def weights_struct(self, tensors, outputs=False):
        """tensors: dict reference-state_dict-key -> contiguous f32 device tensor.  outputs: the struct will be WRITTEN by
        the library (a gradient struct): padded stand-ins start as zeros and are copied back by _write_back."""
        w = Weights()
        w._keep, w._writeback = [], []
        for field, key, is_param in weight_fields(self.use_bn, self.logit_layers):
            t = tensors.get(key) if (is_param or outputs) else tensors.get(key, self.buffers.get(key))
            if t is None:
                if is_param:
                    raise KeyError(key)
                continue                      # gradient structs carry no running statistics
            if t.dtype != torch.float32:
                raise RuntimeError("parameter %s must be float32, got %s" % (key, t.dtype))
            tmp = self._pad_for_library(key, t, outputs)
            if tmp is not None:
                w._keep.append(tmp)
                if outputs or not is_param:   # gradients; running statistics (updated in place by a train-mode forward)
                    w._writeback.append((tmp, t))
                t = tmp
            if ":" in field:                  # element of a pointer array (hidden logit blocks)
                name, idx = field.split(":")
                getattr(w, name)[int(idx)] = ptr(t)
            else:
                setattr(w, field, ptr(t))
        return w

    def hold_weights(self):
        """Context manager: the caller promises that the parameters do not change inside (e.g. the sampling pass, the greedy
        baseline and the replay of one self-critical step), so the operand-dtype / transposed weight copies are rebuilt by
        the first call only instead of by every call."""
        eng = self

        class _Hold(object):
            def __enter__(self):
                eng._hold = {"fresh": False}
                return eng

            def __exit__(self, *a):
                eng._hold = None
                return False
        return _Hold()

    def refresh(self, params, d, defer=False):
        """Rebuild operand-dtype / transposed weight copies from the f32 masters.  defer: the caller's very next library call
        consumes them (xe_train_step), so the transposes may be left to that call (uic_topdown_refresh_weights_deferred)."""
        hold = getattr(self, "_hold", None)
        if hold is not None and hold["fresh"] and self._derived is not None:
            return self.weights_struct(params)          # same masters as the call that refreshed: pointers only
        if hold is not None:
            hold["fresh"] = True
        dev = next(iter(params.values())).device
        nbytes = self.lib.uic_topdown_derived_bytes(C.byref(d))
        if self._derived is None or self._derived.numel() < nbytes or self._derived.device != dev:
            self._derived = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        w = self.weights_struct(params)
        gw = self.gathered
        if gw is not None:
            g = Gathered()
            g._keep = []
            for key, (field, _grp) in gw.keys.items():
                t = gw.views[key]
                if ":" in field:
                    name, idx = field.split(":")
                    getattr(g, name)[int(idx)] = ptr(t)
                else:
                    setattr(g, field, ptr(t))
            for grp, ev in enumerate(gw.events):
                g.ready[grp] = ev.cuda_event if ev is not None else None
            g._keep = list(gw.events)
            check(self.lib.uic_topdown_refresh_weights_gathered(C.byref(d), C.byref(w), C.byref(g), ptr(self._derived), int(bool(defer)),
                                                                stream()), "refresh_weights_gathered")
            w._gathered = g                # (the library keeps the late groups' events until the consumer runs: keep them alive)
            return w
        fn = self.lib.uic_topdown_refresh_weights_deferred if defer else self.lib.uic_topdown_refresh_weights
        check(fn(C.byref(d), C.byref(w), ptr(self._derived), stream()), "refresh_weights")
        return w

    def input_grad_buffers(self, fc, att, want_fc=True, want_att=True):
        """Output tensors for Batch.d_fc_feats / d_att_feats (att at the padded row width; slice [..., :D] afterwards)."""
        d_fc = torch.empty(fc.shape, dtype=torch.float32, device=fc.device) if want_fc else None
        d_att = torch.empty(att.shape[:-1] + (self.Dp,), dtype=torch.float32, device=att.device) if want_att else None
        return d_fc, d_att

    def batch_struct(self, fc, att, att_masks, labels=None, masks=None, grad_scale=None, ss_prob=0.0, d_fc=None, d_att=None, live=None):
        """live: (live_rows, live_count) -- uic_topdown_batch.live_rows / live_count: the per-step counts of unmasked positions (host
        int32 numpy, live_counts(masks)) and, optionally, the list itself (device int32, live_positions(masks)); with live_rows
        None the step compacts the masks on the device."""
        b = Batch()
        if live is not None and live[1] is not None:
            rows, count = live
            count = np.ascontiguousarray(count, dtype=np.int32)
            T = (labels.shape[1] - 1) if labels is not None else 0
            if count.shape != (T,):
                raise ValueError("live_count must hold one entry per decode step (%d)" % T)
            if rows is not None:
                if rows.dtype != torch.int32 or not rows.is_contiguous() or rows.device != fc.device:
                    raise ValueError("live_rows must be a contiguous int32 tensor on the batch's device")
                if rows.numel() != ((int(count.sum()) + 127) // 128) * 128:
                    raise ValueError("live_rows must hold roundup(sum(live_count), 128) entries")
                b.live_rows = rows.data_ptr()
            b._keep_live = (rows, count)
            b.live_count = count.ctypes.data_as(C.POINTER(C.c_int32))
        att = self._pad_att(att)
        b._keep = att                          # (a padded copy must outlive the call)
        b.d_fc_feats = ptr(d_fc)
        b.d_att_feats = ptr(d_att)
        b.fc_feats = ptr(fc)
        b.att_feats = ptr(att)
        b.att_masks = ptr(att_masks)
        b.labels = ptr(labels)
        b.ld_labels = labels.shape[1] if labels is not None else 0
        b.masks = ptr(masks)
        b.ld_masks = masks.shape[1] if masks is not None else 0
        b.grad_scale = ptr(grad_scale)
        b.ld_grad_scale = grad_scale.shape[1] if grad_scale is not None else 0
        b.ss_prob = float(ss_prob)
        return b

    # ------------------------------------------------------------------ calls
    def forward(self, params, fc, att, att_masks, labels, t_run, training, seed, want_logprobs=True, masks=None, ss_prob=0.0):
        (N, S), R = self._rows(att, labels), att.shape[1]
        T = labels.shape[1] - 1
        d = self.dims(N, R, T, S)
        w = self.refresh(params, d)
        ws = self.checkout(d, fc.device)
        b = self.batch_struct(fc, att, att_masks, labels, masks, ss_prob=ss_prob)
        logp = torch.zeros(N, T, d.V1, dtype=torch.float32, device=fc.device) if want_logprobs else None
        check(self.lib.uic_topdown_forward(C.byref(d), C.byref(w), ptr(self._derived), C.byref(b), t_run,
                                           int(training), seed & 0xFFFFFFFF, ptr(ws.buf), ptr(logp), stream()), "forward")
        self._write_back(w)
        return logp, ws, (d, w, b)

    def xe_loss(self, ws, d, b, t_run, inv_den=None):
        out = torch.empty(2, dtype=torch.float32, device=ws.buf.device)
        check(self.lib.uic_topdown_xe_loss(C.byref(d), C.byref(b), t_run, ptr(ws.buf), ptr(inv_den),
                                           out.data_ptr(), out.data_ptr() + 4, stream()), "xe_loss")
        return out            # [loss, sum(mask)]

    def backward(self, ws, d, w, b, t_run, training, seed, grads, dlogprobs=None, logprobs=None):
        g = self.weights_struct(grads, outputs=True)
        check(self.lib.uic_topdown_backward(C.byref(d), C.byref(w), ptr(self._derived), C.byref(b), t_run, int(training),
                                            seed & 0xFFFFFFFF, ptr(ws.buf), ptr(dlogprobs), ptr(logprobs), C.byref(g),
                                            stream()), "backward")
        self._write_back(g)

    def xe_train_step(self, params, fc, att, att_masks, labels, masks, t_run, training, seed, grads, inv_den=None,
                      grad_scale=None, ss_prob=0.0, keep_workspace=False, d_fc=None, d_att=None, resume_ws=None, out=None, live=None):
        """Fused forward + criterion + backward on two HIP streams; returns a device tensor [loss, sum(mask)].
        live: live_positions(masks) -- the logit layer and the criterion then run over the unmasked positions only.
        resume_ws: the workspace sample(..., keep_forward=True) left behind for these labels (same weights, seed, L): the
        step starts at the criterion (training bit 2) and the workspace is released afterwards."""
        (N, S), R = self._rows(att, labels), att.shape[1]
        T = labels.shape[1] - 1
        d = self.dims(N, R, T, S)
        w = self.refresh(params, d, defer=True)
        if resume_ws is not None:
            ws = resume_ws
            training = int(training) | 4
        else:
            ws = self.checkout(d, fc.device)
        b = self.batch_struct(fc, att, att_masks, labels, masks, grad_scale, ss_prob, d_fc=d_fc, d_att=d_att, live=live)
        g = self.weights_struct(grads, outputs=True)
        if out is None:
            out = torch.empty(2, dtype=torch.float32, device=fc.device)
        try:
            check(self.lib.uic_topdown_xe_train_step(C.byref(d), C.byref(w), ptr(self._derived), C.byref(b), t_run,
                                                     int(training), seed & 0xFFFFFFFF, ptr(ws.buf), ptr(inv_den),
                                                     out.data_ptr(), out.data_ptr() + 4, C.byref(g), stream()),
                  "xe_train_step")
            self._write_back(w)
            self._write_back(g)
        finally:
            if not keep_workspace:
                self.release(ws)
        if keep_workspace:
            return out, ws
        return out

    def sample(self, params, fc, att, att_masks, L, sample_max=1, temperature=1.0, decoding_constraint=0, seed=0,
               forced=None, training=False, seq_per_img=1, keep_forward=False):
        """seq_per_img = S > 1: features come once per image and S captions are decoded per image (rows image * S + j).
        keep_forward: run the pass in the training layout (uic_topdown_sample_train) and return (seq, lp, ws) with the
        workspace still checked out -- it holds the whole forward pass for xe_train_step(..., resume_ws=ws)."""
        N, R = att.shape[0] * seq_per_img, att.shape[1]
        d = self.dims(N, R, L + 1, seq_per_img)
        w = self.refresh(params, d)
        ws = self.checkout(d, fc.device)
        b = self.batch_struct(fc, att, att_masks)
        seq = torch.zeros(N, L, dtype=torch.int64, device=fc.device)
        lp = torch.zeros(N, L, dtype=torch.float32, device=fc.device)
        fn = self.lib.uic_topdown_sample_train if keep_forward else self.lib.uic_topdown_sample
        ok = False
        try:
            check(fn(C.byref(d), C.byref(w), ptr(self._derived), C.byref(b), L, int(sample_max),
                     float(temperature), int(decoding_constraint), seed & 0xFFFFFFFF, ptr(forced),
                     int(training), ptr(ws.buf), ptr(seq), ptr(lp), stream()), "sample")
            self._write_back(w)
            ok = True
        finally:
            if not (keep_forward and ok):
                self.release(ws)
        if keep_forward:
            return seq, lp, ws
        return seq, lp

    def sample_beam(self, params, fc, att, att_masks, L, beam_size, decoding_constraint=0, max_ppl=0, done_lists=False):
        """Beam search for all images at once: rows = (image, beam); the beam_size-fold replication of every image's
        features (AttModel.py:180-184) happens on the device (dims.seq_per_img = beam_size).  done_lists: also return the
        whole done list of every image (count [n_img], p [n_img, L*B], seq [n_img, L*B, L], logps likewise)."""
        n_img = att.shape[0]
        d = self.dims(n_img * beam_size, att.shape[1], L + 1, beam_size)
        w = self.refresh(params, d)
        ws = self.checkout(d, fc.device)
        b = self.batch_struct(fc, att, att_masks)
        seq = torch.zeros(n_img, L, dtype=torch.int64, device=fc.device)
        lp = torch.zeros(n_img, L, dtype=torch.float32, device=fc.device)
        lists = None
        try:
            check(self.lib.uic_topdown_sample_beam(C.byref(d), C.byref(w), ptr(self._derived), C.byref(b), L, int(beam_size),
                                                   int(decoding_constraint), int(max_ppl), ptr(ws.buf), ptr(seq), ptr(lp), stream()),
                  "sample_beam")
            if done_lists:
                LB = L * beam_size
                cnt = torch.zeros(n_img, dtype=torch.int32, device=fc.device)
                dp = torch.zeros(n_img, LB, dtype=torch.float32, device=fc.device)
                dseq = torch.zeros(n_img, LB, L, dtype=torch.int64, device=fc.device)
                dlp = torch.zeros(n_img, LB, L, dtype=torch.float32, device=fc.device)
                check(self.lib.uic_topdown_beam_done_lists(C.byref(d), ptr(ws.buf), L, int(beam_size), ptr(cnt), ptr(dp), ptr(dseq),
                                                           ptr(dlp), stream()), "beam_done_lists")
                lists = (cnt, dp, dseq, dlp)
        finally:
            self.release(ws)
        if done_lists:
            return seq, lp, lists
        return seq, lp

    def prepare_feature(self, params, fc, att, att_masks, training=False, seed=0):
        """AttModel._prepare_feature: (fc', att', p_att) as f32 tensors."""
        N, R = att.shape[0], att.shape[1]
        d = self.dims(N, R, 2)
        w = self.refresh(params, d)
        ws = self.checkout(d, fc.device)
        b = self.batch_struct(fc, att, att_masks)
        s = self.sizes
        fc_o = torch.empty(N, s["H"], dtype=torch.float32, device=fc.device)
        att_o = torch.empty(N, R, s["H"], dtype=torch.float32, device=fc.device)
        patt_o = torch.empty(N, R, s["A"], dtype=torch.float32, device=fc.device)
        try:
            check(self.lib.uic_topdown_prepare_feature(C.byref(d), C.byref(w), ptr(self._derived), C.byref(b), int(training),
                                                       seed & 0xFFFFFFFF, ptr(ws.buf), ptr(fc_o), ptr(att_o), ptr(patt_o), stream()),
                  "prepare_feature")
            self._write_back(w)
        finally:
            self.release(ws)
        return fc_o, att_o, patt_o

    def logprobs_state(self, params, it, fc, att, p_att, att_masks, h, c, t=0, training=False, seed=0):
        """AttModel.get_logprobs_state: one decode step; returns (logprobs [N, V1], h' [2, N, H], c' [2, N, H])."""
        N, R = att.shape[0], att.shape[1]
        d = self.dims(N, R, 2)
        w = self.refresh(params, d)
        ws = self.checkout(d, fc.device)
        s = self.sizes
        logp = torch.empty(N, s["V1"], dtype=torch.float32, device=fc.device)
        h2, c2 = torch.empty_like(h), torch.empty_like(c)
        try:
            check(self.lib.uic_topdown_logprobs_state(C.byref(d), C.byref(w), ptr(self._derived), ptr(it), ptr(fc), ptr(att), ptr(p_att),
                                                      ptr(att_masks), ptr(h), ptr(c), int(t), int(training), seed & 0xFFFFFFFF,
                                                      ptr(ws.buf), ptr(logp), ptr(h2), ptr(c2), stream()), "logprobs_state")
        finally:
            self.release(ws)
        return logp, h2, c2

    def workspace_tensor(self, ws, name, shape, dtype):
        """View of a named activation inside a workspace (tests)."""
        p = self.lib.uic_topdown_workspace_ptr(C.byref(ws.dims), ptr(ws.buf), name.encode())
        if not p:
            raise KeyError(name)
        offset = p - ws.buf.data_ptr()
        n = 1
        for s in shape:
            n *= s
        esz = torch.empty(0, dtype=dtype).element_size()
        return ws.buf[offset:offset + n * esz].view(dtype).view(*shape)


def live_counts(masks):
    """Per decode step, the number of positions whose mask is not zero (uic_topdown_batch.live_count): masks [N, T + 1] host
    array (the loader makes it on the host, P/misc/dataloader/dataloader.py:200-203) -> int32 numpy [T]."""
    m = masks.detach().cpu().numpy() if torch.is_tensor(masks) else np.asarray(masks)
    return np.count_nonzero(m[:, 1:], axis=0).astype(np.int32)


def live_positions(masks, device=None):
    """The list of live positions of a batch (uic_topdown_batch.live_rows / live_count, include/uic_hip.h): position (t, n) is
    live when masks[n, 1 + t] != 0 -- LanguageModelCriterion multiplies every other one by zero (P/misc/utils.py:62-73).
    masks: [N, T + 1] host array or tensor (the loader makes it on the host, P/misc/dataloader/dataloader.py:200-203).
    Returns (live_rows: int32 tensor on `device`, live_count: int32 numpy [T])."""
    if torch.is_tensor(masks):
        device = device if device is not None else masks.device
        m = masks.detach().cpu().numpy()
    else:
        m = np.asarray(masks)
    live = np.ascontiguousarray(m[:, 1:].T != 0)                     # [T, N], step-major
    count = live.sum(1).astype(np.int32)
    flat = np.flatnonzero(live).astype(np.int32)                     # t * N + n, ascending = step-major
    rows = np.full(((flat.size + 127) // 128) * 128, -1, dtype=np.int32)
    rows[:flat.size] = flat
    t = torch.from_numpy(rows)
    if device is not None and torch.device(device).type != "cpu":
        t = t.to(device, non_blocking=False)
    return t, count
