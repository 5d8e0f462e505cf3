"""One training step of the captioner, as the reference's Trainer.train does it
(P/trainer.py:141-193): host batch -> device, forward, LanguageModelCriterion, backward,
Adam (P/misc/optimizer.py:89-93; the captioner's gradient clipping is a no-op there because
`i2t_params` is a consumed generator, :78-79,92, and is therefore not applied here either).

MI355X-first differences (results-identical): log_softmax + criterion are fused over all T*N
rows (no [N,T,V1] log-prob tensor), parameters / gradients / Adam moments live in flat f32
arenas so the optimizer is one kernel and data parallelism is ONE RCCL all-reduce per step
(instead of DataParallel's per-step broadcast + gather + reduce, P/trainer.py:74).
"""
import os

import numpy as np
import torch

from . import _lib, models
from ._lib import check, ptr, stream
from .topdown_engine import live_counts
from .misc.optimizer import FlatArena, Optim  # noqa: F401
from .parallel_exchange import GradientExchange


def _steps_from_host_labels(labels_np):
    """Early break of AttModel._forward (:151) decided on the host copy of the labels: no device sync."""
    T = labels_np.shape[1] - 1
    colsum = labels_np[:, 1:T].sum(0)
    z = np.nonzero(colsum == 0)[0]
    return int(z[0]) + 1 if z.size else T


def xe_step(model, batch, t_run=None, inv_den=None, grads=None, return_seed=False, fused=True, d_fc=None, d_att=None, out=None):
    """forward + fused criterion + backward on device tensors.  Returns (loss[device scalar], grads dict).
    fused=True: one library call scheduled on two HIP streams; False: the three separate calls.
    d_fc / d_att (eng.input_grad_buffers): optional outputs for the gradients w.r.t. the input features (an encoder in
    front of the captioner; the reference's features are data).  out (fused only): a 2-float device tensor that receives
    [loss, sum(mask)] instead of a fresh one (the sharded exchange hands over scalar slots of its gradient arena)."""
    eng = model.engine
    labels = batch["labels"]
    if t_run is None:
        t_run = model._steps_to_run(labels)
    pd = {k: v.detach() for k, v in model.param_dict().items()}
    seed = model.next_seed()
    training = model.training
    ss_prob = float(getattr(model, 'ss_prob', 0.0)) if training else 0.0      # scheduled sampling, AttModel.py:130
    if hasattr(model, '_bn_count_batch'):
        model._bn_count_batch()
    if fused:
        if grads is None:
            grads = {k: torch.empty_like(v) for k, v in pd.items()}
        live = (batch.get("live_rows"), batch["live_count"]) if batch.get("live_count") is not None else None
        out = eng.xe_train_step(pd, batch["fc_feats"], batch["att_feats"], batch.get("att_masks"), labels, batch["masks"],
                                t_run, training, seed, grads, inv_den, ss_prob=ss_prob, d_fc=d_fc, d_att=d_att, out=out, live=live)
        if return_seed:
            return out[0], grads, seed
        return out[0], grads
    _, ws, (d, w, b) = eng.forward(pd, batch["fc_feats"], batch["att_feats"], batch.get("att_masks"), labels, t_run,
                                   training, seed, want_logprobs=False, masks=batch["masks"], ss_prob=ss_prob)
    out = eng.xe_loss(ws, d, b, t_run, inv_den)
    if grads is None:
        grads = {k: torch.empty_like(v) for k, v in pd.items()}
    if d_fc is not None or d_att is not None:
        b = eng.batch_struct(batch["fc_feats"], batch["att_feats"], batch.get("att_masks"), labels, batch["masks"], ss_prob=ss_prob,
                             d_fc=d_fc, d_att=d_att)
    eng.backward(ws, d, w, b, t_run, training, seed, grads)
    eng.release(ws)
    if return_seed:
        return out[0], grads, seed
    return out[0], grads


class _GatheredWeights(object):
    """What TopDownEngine.refresh hands uic_topdown_refresh_weights_gathered: the arena's all-gathered operand-dtype views by
    reference key, and per gather group of the C struct the event behind the all-gather that completed it."""

    def __init__(self, arena, keys):
        self.arena = arena
        self.keys = keys                                  # {reference key: (struct field, C gather group)}
        self.views = {k: arena.gathered_views[k] for k in keys}
        self.piece_of = {}
        for i, names in enumerate(arena.piece_names):
            for k in names:
                self.piece_of[k] = i
        self.events = [None] * 4

    def set_events(self, piece_events):
        """piece_events: {arena piece index: event recorded behind its all-gather}, in the order the gathers were enqueued on the
        communication stream: the event of the LAST-gathered piece a group needs covers its others."""
        order = {i: n for n, i in enumerate(piece_events)}
        ev = [None] * 4
        for k, (_, grp) in self.keys.items():
            i = self.piece_of[k]
            if ev[grp] is None or order[i] > order[ev[grp]]:
                ev[grp] = i
        self.events = [piece_events[i] if i is not None else None for i in ev]


class Trainer(object):
    """The reference Trainer's two halves: captioner (i2t: build, XE / self-critical step, save) and, after
    `build_nmt`, the pivot NMT teacher (P/trainer.py:56-58: each half is built only when its flag asks for it)."""

    def __init__(self, opt, exchange=None):
        self.opt = opt
        self.i2t_train_flag = getattr(opt, 'i2t_train_flag', 1)
        self.i2t_model = models.setup(opt) if getattr(opt, 'caption_model', None) else None
        self.dp_i2t_model = self.i2t_model
        if self.i2t_model is not None:
            self.i2t_model.train(bool(self.i2t_train_flag))
        self.nmt_model = None
        self.i2t_train_loss = 0.0
        self.sc_flag = False
        self.exchange = exchange if exchange is not None else GradientExchange()
        if self.i2t_model is not None:
            self._mix_rank_into_seed(self.i2t_model, self.exchange)
            eng = getattr(self.i2t_model, 'engine', None)
            if eng is not None and hasattr(self.exchange, 'ranks_share_a_device') and self.exchange.ranks_share_a_device():
                eng.recurrence |= _lib.REC_FWD_CHAIN       # several ranks on ONE GPU: per-step launches (include/uic_hip.h)
            if eng is not None and self.exchange.world_size > 1:
                # the overlapped exchange keeps a communication queue busy beside the step: the step then stays on two hardware
                # queues of its own (a fourth busy queue slows every dependent launch of the BPTT loop, include/uic_hip.h)
                eng.recurrence |= _lib.REC_COMM_STREAM
        if self.exchange.world_size > 1:
            self._require_two_hw_queues(opt)
        self.lr = getattr(opt, 'i2t_learning_rate', 4e-4)
        self.i2t_current_lr = self.lr
        self.betas = (getattr(opt, 'i2t_optim_alpha', 0.9), getattr(opt, 'i2t_optim_beta', 0.999))
        self.eps = getattr(opt, 'i2t_optim_epsilon', 1e-8)
        if getattr(opt, 'i2t_optim', 'adam') != 'adam':
            raise NotImplementedError("only Adam is on the MI355X hot path (i2t_optim=%s)" % opt.i2t_optim)
        if getattr(opt, 'i2t_weight_decay', 0):
            raise NotImplementedError("i2t_weight_decay != 0 is not on the MI355X hot path")
        self._step = 0
        self.arena = None
        self.last_loss = None

    @staticmethod
    def _require_two_hw_queues(opt):
        """A data-parallel process has more HIP streams than the four hardware queues ROCclr opens by default (the step's three,
        the communication stream, torch.distributed's own), and on MI355X a process with more than four BUSY hardware queues
        dispatches every dependent launch 1.5-2.5x slower (profiles/r05_v4_queue_probe.txt).  GPU_MAX_HW_QUEUES=2 multiplexes the
        streams onto two queues and removes the effect at no cost.  The HIP runtime reads it when it initialises: it is set here
        when that has not happened yet, and a process that comes too late is told so LOUDLY instead of silently running at half
        speed (opt.allow_many_hw_queues = 1 turns the error into a warning)."""
        val = os.environ.get("GPU_MAX_HW_QUEUES")
        if val is not None and val.strip().isdigit() and int(val) <= 2:
            return
        if val is None and not torch.cuda.is_initialized():
            os.environ["GPU_MAX_HW_QUEUES"] = "2"
            return
        msg = ("data-parallel Trainer: GPU_MAX_HW_QUEUES is %s but the HIP runtime is already initialised -- export GPU_MAX_HW_QUEUES=2 "
               "before the first HIP call (before `import torch` is safest), or the BPTT loop's launches dispatch 1.5-2.5x slower "
               "beside the communication stream (DESIGN.md section 6)" % ("unset" if val is None else "'%s'" % val))
        if getattr(opt, 'allow_many_hw_queues', 0) or (val is not None and not (val.strip().isdigit() and int(val) > 2)):
            import warnings
            warnings.warn(msg)
        else:
            raise RuntimeError(msg + "; opt.allow_many_hw_queues = 1 overrides")

    @staticmethod
    def _mix_rank_into_seed(model, exchange):
        """Dropout masks, multinomial draws and scheduled-sampling decisions are hashes of (seed, site, LOCAL row index): under
        data parallelism every rank must start from a different seed, or all shards share their noise."""
        rank = exchange.rank if exchange is not None else 0
        if rank and hasattr(model, '_seed_counter'):
            model._seed_counter = (model._seed_counter + 0x9E3779B1 * rank) & 0x7FFFFFFF

    # flat-arena order [FIRST_GRADS | LSTM_W_GRADS | rest | LATE_GRADS] = gradient groups 0 / 1 / 2 / tail of uic_topdown_grad_ready_wait
    # (include/uic_hip.h): each piece's all-reduce starts while the step is still computing the following ones
    FIRST_GRADS = ("logit.",)
    LSTM_W_GRADS = ("core.lang_lstm.weight_", "core.att_lstm.weight_hh")      # final right after the BPTT loop
    # opt.early_grads (UIC_REC_EARLY_GRADS, include/uic_hip.h): these two join them, 62 % of the bytes final 0.17 ms before the
    # step ends for a step that is 4 % longer on its own -- a data-parallel run's choice
    LSTM_W_GRADS_EARLY = ("core.lang_lstm.weight_", "core.att_lstm.weight_", "embed.")
    LATE_GRADS = ("att_embed.", "ctx2att.", "core.attention.")

    def build_optimizer(self):
        self.i2t_model.cuda()
        names = self.i2t_model.param_names
        first = [k for k in names if k.startswith(self.FIRST_GRADS)]
        early_order = bool(getattr(self.opt, 'early_grads', False))
        eng = getattr(self.i2t_model, 'engine', None)
        if eng is not None:
            eng.recurrence = (eng.recurrence | _lib.REC_EARLY_GRADS) if early_order else (eng.recurrence & ~_lib.REC_EARLY_GRADS)
        lstm_w = [k for k in names if k.startswith(self.LSTM_W_GRADS_EARLY if early_order else self.LSTM_W_GRADS)]
        late = [k for k in names if k.startswith(self.LATE_GRADS)]
        early = [k for k in names if k not in first and k not in lstm_w and k not in late]
        groups = (first, lstm_w, early, late)
        # gradient groups 4 / 3 of uic_topdown_grad_ready_wait: the embedding table is the first piece of the tail to be final, and
        # att_lstm.weight_ih is complete with the recurrent matrices (both empty lists under opt.early_grads: group 1 has them then)
        emb = [k for k in early if k == "embed.0.weight"]
        wih = [k for k in early if k == "core.att_lstm.weight_ih"]
        ex = self.exchange
        self.sharded = bool(ex.world_size > 1 and eng is not None and hasattr(self.i2t_model, 'use_bn') and
                            not getattr(self.opt, 'allreduce_exchange', 0))
        if self.sharded:
            # the SHARDED exchange (parallel_exchange.py): only the matrices the kernels consume in the operand dtype are sharded;
            # everything the kernels read as f32 (biases, alpha_net, BatchNorm tensors -- and att_embed's Linear when it has to be
            # folded with the BatchNorm or zero-padded from its f32 master) stays replicated in the arena's tail
            # pieces in the order the backward pass finishes them, each with the gradient group whose event releases its
            # reduce-scatter (None: after the step has joined): logit | embedding table | recurrent LSTM matrices + att_lstm.weight_ih |
            # fc_embed + the late group's matrices
            mats = self._gathered_keys()
            cand = [(first, 0), (emb, 4), (lstm_w + wih, 3 if wih else 1), ([k for k in early if k not in emb + wih] + late, None)]
            pieces = [[k for k in g if k in mats] for g, _ in cand]
            self.piece_groups = [gid for (g, gid), p in zip(cand, pieces) if p]
            op_dtype = _lib.TORCH_DTYPE[eng.dtype]
            self.arena = FlatArena(self.i2t_model, [k for g, _ in cand for k in g], world=ex.world_size, rank=ex.rank,
                                   pieces=[p for p in pieces if p], operand_dtype=op_dtype)
            self.arena_splits = []
            eng.gathered = _GatheredWeights(self.arena, mats)
            self._next_den = None
            arena = self.arena
            self.i2t_model._stale_masters = lambda: arena.masters_stale
        else:
            self.arena = FlatArena(self.i2t_model, first + lstm_w + early + late)
            self.arena_splits = [self.arena.offsets[g[0]] for g in groups[1:]] if first and all(groups[1:]) else []
            if eng is not None:
                eng.gathered = None
            self.i2t_model._stale_masters = None
        self._step = 0

    def _gathered_keys(self):
        """{reference key: (uic_topdown_gathered field, gather group)} of the parameters the sharded exchange all-gathers in the
        operand dtype."""
        m = self.i2t_model
        eng = m.engine
        by_field = {f: k for f, k, is_param in _lib.weight_fields(m.use_bn, m.logit_layers) if is_param}
        out = {}
        for gf, wf, grp in _lib.GATHERED_FIELDS:
            if gf == "att_w" and (m.use_bn or eng.Dp != eng.D):
                continue
            out[by_field[wf]] = (gf, grp)
        for l in range(m.logit_layers - 1):
            out[by_field["logit_h_w:%d" % l]] = ("logit_h_w:%d" % l, 0)
        return out

    def update_LearningRate(self, epoch):
        """Optim.update_LearningRate('i2t', epoch), P/misc/optimizer.py:114-122."""
        o = self.opt
        start = getattr(o, 'i2t_learning_rate_decay_start', -1)
        if epoch > start and start >= 0:
            frac = (epoch - start) // getattr(o, 'i2t_learning_rate_decay_every', 3)
            self.i2t_current_lr = self.lr * getattr(o, 'i2t_learning_rate_decay_rate', 0.8) ** frac
        else:
            self.i2t_current_lr = self.lr

    def to_device(self, data, per_image=True):
        """numpy batch dict of DataLoader.get_batch -> device tensors (P/trainer.py:147-149).

        The loader replicates every image's features seq_per_img times on the host (P/misc/dataloader/dataloader.py:
        270-277).  With opt.seq_per_img > 1 only every seq_per_img-th feature row crosses PCIe (37.7 MB instead of
        188.7 MB at config 2) and the replication happens on the device (uic_topdown_dims.seq_per_img); the first batch
        is checked to really be replicated.  opt.ship_replicated_features = 1 keeps the reference's behaviour."""
        S = int(getattr(self.opt, 'seq_per_img', 1) or 1)
        if not per_image or getattr(self.opt, 'ship_replicated_features', 0) or not hasattr(self.i2t_model, 'use_bn'):
            S = 1
        out = {}
        for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks"):
            v = data.get(k)
            if v is None:
                continue
            if S > 1 and k in ("fc_feats", "att_feats", "att_masks") and isinstance(v, np.ndarray) and len(v) == len(data["labels"]):
                if not getattr(self, '_replication_checked', False):
                    for j in range(1, S):
                        if not (np.asarray(v[j::S]) == np.asarray(v[::S])).all():
                            raise ValueError("opt.seq_per_img=%d but the rows of %s are not %d-fold replicated; set "
                                             "opt.ship_replicated_features=1 for per-caption features" % (S, k, S))
                v = v[::S]
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v if v.flags.writeable else v.copy())
            out[k] = self._ship(k, v, torch.int64 if k == "labels" else torch.float32)
        self._replication_checked = True
        if getattr(self.opt, 'live_positions', 1) and data.get("masks") is not None and not torch.is_tensor(data["masks"]):
            # how many positions of each decode step lie in front of their caption's end is known here, on the host, where the
            # loader made the masks: the step's logit layer and criterion skip the others (uic_topdown_batch.live_count: the step
            # compacts the device masks itself, nothing more is shipped; opt.live_positions = 0 computes every position)
            out["live_count"] = live_counts(data["masks"])
        return out

    @staticmethod
    def attach_live(batch):
        """Adds the per-step counts of unmasked positions to a DEVICE batch (a resident benchmark batch; Trainer.to_device does it
        for host batches).  One device-to-host read of the masks: call it once per batch, outside a timed region."""
        batch["live_count"] = live_counts(batch["masks"])
        return batch

    def _ship(self, key, t, dtype):
        """Host tensor (any strides / dtype) -> device: ONE threaded gather-and-convert pass into a pinned staging buffer
        (two per key, guarded by events) and an asynchronous copy, instead of ascontiguousarray + a pageable hipMemcpy
        that stages a second time."""
        if t.is_cuda:
            return t.to(dtype)
        ring = self.__dict__.setdefault('_pin', {}).setdefault(key, [])
        slot = self.__dict__.setdefault('_pin_slot', {})
        i = slot.get(key, 0)
        slot[key] = (i + 1) % 2
        while len(ring) <= i:
            ring.append(None)
        if ring[i] is None or ring[i][0].shape != t.shape or ring[i][0].dtype != dtype:
            ring[i] = [torch.empty(t.shape, dtype=dtype, pin_memory=True), None]
        pin, ev = ring[i]
        if ev is not None:
            ev.synchronize()                       # the copy that last read this buffer has finished
        # at most 8 copy threads: a machine-wide OpenMP team (256 threads on the MI355X hosts) keeps spinning after the
        # copy and slows the following kernel enqueues 4x (measured: the replay + BPTT enqueue 4.2 -> 16.7 ms)
        nthr = torch.get_num_threads()
        if nthr > 8:
            torch.set_num_threads(8)
        try:
            pin.copy_(t)
        finally:
            if nthr > 8:
                torch.set_num_threads(nthr)
        dev = pin.cuda(non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring[i][1] = ev
        return dev

    def train_device_batch(self, batch, t_run, den_local, next_den_local=None):
        """The timed hot path: everything from device-resident inputs to updated weights.  next_den_local (data parallel): the
        NEXT batch's local mask sum, when the caller knows it (Trainer.train(next_data=...), a benchmark's resident batch): it
        rides in this step's small all-reduce, so that no collective sits in front of the next step's forward pass."""
        if self.arena is None:
            self.build_optimizer()
        dev = batch["fc_feats"].device
        if getattr(self, 'sharded', False):
            a = self.arena
            inv_den = self._take_next_inv_den(den_local)
            if inv_den is None:
                inv_den = self.exchange.global_inv_den(den_local, dev)
            loss_slot = a.scalars[0:2]                              # the fused step writes [loss, sum(mask)]; slot 1 is rewritten below
            loss, _ = xe_step(self.i2t_model, batch, t_run=t_run, inv_den=inv_den, grads=a.grad_views, out=loss_slot)
            loss = self._sharded_update(loss, 1.0, next_den_local)
            self.last_loss = loss
            return loss
        inv_den = self.exchange.global_inv_den(den_local, dev)
        loss, _ = xe_step(self.i2t_model, batch, t_run=t_run, inv_den=inv_den, grads=self.arena.grad_views)
        loss = self._exchange_and_adam(loss, 1.0)
        self.last_loss = loss
        return loss

    def _exchange_and_adam(self, loss, grad_scale, next_den_local=None):
        """Gradient exchange + optimizer step behind a backward pass of the captioner (the XE step and the self-critical step's
        resumed backward both end in uic_topdown_xe_train_step, whose gradient-group events the overlapped forms wait on)."""
        if getattr(self, 'sharded', False):
            return self._sharded_update(loss, grad_scale, next_den_local)
        if self.exchange.world_size > 1:
            # opt.allreduce_exchange: round 5's exchange -- the arena all-reduced in four pieces, the first three on the communication
            # stream as they become final; Adam on the whole arena on every rank
            lib = _lib.load()
            self.exchange.allreduce_sum_overlapped(
                self.arena.grad, getattr(self, 'arena_splits', []) if hasattr(self.i2t_model, 'use_bn') else [],
                lambda raw, g: check(lib.uic_topdown_grad_ready_wait(raw, g), "grad_ready_wait"))
        return self._guarded_adam(loss, grad_scale)

    # ------------------------------------------------------------------ the sharded exchange
    def _take_next_inv_den(self, den_local):
        """1 / (global mask sum) of THIS batch if the previous step's all-reduce carried it (its local share must be the value
        announced then), else None."""
        nd = getattr(self, '_next_den', None)
        self._next_den = None
        if nd is not None and den_local is not None and abs(float(nd[0]) - float(den_local)) <= 1e-6 * max(1.0, abs(float(den_local))):
            return nd[1]
        return None

    def _comm(self, dev):
        if getattr(self, '_comm_stream', None) is None or self._comm_stream.device != dev:
            self._comm_stream = torch.cuda.Stream(device=dev)
        return self._comm_stream

    def _sharded_update(self, loss, grad_scale, next_den_local=None, overlap=True):
        """The sharded exchange of the captioner step (FlatArena.sharded_step): reduce-scatter of the gradient pieces -- the first
        three beside the rest of the backward pass, behind uic_topdown_grad_ready_wait --, one small all-reduce of the replicated
        tail with the step's scalars, Adam on this rank's slices (skipped on the device on every rank if a persistent launch timed
        out anywhere), all-gather of the updated operand-dtype weights on the communication stream in the order the next forward
        pass consumes them: the next refresh waits per group (uic_topdown_refresh_weights_gathered), nothing here does.
        Returns the whole batch's loss (device scalar)."""
        a = self.arena
        dev = a.flat.device
        lib = _lib.load()
        sc = a.scalars
        # scalar slots: [0] loss, [1] status flag, [2] (clip norm, unused here), [3] the next batch's mask sum
        if loss.data_ptr() != sc.data_ptr():
            sc[0:1].copy_(loss.detach().reshape(1))
        if getattr(self.i2t_model, 'engine', None) is not None:
            sc[1:2].copy_(_lib.status_words(dev)[0:1])             # (int32 -> float: non-zero stays non-zero)
        else:
            sc[1:2].zero_()
        if next_den_local is not None:
            sc[3:4].copy_(self._host_float(float(next_den_local), dev), non_blocking=True)
        groups = self.piece_groups

        def wait_piece(raw, i):
            if not overlap or groups[i] is None:
                return False                                        # the last piece: the step's own stream has joined
            check(lib.uic_topdown_grad_ready_wait(raw, groups[i]), "grad_ready_wait")
            return True
        self._step += 1
        # the logit layer's piece -- a quarter of the bytes, final when the BPTT loop STARTS -- is reduce-scattered, updated and
        # all-gathered on the communication stream while the loop runs (no kernel of the step reads logit.weight after that point);
        # the status word it is guarded by is final then too (the step's only persistent launch is the forward recurrence)
        has_eng = getattr(self.i2t_model, 'engine', None) is not None
        pipe = (0,) if (has_eng and overlap and groups and groups[0] == 0 and not getattr(self.opt, 'no_pipelined_logit_piece', 0)) else ()
        # opt.bf16_gradient_exchange: every piece that is NOT pipelined behind the BPTT loop -- i.e. the bytes that are still on the
        # wire when the step has joined -- is reduce-scattered as bf16
        half = tuple(i for i in range(len(a.pieces)) if i not in pipe) if getattr(self.opt, 'bf16_gradient_exchange', 0) else ()
        pair, events = a.sharded_step(self.exchange, self.i2t_current_lr, self.betas, self.eps, self._step, grad_scale,
                                      wait_piece=wait_piece, comm=self._comm(dev), gather_async=True,
                                      early_guard=_lib.status_words(dev) if pipe else None, pipeline=pipe, half=half)
        self._guard_pair = pair
        if next_den_local is not None:
            self._next_den = (float(next_den_local), sc[3:4].reciprocal())
        eng = getattr(self.i2t_model, 'engine', None)
        if eng is not None and getattr(eng, 'gathered', None) is not None:
            eng.gathered.set_events(events)
        return pair[0]

    def _host_float(self, x, dev):
        """A host float as a 1-element device tensor without a synchronous pageable copy (ring of pinned floats)."""
        if getattr(self, '_pinf', None) is None:
            self._pinf = torch.zeros(64, dtype=torch.float32).pin_memory()
            self._pinf_i = 0
        i = self._pinf_i
        self._pinf_i = (i + 1) % 64
        self._pinf[i] = x
        return self._pinf[i:i + 1].to(dev, non_blocking=True)

    def gather_masters(self):
        """Collective (every rank calls it): after a sharded bf16 step a rank's f32 master weights are current only inside its own
        shard -- this all-gathers them, so that state_dict() / save_models() see the full f32 weights everywhere."""
        a = self.arena
        if a is not None and getattr(a, 'masters_stale', False):
            dev = a.flat.device
            torch.cuda.current_stream(dev).wait_stream(self._comm(dev))
            a.gather_masters(self.exchange)

    def _guarded_adam(self, loss, grad_scale):
        """Adam on the flat arena, skipped ON THE DEVICE if a persistent recurrence launch of this step timed out
        (uic_adam_step_guarded reads the status word the kernels set): a bad step never reaches the weights or the moments.
        Data parallel: [loss, status] travel in ONE 2-float all-reduce issued after the gradient exchange, so the returned loss
        is the sum over the ranks and every rank skips (and later raises, _finish_step) together.  Returns the loss, a device
        scalar."""
        a = self.arena
        self._step += 1
        has_eng = getattr(self.i2t_model, 'engine', None) is not None
        status = _lib.status_words(a.flat.device) if has_eng else None
        self._guard_pair = None
        if self.exchange.world_size > 1:
            flag = status[0:1].float() if status is not None else loss.new_zeros(1)
            pair = torch.cat([loss.detach().float().reshape(1), flag])
            self.exchange._sum(pair)
            self._guard_pair = pair
            guard, loss = pair[1:], pair[0]
        else:
            guard = status
        check(_lib.load().uic_adam_step_guarded(ptr(a.flat), ptr(a.grad), ptr(a.exp_avg), ptr(a.exp_avg_sq), a.numel,
                                                self.i2t_current_lr, self.betas[0], self.betas[1], self.eps, self._step,
                                                grad_scale, ptr(guard), stream()), "adam_step_guarded")
        return loss

    def _raise_if_timed_out(self, cause):
        """After a failure inside a step: if a persistent launch of the step timed out, clear the status word and raise
        PersistentTimeout (chained to `cause`) -- the step's update never ran or was skipped on the device."""
        if isinstance(cause, _lib.PersistentTimeout) or getattr(self.i2t_model, 'engine', None) is None or self.arena is None:
            return
        word = _lib.status_words(self.arena.flat.device)
        try:
            code = int(word[0].item())
        except Exception:
            return
        if code != 0:
            word[0:1].zero_()
            raise _lib.PersistentTimeout("persistent recurrence kernel timed out (code 0x%x): the step failed before its update" % code) from cause

    def _finish_step(self, loss):
        """The step's ONE host sync (the reference's loss.item(), P/trainer.py:172): returns the loss as a float and raises --
        on every rank of a data-parallel run -- if a persistent launch of the step timed out (another process on the GPU, a
        long kernel on another stream).  The update was then already skipped on the device (_guarded_adam): weights, moments
        and the step counter are those of before the step, and training can go on after the caller has dealt with the cause
        (engine.recurrence |= _lib.REC_FWD_CHAIN selects per-step launches, which cannot time out)."""
        pair = getattr(self, '_guard_pair', None)
        self._guard_pair = None
        try:
            if pair is not None:
                vals = pair.cpu().tolist()
                if vals[1] != 0:
                    raise _lib.PersistentTimeout("persistent recurrence kernel timed out on some rank (summed status flag %r): this step's "
                                                 "update was skipped on every rank" % (vals[1],))
                _lib.persistent_status(self.arena.flat.device)  # (clears nothing: the local word is zero too)
                return vals[0]
            val = loss.item()
            if getattr(self.i2t_model, 'engine', None) is not None:
                _lib.persistent_status(self.arena.flat.device)  # raises and clears the word if the launch timed out
            return val
        except _lib.PersistentTimeout:
            # ONLY the time-out rolls the step back (the update was skipped on the device): any other failure -- a HIP error out
            # of loss.item(), a collective's -- leaves the counters and the launch statistics alone
            if getattr(self.i2t_model, 'engine', None) is not None:
                _lib.status_words(self.arena.flat.device)[0:1].zero_()
            self._step -= 1                                     # the skipped step does not count for Adam's bias correction
            raise

    def prefetch(self, data, per_image=True):
        """Ship a batch to the device on a copy stream NOW (pinned staging + asynchronous H2D), to be consumed by the next
        train(data) / train_self_critical(data) call with this very dict.  Called by train(..., next_data=...) right after a
        step is enqueued, so the host gather and the PCIe transfer of batch k+1 run while the GPU computes batch k."""
        if getattr(self, '_copy_stream', None) is None:
            self._copy_stream = torch.cuda.Stream()
        with torch.cuda.stream(self._copy_stream):
            if callable(data):
                # e.g. `lambda: loader.get_batch('train')`: the loader's host work, its H2D copies and its assembly kernel
                # all happen here -- after this step was enqueued, on the copy stream; the batch is left in self.next_data
                data = data()
            self.next_data = data
            if data is None:                          # the caller's source is exhausted
                self._prefetched = None
                return
            batch = self.to_device(data, per_image)
            ev = torch.cuda.Event()
            ev.record()
        self._prefetched = (id(data), per_image, batch, ev)

    def _device_batch(self, data, per_image=True):
        pf = getattr(self, '_prefetched', None)
        self._prefetched = None
        if pf is not None and pf[0] == id(data) and pf[1] == per_image:
            cur = torch.cuda.current_stream()
            cur.wait_event(pf[3])
            for t in pf[2].values():
                if torch.is_tensor(t):
                    t.record_stream(cur)               # allocated on the copy stream, consumed on this one
            return pf[2]
        return self.to_device(data, per_image)

    def train(self, data, loader=None, iteration=None, epoch=None, nmt_epoch=None, next_data=None):
        """Trainer.train for the XE captioner step (P/trainer.py:141-173,193).  next_data (optional extension): the
        following batch -- or a callable returning it, e.g. `lambda: loader.get_batch('train')` -- fetched / shipped to the
        device while this step computes (see prefetch); afterwards it is `self.next_data`."""
        labels_np = np.asarray(data["labels"])
        t_run = _steps_from_host_labels(labels_np)
        T = labels_np.shape[1] - 1
        den_local = self._mask_sum(data)
        batch = self._device_batch(data)
        if getattr(self, 'sharded', False) or (self.arena is None and self.exchange.world_size > 1):
            # sharded exchange: the NEXT batch's mask sum rides in this step's small all-reduce, so that no collective sits in
            # front of the next forward pass.  A dict is known now; a callable is fetched after the step's compute is enqueued (its
            # host work then overlaps the GPU's), and the exchange -- which needs the number -- is enqueued behind it
            if self.arena is None:
                self.build_optimizer()
        if getattr(self, 'sharded', False) and next_data is not None:
            if callable(next_data):
                a = self.arena
                inv_den = self._take_next_inv_den(den_local)
                if inv_den is None:
                    inv_den = self.exchange.global_inv_den(den_local, batch["fc_feats"].device)
                loss, _ = xe_step(self.i2t_model, batch, t_run=t_run, inv_den=inv_den, grads=a.grad_views, out=a.scalars[0:2])
                self.prefetch(next_data)
                nd = self._mask_sum(self.next_data) if self.next_data is not None else None
                loss = self._sharded_update(loss, 1.0, nd)
                self.last_loss = loss
            else:
                loss = self.train_device_batch(batch, t_run, den_local, self._mask_sum(next_data))
                self.prefetch(next_data)
        else:
            loss = self.train_device_batch(batch, t_run, den_local)       # (already summed over the ranks)
            if next_data is not None:
                self.prefetch(next_data)
        self.i2t_train_loss = self._finish_step(loss)     # the reference's per-step host sync (trainer.py:172)
        return self.i2t_train_loss

    @staticmethod
    def _mask_sum(data):
        """LanguageModelCriterion's denominator of a host batch: sum of masks[:, 1:] (P/misc/criterion.py:147-149)."""
        m = np.asarray(data["masks"])
        return float(m[:, 1:].sum())

    def _sample_opt(self, S):
        """Options of the self-critical step's sampling pass (P/trainer.py:167: sample_max = 0).  self.forced_samples (tests):
        an int64 [rows, L] tensor the pass replays instead of drawing -- its log-probs and gradients are those of these tokens."""
        o = {'sample_max': 0, 'captions_per_image': S}
        forced = getattr(self, 'forced_samples', None)
        if forced is not None:
            o['forced_tokens'] = forced
        return o

    def _scst_decode(self, model, eng, fc, att, am, S, overlap, cur):
        """The sampling pass (train mode) and the greedy baseline (eval mode, P/misc/rewards.py:42-47) of the self-critical
        step; overlap: the baseline on a second stream beside the sampling pass."""
        if overlap:
            if getattr(self, '_baseline_stream', None) is None:
                self._baseline_stream = torch.cuda.Stream()
            # the derived weight copies are rebuilt once, here, on the current stream (hold_weights: later calls reuse them);
            # the baseline stream then orders itself behind that and behind the batch's H2D copies
            eng.refresh({k: v.detach() for k, v in model.param_dict().items()}, eng.dims(att.shape[0], att.shape[1], model.seq_length + 1))
            self._baseline_stream.wait_stream(cur)
            # (every pass draws its seed from the model's counter: the sampling pass gets the seed it has in the serial order,
            # so that the two orders train identically, bit for bit)
            c0 = getattr(model, '_seed_counter', None)
            if c0 is not None:
                model.next_seed()
            model.eval()
            with torch.cuda.stream(self._baseline_stream), torch.no_grad():
                greedy_res, _ = model(fc, None, att, am, opt={'sample_max': 1}, mode='sample')
            model.train()
            c2 = getattr(model, '_seed_counter', None)
            if c0 is not None:
                model._seed_counter = c0
            gen_result, sample_logprobs = model(fc, None, att, am, opt=self._sample_opt(S), mode='sample')
            if c0 is not None:
                model._seed_counter = c2
            cur.wait_stream(self._baseline_stream)
            greedy_res.record_stream(cur)
        else:
            model.train()
            gen_result, sample_logprobs = model(fc, None, att, am, opt=self._sample_opt(S), mode='sample')
            model.eval()
            with torch.no_grad():                                       # rewards.py:42-47: greedy baseline, eval mode
                greedy_res, _ = model(fc, None, att, am, opt={'sample_max': 1}, mode='sample')
            model.train()
        return gen_result, sample_logprobs, greedy_res

    @staticmethod
    def _reward_mask_sum(seq):
        """Sum of RewardCriterion's mask (P/misc/criterion.py:118-120): (seq > 0) shifted right by one with a leading 1 -- every
        row counts its tokens up to and including the first 0.  A device scalar, no host sync."""
        m = (seq[:, :-1] > 0).sum() + seq.shape[0]
        return m.to(torch.float32).reshape(1)

    def train_self_critical(self, data, reward_fn=None, next_data=None):
        """The self-critical branch of Trainer.train (P/trainer.py:166-171).  reward_fn None: the reference's reward,
        CIDEr-D(sampled) - CIDEr-D(greedy) against data['gts'] with the cached document frequencies of
        opt.cached_tokens (P/misc/rewards.py:37-81), scored on the device -- the captions never visit the host and the
        step synchronises once, for loss.item().  A callable `reward_fn(data, sampled, greedy)` -> float array [N, L]
        replaces the scorer (host round trip)."""
        from .misc.criterion import RewardCriterion
        from .misc import rewards
        if self.arena is None:
            self.build_optimizer()
        batch = self._device_batch(data)
        model = self.i2t_model
        fc, att, am = batch["fc_feats"], batch["att_feats"], batch.get("att_masks")
        n_rows = len(data["labels"]) if data.get("labels") is not None else att.shape[0]
        S = n_rows // att.shape[0]                        # > 1 when to_device shipped every image once
        import contextlib
        eng = getattr(model, 'engine', None)
        try:
            # the weights stay put until Adam below: ONE weight refresh serves the sampling pass, the greedy baseline and the replay
            with (eng.hold_weights() if hasattr(eng, 'hold_weights') else contextlib.nullcontext()):
                # The sampling pass (train mode) and the greedy baseline (eval mode, rewards.py:42-47) are independent latency chains
                # of ~7 small launches per decode step: the baseline runs on a second stream beside the sampling pass.  Not with
                # BatchNorm in att_embed: there the train-mode pass updates the running statistics the eval-mode pass reads (the
                # reference runs them in this order), so the two stay serial.
                cur = torch.cuda.current_stream()
                # self.persistent_decode: each pass as ONE persistent launch (rnn_persist.hip's decode mode).  A persistent launch
                # holds every CU, so the two passes then run one after the other -- measured slower for this step than the two
                # launch chains side by side (profiles/), hence off by default here; a lone decode pass (eval) takes it by itself.
                persistent = bool(getattr(self, 'persistent_decode', False))
                overlap = hasattr(eng, 'hold_weights') and int(getattr(model, 'use_bn', 0) or 0) == 0 and \
                    not getattr(self, 'serial_baseline', False) and not persistent
                rec0 = getattr(eng, 'recurrence', 0)
                if eng is not None and not persistent:
                    eng.recurrence = rec0 | _lib.REC_FWD_CHAIN
                defer0 = getattr(model, 'defer_status_check', False)
                model.defer_status_check = True               # (no host sync between the passes: _finish_step reads the status word)
                try:
                    gen_result, sample_logprobs, greedy_res = self._scst_decode(model, eng, fc, att, am, S, overlap, cur)
                finally:
                    model.defer_status_check = defer0
                    if eng is not None:
                        eng.recurrence = rec0
                if reward_fn is None:
                    scorer = rewards.init_scorer(getattr(self.opt, 'cached_tokens', 'corpus'))
                    reward_t = rewards.self_critical_reward_device(scorer, gen_result, greedy_res, data['gts'],
                                                                   float(getattr(self.opt, 'cider_reward_weight', 1)),
                                                                   float(getattr(self.opt, 'bleu_reward_weight', 0)))
                else:
                    if S > 1:                                     # eval mode is deterministic: the S replicas decode identically
                        greedy_res = greedy_res.repeat_interleave(S, 0)
                    reward = np.asarray(reward_fn(data, gen_result.cpu().numpy(), greedy_res.cpu().numpy()), dtype=np.float32)
                    reward_t = torch.from_numpy(reward).cuda()
                dw = float(getattr(self.opt, 'disc_reward_weight', 0) or 0)
                if dw > 0 and getattr(self, 'discriminator', None) is not None:
                    # adversarial reward of BASELINE configs[3]: + w (D(sampled) - D(greedy)), the same self-critical form, per row
                    g_rows = greedy_res.repeat_interleave(S, 0) if greedy_res.shape[0] != gen_result.shape[0] else greedy_res
                    adv = self.discriminator_scores(gen_result) - self.discriminator_scores(g_rows)
                    reward_t = reward_t + dw * adv[:, None]
                loss = RewardCriterion()(sample_logprobs, gen_result, reward_t)
                # Data parallel: the reference gathers every replica's output to GPU0 and divides by the mask sum of the WHOLE batch
                # (P/misc/criterion.py:117-122 on the gathered tensors, P/trainer.py:168-170); ranks whose captions differ in length must
                # therefore not average their per-rank means.  Each rank rescales its mean by (its mask sum) / (sum over ranks) -- a
                # 1-float all-reduce before the backward pass -- and the gradients (and losses) are then SUMMED, as in the XE step.
                if self.exchange.world_size > 1:
                    loss = loss * self.exchange.global_share(self._reward_mask_sum(gen_result))
                # every p.grad is its view of the flat gradient arena: the backward kernels write there directly (models/AttModel.py
                # _TopDownSample.backward); a model without that path hands tensors back to autograd, copied below
                params = dict(model.named_parameters())
                views = self.arena.grad_views
                sink_ok = getattr(model, 'supports_grad_sink', False)
                for k, p in params.items():
                    p.grad = views[k] if (sink_ok and k in views) else None
                model._grad_sink = views if sink_ok else None
                try:
                    loss.backward()
                finally:
                    model._grad_sink = None
            for k, view in views.items():
                if params[k].grad is not view:
                    view.copy_(params[k].grad)
            # (the backward pass above ended in the fused step: its gradient-group events release the exchange's first pieces
            # while the tail still runs, exactly as in the XE step; a model without that path exchanges after the pass)
            if sink_ok:
                loss_d = self._exchange_and_adam(loss.detach(), 1.0)          # (already the whole batch's loss: summed over the ranks)
            else:
                self.exchange.allreduce_sum(self.arena.grad)
                loss_d = self._guarded_adam(loss.detach(), 1.0)
        except BaseException as e:
            # whatever went wrong between the decode passes and the guarded Adam (a reward function fed the -1 tokens of a
            # timed-out pass, for one): a time-out word left behind would make the NEXT, good step skip its update and raise
            self._raise_if_timed_out(e)
            raise
        avg = reward_t[:, 0].mean()
        if next_data is not None:
            self.prefetch(next_data)              # the next batch crosses PCIe while this step computes (see train)
        val = self._finish_step(loss_d)
        self.i2t_train_loss = val
        self.i2t_avg_reward = float(avg.item())
        return self.i2t_train_loss

    # ------------------------------------------------------------------ sentence discriminator (BASELINE configs[3]; parity unpinned)
    def build_discriminator(self):
        """The CNN sentence discriminator of the unpaired / adversarial configuration (models/Discriminator.py) with its own flat
        Adam arena; data parallel like the captioner (gradient arena summed over the ranks)."""
        from .models import SentenceDiscriminator
        from .misc.optimizer import FlatArena
        self.discriminator = SentenceDiscriminator(self.opt).cuda()
        rank = self.exchange.rank if self.exchange is not None else 0
        self.discriminator.seed = (self.discriminator.seed + 0x9E3779B1 * rank) & 0x7FFFFFFF    # per-rank dropout noise
        ex = self.exchange
        self.disc_sharded = bool(ex is not None and ex.world_size > 1 and not getattr(self.opt, 'allreduce_exchange', 0))
        if self.disc_sharded:        # one piece, f32 masters gathered in place (the discriminator's kernels read the masters)
            names = [k for k, _ in self.discriminator.named_parameters()]
            self.disc_arena = FlatArena(self.discriminator, names, world=ex.world_size, rank=ex.rank, pieces=[names])
        else:
            self.disc_arena = FlatArena(self.discriminator)
        self.disc_lr = float(getattr(self.opt, 'disc_learning_rate', 1e-4) or 1e-4)
        self._disc_step = 0
        return self.discriminator

    def _pad_rows(self, seq):
        """Caption rows [N, L'] (int64, 0 = end) as the discriminator's [N, seq_length] rows."""
        L = self.discriminator.L
        seq = seq.cuda() if not seq.is_cuda else seq
        if seq.shape[1] < L:
            seq = torch.cat([seq, seq.new_zeros(seq.shape[0], L - seq.shape[1])], 1)
        return seq[:, :L].contiguous().long()

    def discriminator_scores(self, seq):
        self.discriminator.eval()
        return self.discriminator.scores(self._pad_rows(seq))

    def train_discriminator(self, real_seq, fake_seq):
        """One BCE step: real caption rows (label 1) against generated rows (label 0).  Returns the loss (one host sync)."""
        if getattr(self, 'discriminator', None) is None:
            self.build_discriminator()
        D, a = self.discriminator, self.disc_arena
        D.train()
        tok = torch.cat([self._pad_rows(real_seq), self._pad_rows(fake_seq)])
        lab = torch.cat([torch.ones(real_seq.shape[0]), torch.zeros(fake_seq.shape[0])]).cuda()
        a.zero_grad()
        loss = D.bce(D(tok), lab)
        loss.backward()
        self._disc_step += 1
        if getattr(self, 'disc_sharded', False):
            a.scalars[0:2].zero_()
            a.scalars[0:1].copy_(loss.detach().reshape(1))
            pair, _ = a.sharded_step(self.exchange, self.disc_lr, self.betas, self.eps, self._disc_step, 1.0 / self.exchange.world_size)
            self.disc_train_loss = float(pair[0].item()) / self.exchange.world_size      # (mean of the ranks' batch means)
            return self.disc_train_loss
        self.exchange.allreduce_sum(a.grad)
        a.adam(self.disc_lr, self.betas, self.eps, self._disc_step, grad_scale=1.0 / self.exchange.world_size)
        self.disc_train_loss = loss.item()
        return self.disc_train_loss

    # ------------------------------------------------------------------ pivot NMT half (P/trainer.py:80-94,175-193)
    def build_nmt(self, src_dict, tgt_dict):
        """Trainer.build_nmt: encoder/decoder/NMTModel, the generator attached as `model.generator`, NMT_loss, and the
        optimizer over the model's flat arena.  `src_dict` / `tgt_dict`: onmt.Dict-like (`.size()`) or plain sizes."""
        import torch.nn as nn
        from .models import NMT_Models
        from .misc import criterion
        opt = self.opt
        tgt_size = int(tgt_dict.size()) if hasattr(tgt_dict, 'size') else int(tgt_dict)
        self.nmt_encoder = NMT_Models.Encoder(opt, src_dict)
        self.nmt_decoder = NMT_Models.Decoder(opt, tgt_dict)
        self.nmt_model = NMT_Models.NMTModel(opt, self.nmt_encoder, self.nmt_decoder, src_dict, tgt_dict, False)
        self._mix_rank_into_seed(self.nmt_model, self.exchange)
        self.nmt_generator = nn.Sequential(nn.Linear(opt.rnn_size, tgt_size), nn.LogSoftmax(dim=-1))
        param_init = getattr(opt, 'param_init', 0.1)
        if param_init:
            for p in list(self.nmt_model.parameters()) + list(self.nmt_generator.parameters()):
                p.data.uniform_(-param_init, param_init)
        self.nmt_model.generator = self.nmt_generator
        self.dp_nmt_model = self.nmt_model
        self.nmt_model.cuda()
        self.nmt_model.train(bool(getattr(opt, 'nmt_train_flag', 1)))
        self.nmt_loss = criterion.NMTCriterion(tgt_size, opt)
        self.nmt_crit = criterion.NMT_loss(opt, self.nmt_generator, self.nmt_loss)
        self.optim = Optim(opt, self.exchange)
        self.optim.set_parameters(None, self.nmt_model)
        self.nmt_train_ppl = self.nmt_train_acc = 0.0
        return self.nmt_model

    def train_nmt(self, nmt_batch, loader=None, nmt_epoch=None, update_lr=False):
        """The NMT branch of Trainer.train (P/trainer.py:175-193): zero_grad, forward, NMT_loss, backward,
        Optim.step (noam LR, clip_grad_norm, Adam).  Returns the summed NLL of the batch (host float)."""
        if update_lr:
            self.optim.update_LearningRate('nmt', nmt_epoch)
        self.optim.zero_grad(nmt_direct=True)         # (no 360 MB fill: the in-place backward below overwrites every gradient)
        self.nmt_model.unit_loss_gradient = True      # loss.backward() below: the kernels may write the gradient arena in place
        outputs, attn, dec_state, upper_bounds = self.dp_nmt_model(nmt_batch.src, nmt_batch.tgt, nmt_batch.lengths, None)
        nmt_loss = self.nmt_crit(loader, nmt_batch, outputs, attn)
        nmt_loss.backward()
        sharded = getattr(self.optim, 'nmt_sharded', False)
        if sharded:
            # the step's loss rides in the sharded exchange's small all-reduce (slot 0 of the arena's scalar slots)
            self.optim.nmt_arena.scalars[0:1].copy_(nmt_loss.detach().reshape(1))
        self.optim.step()
        # the statistics are read AFTER the whole step is enqueued (one host sync per step, at its end)
        self.nmt_crit.report_stats.n_src_words += int(nmt_batch.lengths.sum())
        self.nmt_train_ppl = self.nmt_crit.report_stats.ppl()
        self.nmt_train_acc = self.nmt_crit.report_stats.accuracy()
        loss_d = nmt_loss.detach()
        if sharded:
            loss_d = self.optim.last_pair[0]
        elif self.exchange is not None and self.exchange.world_size > 1:
            # the criterion is a SUM over target words (size_average=False, P/misc/criterion.py:126-136): the whole batch's loss
            # is the sum over the ranks' column shards, as DataParallel(dim=1) + gather gives the reference (P/trainer.py:88,178)
            loss_d = loss_d.float().reshape(1).clone()
            self.exchange._sum(loss_d)
        val = float(loss_d)                           # (the step's host sync)
        guard = getattr(self.optim, 'last_guard', None)
        if guard is not None and float(guard[0].item()) != 0:
            # a persistent launch of this step timed out (on some rank): Optim.step skipped the update on the device
            _lib.status_words(self.optim.nmt_arena.flat.device)[0:1].zero_()
            self.optim._step -= 1
            self.optim._nmt_steps -= 1
            raise _lib.PersistentTimeout("persistent NMT kernel timed out: this step's update was skipped (weights and moments unchanged); "
                                         "engine.recurrence |= _lib.REC_FWD_CHAIN selects per-step launches")
        return val

    def save_models(self, tag=''):
        """P/trainer.py:98-104: model_i2t[-best].pth = state_dict of the un-wrapped module."""
        path = self.opt.checkpoint_path
        os.makedirs(path, exist_ok=True)
        self.gather_masters()        # (sharded bf16 exchange: a collective -- every rank calls save_models, as every rank runs the loop)
        if self.exchange is not None and self.exchange.rank != 0:
            return                   # one writer: the ranks hold identical weights
        if self.i2t_model is not None:
            torch.save({k: v.detach().cpu().clone() for k, v in self.i2t_model.state_dict().items()},
                       os.path.join(path, 'model_i2t' + tag + '.pth'))
        if getattr(self, 'nmt_model', None) is not None and getattr(self.opt, 'nmt_train_flag', 0):
            torch.save({k: v.detach().cpu().clone() for k, v in self.nmt_model.state_dict().items()},
                       os.path.join(path, 'model_nmt' + tag + '.pth'))
