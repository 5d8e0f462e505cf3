"""`Optim` with the reference's surface (P/misc/optimizer.py:9-131): set_parameters / step / zero_grad /
update_LearningRate / update_ScheduledSampling_prob, for the captioner ('i2t') and the pivot NMT model ('nmt').

MI355X-first: each model's parameters, gradients and Adam moments live in ONE flat f32 arena (`FlatArena`), so the
optimizer step is one kernel launch (`uic_adam_step[_clip]`) instead of ~40 per-tensor launches, gradient clipping
reads the squared norm from the device (no host sync), and data parallelism is one RCCL all-reduce of the gradient
arena.  `p.grad` of every parameter is a view of the arena, so `loss.backward()` accumulates straight into it.

Reference behaviour kept on purpose:
  * the captioner's clip_grad_norm is a NO-OP in the reference (`i2t_params` is a generator already consumed by the
    optimizer's constructor, :78-79,92), so it is not applied here either;
  * the NMT clip uses --nmt_max_grad_norm (default 5, opts.py:123) on the global L2 norm, coefficient
    max_norm / (norm + 1e-6) applied only when < 1 (torch.nn.utils.clip_grad_norm);
  * 'noam' sets lr = nmt_lr * rnn_size^-0.5 * min(step^-0.5, step * warmup^-1.5) from the SHARED step counter (:95-98);
  * update_LearningRate('nmt') is a single-shot decay (lr * rate, not rate^k, :125-131).
Only Adam is on the hot path; other methods raise NotImplementedError.
"""
import torch

from .. import _lib
from .._lib import check, ptr, stream


class FlatArena(object):
    """All parameters of a module re-homed into one flat f32 tensor (plus same-layout grad / Adam arenas).

    pieces (data parallel, world > 1): lists of parameter names -- the SHARDED gradient pieces, in the order in which the backward
    pass makes them final.  Each piece is padded to a multiple of 64 * world elements and split evenly over the ranks: rank r owns
    elements [off + r * len / world, off + (r + 1) * len / world) of every piece -- it receives that slice of the summed gradient
    from the reduce-scatter, runs Adam on it and contributes it to the all-gather of the updated weights.  Parameters of `names` in
    no piece are REPLICATED: they sit in a tail region that is all-reduced and updated identically on every rank (biases and the other
    small tensors the kernels read as f32), followed by 64 scalar slots that ride in the same all-reduce (loss, status word, the
    next batch's mask sum, squared gradient norm).  operand_dtype (torch.bfloat16): the all-gather distributes the weights in the
    operand dtype from `w16` (half the bytes; a rank's f32 masters are then current only inside its own shard until
    gather_masters()); None / float32: the all-gather runs on the f32 masters in place."""

    SCALAR_SLOTS = 64

    def __init__(self, module, names=None, world=1, rank=0, pieces=None, operand_dtype=None):
        params = dict(module.named_parameters())
        self.world, self.rank = int(world), int(rank)
        self.pieces, self.piece_names = [], []
        self.offsets = {}
        off = 0
        if pieces:
            all_names = list(names) if names is not None else list(params.keys())
            unit = 64 * self.world
            sharded = []
            for piece in pieces:
                piece = [k for k in piece if k in params]
                if not piece:
                    continue
                start = off
                for k in piece:
                    self.offsets[k] = off
                    off += (params[k].numel() + 63) // 64 * 64
                off = start + (off - start + unit - 1) // unit * unit
                self.pieces.append((start, off - start))
                self.piece_names.append(piece)
                sharded += piece
            self.replicated = [k for k in all_names if k not in self.offsets]
            self.names = sharded + self.replicated
            self.repl_off = off
            for k in self.replicated:
                self.offsets[k] = off
                off += (params[k].numel() + 63) // 64 * 64
            self.repl_end = off
        else:
            self.names = list(names) if names is not None else list(params.keys())
            self.replicated = []
            for k in self.names:
                self.offsets[k] = off
                off += (params[k].numel() + 63) // 64 * 64          # 256-byte aligned blocks
            self.repl_off = self.repl_end = off
        self.scalars_off = off
        if pieces:
            off += self.SCALAR_SLOTS
        dev = params[self.names[0]].device
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros(1024 + 8, dtype=torch.float32, device=dev)    # sqnorm partials + result
        self.scalars = self.grad[self.scalars_off:self.scalars_off + self.SCALAR_SLOTS] if pieces else None
        self.grad_views = {}
        self.params = {}
        for k in self.names:
            p = params[k]
            o, n = self.offsets[k], p.numel()
            view = self.flat[o:o + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            self.grad_views[k] = self.grad[o:o + n].view(p.shape)
            self.params[k] = p
        # the all-gathered weights: operand-dtype copies of the sharded region (bf16 runs) or the masters themselves
        self.operand_dtype = operand_dtype if (pieces and operand_dtype not in (None, torch.float32)) else None
        self.w16 = None
        self.g16 = None                 # bf16 staging of the gradient pieces that sharded_step(half=...) exchanges in bf16 (made on first use)
        self.masters_stale = False
        if self.operand_dtype is not None:
            # (the whole arena's length: uic_adam_step_ranges leaves the operand copy of EVERY element it updates, the replicated
            # tail's too -- only the sharded region [0, repl_off) is ever gathered or read)
            self.w16 = torch.zeros(self.numel, dtype=self.operand_dtype, device=dev)
            self.sync_operand_copy()
        self.gathered_views = {}
        for piece in self.piece_names:
            for k in piece:
                o, n = self.offsets[k], self.params[k].numel()
                src = self.w16 if self.w16 is not None else self.flat
                self.gathered_views[k] = src[o:o + n].view(self.params[k].shape)

    # ------------------------------------------------------------------ sharded data parallelism
    def sync_operand_copy(self):
        """w16 <- cast(masters) over the whole sharded region: after the arena is built and after the masters were written from
        outside (load_state_dict)."""
        if self.w16 is not None and self.w16.is_cuda:
            check(_lib.load().uic_cast_from_f32(_lib.dtype_id("bf16"), ptr(self.flat), ptr(self.w16), self.repl_off, stream()), "cast_from_f32")
        elif self.w16 is not None:
            self.w16[:self.repl_off].copy_(self.flat[:self.repl_off])

    def shard(self, piece, rank=None):
        """(lo, hi) of rank's slice of gradient piece `piece`."""
        r = self.rank if rank is None else rank
        off, n = self.pieces[piece]
        per = n // self.world
        return off + r * per, off + (r + 1) * per

    def owned_ranges(self):
        """The index ranges this rank's Adam updates: its slice of every piece + the replicated tensors (not the scalar slots)."""
        out = [self.shard(g) for g in range(len(self.pieces))]
        if self.repl_end > self.repl_off:
            out.append((self.repl_off, self.repl_end))
        return out

    def adam_owned(self, lr, betas, eps, step, grad_scale=1.0, max_norm=0.0, sqnorm=None, guard=None, ranges=None):
        """uic_adam_step_ranges on owned_ranges() (or the given subset of them); in operand-dtype mode the updated shard is also
        written to w16 (the rank's contribution to the all-gather)."""
        import ctypes as C
        rg = self.owned_ranges() if ranges is None else list(ranges)
        if not rg:
            return
        lo = (C.c_uint64 * len(rg))(*[a for a, _ in rg])
        hi = (C.c_uint64 * len(rg))(*[b for _, b in rg])
        clip = bool(max_norm and max_norm > 0)
        check(_lib.load().uic_adam_step_ranges(ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), len(rg), lo, hi,
                                               lr, betas[0], betas[1], eps, step, grad_scale, float(max_norm) if clip else 0.0,
                                               ptr(sqnorm) if clip else None, ptr(guard), ptr(self.w16),
                                               _lib.dtype_id("bf16") if self.w16 is not None else 0, stream()), "adam_step_ranges")
        if self.w16 is not None and self.world > 1:
            self.masters_stale = True

    def shard_sqnorm(self):
        """Sum of g^2 over this rank's slices of the pieces (after the reduce-scatter: of the SUMMED gradient), as a 1-element
        device tensor; the ranks' values add up to the squared norm of the sharded region."""
        lib = _lib.load()
        outs = self.scratch[1025:1025 + len(self.pieces)]
        for g in range(len(self.pieces)):
            lo, hi = self.shard(g)
            check(lib.uic_grad_sqnorm(ptr(self.grad[lo:hi]), hi - lo, ptr(self.scratch), ptr(outs[g:g + 1]), stream()), "grad_sqnorm")
        return outs.sum().reshape(1)

    def sharded_step(self, ex, lr, betas, eps, step, grad_scale=1.0, max_norm=0.0, wait_piece=None, comm=None, gather_async=False,
                     early_guard=None, pipeline=(), half=()):
        """One optimizer step of the sharded exchange on this arena (gradients final or becoming final on the current stream):

          1. reduce-scatter every piece -- piece i on the communication stream `comm` behind wait_piece(comm raw stream, i) when
             that returns True (the piece becomes final while the backward pass still runs), else on the current stream;
          2. one small all-reduce of the replicated tail + the scalar slots the caller filled beforehand ([0] loss, [1] status
             flag, [3] the next batch's mask sum; [2] is used here for the clip norm);
          3. Adam on owned_ranges(), clipped by the GLOBAL gradient norm when max_norm > 0, skipped on the device on every rank when
             the summed status flag is non-zero;
          4. all-gather of the updated weights (operand-dtype copy or the f32 masters in place), last piece first -- the order a
             forward pass consumes them in.  gather_async: on `comm`, one event per piece, the current stream does NOT wait
             (the next refresh does, group by group); otherwise on the current stream.

        pipeline (with early_guard; needs `comm`, no clipping): pieces whose Adam and all-gather run on the communication stream
        RIGHT BEHIND their reduce-scatter, while the backward pass still computes the later pieces -- the logit layer's quarter of
        the bytes is then back on every rank long before the step ends, instead of in the next step's prologue.  early_guard: the
        status word (device int32), already final when the first overlapped piece is released; it is summed over the ranks there
        (one tiny all-reduce) and guards EVERY Adam launch of the step, so that the ranks skip together or not at all.

        half: pieces whose reduce-scatter travels as bf16 (each rank's gradient rounded once, the sum in the collective's bf16
        arithmetic, this rank's slice widened back to f32 for Adam) -- half the bytes on the wire for pieces that cannot hide behind
        the backward pass; NOT bit-reproducible against the f32 exchange (opt.bf16_gradient_exchange, off by default).

        Returns (pair, events): pair = [summed loss, summed status flag] (a fresh 2-float tensor), events = {piece: event} or None."""
        cur = torch.cuda.current_stream(self.flat.device)
        clip = bool(max_norm and max_norm > 0)
        pipeline = tuple(pipeline) if (comm is not None and early_guard is not None and not clip) else ()
        buf = self.w16 if self.w16 is not None else self.flat
        guard = None
        events = {}
        sc = self.scalars
        # With a communication stream EVERY collective of the step runs there, in one order (RCCL serialises the collectives of a
        # communicator anyway; so does the one-GPU stand-in this way): the pieces that become final during the backward pass behind
        # their events, the rest behind the point the current stream has reached (the step has joined there).
        joined = False
        on = comm if comm is not None else cur
        with torch.cuda.stream(on):
            for i in range(len(self.pieces)):
                early = comm is not None and wait_piece is not None and not joined and wait_piece(comm.cuda_stream, i)
                if not early and comm is not None and not joined:
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    comm.wait_event(ev)
                    joined = True
                if early and guard is None and early_guard is not None:
                    guard = early_guard[0:1].to(torch.float32)
                    ex._sum(guard)
                if i in half:
                    if self.g16 is None:
                        self.g16 = torch.empty(self.repl_off, dtype=torch.bfloat16, device=self.grad.device)
                    off, n = self.pieces[i]
                    self.g16[off:off + n].copy_(self.grad[off:off + n])
                    ex.reduce_scatter(self.g16, off, n)
                    lo, hi = self.shard(i)
                    self.grad[lo:hi].copy_(self.g16[lo:hi])
                else:
                    ex.reduce_scatter(self.grad, *self.pieces[i])
                if early and i in pipeline and guard is not None:
                    self.adam_owned(lr, betas, eps, step, grad_scale, guard=guard, ranges=[self.shard(i)])
                    ex.all_gather(buf, *self.pieces[i])
                    e = torch.cuda.Event()
                    e.record(comm)
                    events[i] = e
            if comm is not None and not joined:
                ev = torch.cuda.Event()
                ev.record(cur)
                comm.wait_event(ev)
            if clip:
                sc[2:3].copy_(self.shard_sqnorm())
            ex._sum(self.grad[self.repl_off:self.scalars_off + 4])
        if comm is not None:
            cur.wait_stream(comm)
        sq = None
        if clip:
            sq = sc[2:3].clone()
            if self.repl_end > self.repl_off:
                rs = self.scratch[1024:1025]
                check(_lib.load().uic_grad_sqnorm(ptr(self.grad[self.repl_off:self.repl_end]), self.repl_end - self.repl_off,
                                                  ptr(self.scratch), ptr(rs), stream()), "grad_sqnorm")
                sq += rs
        pair = torch.cat([sc[0:1], guard]) if guard is not None else sc[0:2].clone()
        rest = [i for i in range(len(self.pieces)) if i not in events]
        rg = [self.shard(i) for i in rest]
        if self.repl_end > self.repl_off:
            rg.append((self.repl_off, self.repl_end))
        self.adam_owned(lr, betas, eps, step, grad_scale, max_norm if clip else 0.0, sq, guard=pair[1:2], ranges=rg)
        if gather_async and comm is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
            comm.wait_event(ev)
            with torch.cuda.stream(comm):
                for i in reversed(rest):
                    ex.all_gather(buf, *self.pieces[i])
                    e = torch.cuda.Event()
                    e.record(comm)
                    events[i] = e
        else:
            for i in reversed(rest):
                ex.all_gather(buf, *self.pieces[i])
            if events:                                    # (the pipelined pieces' gathers ran on comm)
                cur.wait_stream(comm)
            events = None
        return pair, events

    def gather_masters(self, exchange):
        """All-gather the f32 masters of every piece in place (a collective: every rank calls it), so that each rank's parameters
        -- state_dict(), a checkpoint -- are the full f32 weights again.  Needed only in operand-dtype mode."""
        if self.masters_stale and exchange is not None and exchange.world_size > 1:
            for g in range(len(self.pieces)):
                exchange.all_gather(self.flat, *self.pieces[g])
        self.masters_stale = False

    def bind_grads(self):
        """Make every p.grad the arena view, so autograd accumulates in place."""
        for k, p in self.params.items():
            p.grad = self.grad_views[k]

    def zero_grad(self):
        self.grad.zero_()
        self.bind_grads()

    def adam(self, lr, betas, eps, step, grad_scale=1.0, max_norm=0.0, guard=None):
        """guard (optional, a device word): the update is skipped on the device when it is non-zero -- the status word of the
        persistent kernels, so that a timed-out launch's gradients never reach the weights (include/uic_hip.h)."""
        lib = _lib.load()
        if max_norm and max_norm > 0:
            sq = self.scratch[1024:1025]
            check(lib.uic_grad_sqnorm(ptr(self.grad), self.numel, ptr(self.scratch), ptr(sq), stream()), "grad_sqnorm")
            check(lib.uic_adam_step_clip_guarded(ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.numel,
                                                 lr, betas[0], betas[1], eps, step, grad_scale, float(max_norm), ptr(sq), ptr(guard),
                                                 stream()), "adam_step_clip_guarded")
        else:
            check(lib.uic_adam_step_guarded(ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.numel,
                                            lr, betas[0], betas[1], eps, step, grad_scale, ptr(guard), stream()), "adam_step_guarded")

    def grad_norm(self):
        """Host value of the global gradient L2 norm (diagnostics; synchronises)."""
        sq = self.scratch[1024:1025]
        check(_lib.load().uic_grad_sqnorm(ptr(self.grad), self.numel, ptr(self.scratch), ptr(sq), stream()), "grad_sqnorm")
        return float(sq.item()) ** 0.5


def _get(opt, name, default):
    return getattr(opt, name, default)


class Optim(object):
    def __init__(self, opt, exchange=None):
        self.last_ppl = None
        self.init_i2t(opt)
        self.init_nmt(opt)
        self._step = 0
        self.opt = opt
        self.exchange = exchange
        self.i2t_arena = None
        self.nmt_arena = None
        self._i2t_steps = 0
        self._nmt_steps = 0

    def init_i2t(self, opt):
        self.i2t_train_flag = _get(opt, 'i2t_train_flag', 0)
        self.i2t_eval_flag = _get(opt, 'i2t_eval_flag', 0)
        self.i2t_method = _get(opt, 'i2t_optim', 'adam')
        self.i2t_lr = _get(opt, 'i2t_learning_rate', 4e-4)
        self.i2t_current_lr = self.i2t_lr
        self.i2t_learning_rate_decay_start = _get(opt, 'i2t_learning_rate_decay_start', 0)
        self.i2t_learning_rate_decay_every = _get(opt, 'i2t_learning_rate_decay_every', 3)
        self.i2t_learning_rate_decay_rate = _get(opt, 'i2t_learning_rate_decay_rate', 0.8)
        self.i2t_optim_alpha = _get(opt, 'i2t_optim_alpha', 0.9)
        self.i2t_optim_beta = _get(opt, 'i2t_optim_beta', 0.999)
        self.i2t_optim_epsilon = _get(opt, 'i2t_optim_epsilon', 1e-8)
        self.i2t_max_grad_norm = _get(opt, 'i2t_max_grad_norm', 0)
        self.i2t_weight_decay = _get(opt, 'i2t_weight_decay', 0)

    def init_nmt(self, opt):
        self.nmt_train_flag = _get(opt, 'nmt_train_flag', 0)
        self.nmt_eval_flag = _get(opt, 'nmt_eval_flag', 0)
        self.nmt_method = _get(opt, 'nmt_optim', 'adam')
        self.nmt_lr = _get(opt, 'nmt_learning_rate', 1e-3)
        self.nmt_current_lr = self.nmt_lr
        self.nmt_learning_rate_decay_start = _get(opt, 'nmt_learning_rate_decay_start', 8)
        self.nmt_learning_rate_decay_every = _get(opt, 'nmt_learning_rate_decay_every', 3)
        self.nmt_learning_rate_decay_rate = _get(opt, 'nmt_learning_rate_decay_rate', 0.5)
        self.nmt_optim_alpha = _get(opt, 'nmt_optim_alpha', 0.9)
        self.nmt_optim_beta = _get(opt, 'nmt_optim_beta', 0.999)
        self.nmt_optim_epsilon = _get(opt, 'nmt_optim_epsilon', 1e-8)
        self.nmt_max_grad_norm = _get(opt, 'nmt_max_grad_norm', 5)
        self.nmt_decay_method = _get(opt, 'nmt_decay_method', '')
        self.nmt_weight_decay = _get(opt, 'nmt_weight_decay', 0)
        self.nmt_warmup_steps = _get(opt, 'nmt_warmup_steps', 4000)
        self.nmt_betas = [0.9, 0.98]

    @staticmethod
    def _check_method(kind, method, weight_decay):
        if method != 'adam':
            raise NotImplementedError("only Adam is on the MI355X hot path (%s_optim=%s)" % (kind, method))
        if weight_decay:
            raise NotImplementedError("%s_weight_decay != 0 is not on the MI355X hot path" % kind)

    def set_parameters(self, i2t_model, nmt_model):
        if i2t_model is not None:
            self._check_method('i2t', self.i2t_method, self.i2t_weight_decay)
            names = getattr(i2t_model, 'param_names', None)
            self.i2t_arena = FlatArena(i2t_model, names)
            self.i2t_arena.bind_grads()
        if nmt_model is not None:
            self._check_method('nmt', self.nmt_method, self.nmt_weight_decay)
            # arena order = the order in which uic_nmt_backward makes the gradients final (include/uic_hip.h,
            # uic_nmt_grad_ready_wait): generator first (before the decoder BPTT starts), the decoder side, the encoder last -- a
            # data-parallel run exchanges each piece while the backward pass computes the following ones
            names = getattr(nmt_model, 'param_names', None)
            self.nmt_splits = []
            if names is not None:
                gen = [k for k in names if k.startswith("generator.")]
                dec = [k for k in names if k.startswith("decoder.")]
                enc = [k for k in names if not k.startswith(("generator.", "decoder."))]
                names = gen + dec + enc
            ex = self.exchange
            self.nmt_sharded = bool(ex is not None and ex.world_size > 1 and names is not None and gen and dec and enc and
                                    not _get(self.opt, 'allreduce_exchange', 0))
            if self.nmt_sharded:
                # sharded exchange with the f32 masters as the gathered weights (the pivot model's operand copies are cast from them
                # by uic_nmt_forward_loss itself): reduce-scatter [generator | decoder | encoder], clipped Adam on this rank's third
                # of a third, all-gather in place -- the bytes of one all-reduce, an eighth of the optimizer's work
                self.nmt_arena = FlatArena(nmt_model, names, world=ex.world_size, rank=ex.rank, pieces=[gen, dec, enc])
            else:
                self.nmt_arena = FlatArena(nmt_model, names)
            if names is not None and gen and dec and enc:
                self.nmt_splits = [self.nmt_arena.offsets[dec[0]], self.nmt_arena.offsets[enc[0]]]
            self.nmt_arena.bind_grads()
            if hasattr(nmt_model, 'grad_sink'):
                nmt_model.grad_sink = self.nmt_arena.grad_views      # backward may write the arena in place (see _NmtStep.backward)
                self._nmt_model = nmt_model

    def _exchange(self, arena, splits=None, wait_group=None):
        """Sum the gradient arena over the ranks.  splits / wait_group: the arena's pieces in the order they become final and the
        call that lets a stream wait for piece g (GradientExchange.allreduce_sum_overlapped): all but the last piece travel on
        the communication stream beside the rest of the backward pass."""
        if self.exchange is not None and self.exchange.world_size > 1:
            if splits and wait_group is not None and arena.grad.is_cuda:
                self.exchange.allreduce_sum_overlapped(arena.grad, splits, wait_group)
            else:
                self.exchange.allreduce_sum(arena.grad)

    def step(self, i2t_grad_scale=1.0, nmt_grad_scale=1.0):
        self._step += 1
        if self.i2t_train_flag and self.i2t_arena is not None:
            self._exchange(self.i2t_arena)
            self._i2t_steps += 1
            # (guarded like the pivot model's step below: a captioner trained through Optim.step must not apply a step whose
            # persistent launch timed out)
            ig = None
            if self.i2t_arena.flat.is_cuda:
                ig = _lib.status_words(self.i2t_arena.flat.device)
                if self.exchange is not None and self.exchange.world_size > 1:
                    ig = ig[0:1].float()
                    self.exchange._sum(ig)
            self.last_i2t_guard = ig
            self.i2t_arena.adam(self.i2t_current_lr, (self.i2t_optim_alpha, self.i2t_optim_beta), self.i2t_optim_epsilon,
                                self._i2t_steps, i2t_grad_scale, 0.0, guard=ig)
        if _get(self.opt, 'nmt_train_flag', 0) and self.nmt_arena is not None:
            if self.nmt_decay_method == "noam":
                self.nmt_current_lr = self.nmt_lr * (self.opt.rnn_size ** (-0.5) *
                                                     min(self._step ** (-0.5), self._step * self.nmt_warmup_steps ** (-1.5)))
            # the pivot model's 344 MB: generator (102 MB, final before the decoder BPTT starts) and the decoder side on the
            # communication stream as they become final, the encoder's share on this stream; the clipped Adam below takes the
            # norm of the summed gradient (one pass over the arena after all three pieces)
            m = getattr(self, '_nmt_model', None)
            direct = m is not None and getattr(m, '_sink_written', False)      # the in-place backward ran: its events are recorded
            # The first two pieces may travel beside the rest of the backward pass ONLY when that pass is a launch chain: its
            # persistent launches (csrc/nmt_persist.hip) want one workgroup on EVERY CU and register with a bounded spin, and a CU
            # that holds an RCCL workgroup waiting for a late peer cannot take one -- every rank would then skip the step with a
            # PersistentTimeout.  With persistent launches the exchange starts after the backward pass.
            chain = m is not None and bool(int(getattr(m.engine, 'recurrence', 0)) & _lib.REC_FWD_CHAIN)
            overlap = direct and chain and self.nmt_arena.grad.is_cuda
            wait_g = lambda raw, g: check(_lib.load().uic_nmt_grad_ready_wait(raw, g), "nmt_grad_ready_wait")
            self._nmt_steps += 1
            # the pivot step's persistent launches report a time-out in the status words: the update is then skipped on the
            # device -- on every rank (the flag is summed over them) -- and Trainer.train_nmt raises
            guard = _lib.status_words(self.nmt_arena.flat.device) if self.nmt_arena.flat.is_cuda else None
            if getattr(self, 'nmt_sharded', False):
                a = self.nmt_arena
                if guard is not None:
                    a.scalars[1:2].copy_(guard[0:1])
                else:
                    a.scalars[1:2].zero_()
                if getattr(self, '_comm_stream', None) is None and overlap:
                    self._comm_stream = torch.cuda.Stream(device=a.flat.device)

                def wait_piece(raw, i):
                    if not overlap or i > 1:
                        return False
                    wait_g(raw, i)
                    return True
                pair, _ = a.sharded_step(self.exchange, self.nmt_current_lr, (self.nmt_optim_alpha, self.nmt_optim_beta), self.nmt_optim_epsilon,
                                         self._nmt_steps, nmt_grad_scale, self.nmt_max_grad_norm, wait_piece=wait_piece,
                                         comm=getattr(self, '_comm_stream', None) if overlap else None)
                self.last_pair = pair                   # [summed loss (slot 0, filled by the caller), summed status flag]
                self.last_guard = pair[1:2]
                return
            self._exchange(self.nmt_arena, getattr(self, 'nmt_splits', None) if overlap else None, wait_g)
            if guard is not None and self.exchange is not None and self.exchange.world_size > 1:
                guard = guard[0:1].float()
                self.exchange._sum(guard)
            self.last_guard = guard
            self.nmt_arena.adam(self.nmt_current_lr, (self.nmt_optim_alpha, self.nmt_optim_beta), self.nmt_optim_epsilon,
                                self._nmt_steps, nmt_grad_scale, self.nmt_max_grad_norm, guard=guard)

    def zero_grad(self, nmt_direct=False):
        """nmt_direct (Trainer.train_nmt): the coming backward pass is the in-place one (models/NMT_Models.py, _NmtStep.backward:
        the kernels OVERWRITE every gradient in the arena, the embedding tables included), so the 360 MB fill of the pivot
        model's arena is skipped; the padding between tensors was zeroed when the arena was made and nothing writes it.  Should
        that backward pass take the accumulating path after all, it clears the arena itself first (model._lazy_zero)."""
        if self.i2t_train_flag and self.i2t_arena is not None:
            self.i2t_arena.zero_grad()
        if self.nmt_train_flag and self.nmt_arena is not None:
            m = getattr(self, '_nmt_model', None)
            if nmt_direct and m is not None and m.grad_sink is not None and self.nmt_arena.flat.is_cuda:
                self.nmt_arena.bind_grads()
                m._lazy_zero = self.nmt_arena.grad.zero_
            else:
                self.nmt_arena.zero_grad()
                if m is not None:
                    m._lazy_zero = None
            if m is not None:
                m._sink_written = False

    def update_ScheduledSampling_prob(self, opt, epoch, dp_i2t_model):
        if epoch > opt.scheduled_sampling_start and opt.scheduled_sampling_start >= 0:
            frac = (epoch - opt.scheduled_sampling_start) // opt.scheduled_sampling_increase_every
            dp_i2t_model.ss_prob = min(opt.scheduled_sampling_increase_prob * frac, opt.scheduled_sampling_max_prob)
        return dp_i2t_model

    def update_LearningRate(self, type, epoch):
        if type == 'i2t':
            if epoch > self.i2t_learning_rate_decay_start and self.i2t_learning_rate_decay_start >= 0:
                frac = (epoch - self.i2t_learning_rate_decay_start) // self.i2t_learning_rate_decay_every
                self.i2t_current_lr = self.i2t_lr * self.i2t_learning_rate_decay_rate ** frac
            else:
                self.i2t_current_lr = self.i2t_lr
        if type == 'nmt':
            if epoch > self.nmt_learning_rate_decay_start and self.nmt_learning_rate_decay_start >= 0:
                self.nmt_current_lr = self.nmt_lr * self.nmt_learning_rate_decay_rate
            else:
                self.nmt_current_lr = self.nmt_lr
