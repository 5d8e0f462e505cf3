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
    """All parameters of a module re-homed into one flat f32 tensor (plus same-layout grad / Adam arenas)."""

    def __init__(self, module, names=None):
        params = dict(module.named_parameters())
        self.names = list(names) if names is not None else list(params.keys())
        self.offsets = {}
        off = 0
        for k in self.names:
            self.offsets[k] = off
            off += (params[k].numel() + 63) // 64 * 64          # 256-byte aligned blocks
        dev = params[self.names[0]].device
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros(1024 + 8, dtype=torch.float32, device=dev)    # sqnorm partials + result
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
            self._exchange(self.nmt_arena, getattr(self, 'nmt_splits', None) if direct else None,
                           lambda raw, g: check(_lib.load().uic_nmt_grad_ready_wait(raw, g), "nmt_grad_ready_wait"))
            self._nmt_steps += 1
            # the pivot step's persistent launches (csrc/nmt_persist.hip) report a time-out in the status words: the update is
            # then skipped on the device -- on every rank (the flag is summed over them) -- and Trainer.train_nmt raises
            guard = None
            if self.nmt_arena.flat.is_cuda:
                guard = _lib.status_words(self.nmt_arena.flat.device)
                if self.exchange is not None and self.exchange.world_size > 1:
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
