"""Criteria of the captioner with the reference's call signatures (P/misc/criterion.py:104-159),
computed by libuic_hip.so on device tensors."""
import ctypes as C
import math
import time

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, ptr, stream


class _LMCriterionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, target, mask):
        lib = _lib.load()
        N, T, V1 = logp.shape
        logp = logp.contiguous()
        target = target[:, :T].contiguous()
        mask = mask[:, :T].contiguous().float()
        loss = torch.empty((), dtype=torch.float32, device=logp.device)
        scratch = torch.empty(2 * N * T + 2, dtype=torch.float32, device=logp.device)
        check(lib.uic_lm_criterion(N, T, V1, ptr(logp), ptr(target), target.shape[1], ptr(mask), mask.shape[1],
                                   loss.data_ptr(), ptr(scratch), None, 1.0, stream()), "lm_criterion")
        ctx.save_for_backward(logp, target, mask)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        logp, target, mask = ctx.saved_tensors
        N, T, V1 = logp.shape
        loss = torch.empty((), dtype=torch.float32, device=logp.device)
        scratch = torch.empty(2 * N * T + 2, dtype=torch.float32, device=logp.device)
        dlogp = torch.empty_like(logp)
        check(lib.uic_lm_criterion(N, T, V1, ptr(logp), ptr(target), target.shape[1], ptr(mask), mask.shape[1],
                                   loss.data_ptr(), ptr(scratch), ptr(dlogp), 1.0, stream()), "lm_criterion backward")
        return dlogp * g, None, None


class LanguageModelCriterion(nn.Module):
    """forward(input [N,T,V1] log-probs, target [N,>=T], mask [N,>=T]) -> scalar (criterion.py:143-159)."""

    def __init__(self, opt=None):
        super(LanguageModelCriterion, self).__init__()
        self.caption_model = getattr(opt, 'caption_model', 'topdown')

    def forward(self, input, target, mask):
        if 'stackcap' in self.caption_model:
            raise NotImplementedError("stackcap is outside the MI355X hot path")
        return _LMCriterionFn.apply(input, target, mask)


class _RewardCriterionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, seq, reward):
        lib = _lib.load()
        N, L = logp.shape
        logp = logp.contiguous().float()
        seq = seq.contiguous()
        reward = reward.contiguous().float()
        loss = torch.empty((), dtype=torch.float32, device=logp.device)
        dlogp = torch.empty_like(logp)
        check(lib.uic_reward_criterion(N, L, ptr(logp), ptr(seq), ptr(reward), loss.data_ptr(), ptr(dlogp), stream()),
              "reward_criterion")
        ctx.save_for_backward(dlogp)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogp,) = ctx.saved_tensors
        return dlogp * g, None, None


class RewardCriterion(nn.Module):
    """forward(input [N,L] sampled log-probs, seq [N,L], reward [N,L]) -> scalar (criterion.py:104-124)."""

    def __init__(self):
        super(RewardCriterion, self).__init__()

    def forward(self, input, seq, reward):
        return _RewardCriterionFn.apply(input, seq, reward)


# ---------------------------------------------------------------- pivot NMT (P/misc/criterion.py:47-101,126-205)
class Statistics(object):
    """Accumulator for loss statistics (P/misc/criterion.py:47-101): accuracy, perplexity, elapsed time."""

    def __init__(self, loss=0, n_words=0, n_correct=0):
        self.loss = loss
        self.n_words = n_words
        self.n_correct = n_correct
        self.n_src_words = 0
        self.start_time = time.time()

    def update(self, stat):
        self.loss += stat.loss
        self.n_words += stat.n_words
        self.n_correct += stat.n_correct

    def accuracy(self):
        return 100 * (float(self.n_correct) / self.n_words)

    def ppl(self):
        return math.exp(min(self.loss / self.n_words, 100))

    def elapsed_time(self):
        return time.time() - self.start_time


class _NMTCriterion(nn.Module):
    """Marker for nn.NLLLoss(weight[PAD]=0, size_average=False) (P/misc/criterion.py:126-136).  The weighted NLL is
    computed inside uic_nmt_forward_loss next to the generator GEMM; this object only carries the vocabulary size."""

    def __init__(self, vocabSize):
        super(_NMTCriterion, self).__init__()
        self.vocabSize = vocabSize
        self.register_buffer('weight', torch.ones(vocabSize))
        self.weight[0] = 0

    def forward(self, scores, target):
        raise RuntimeError("the NMT NLL is fused into uic_nmt_forward_loss; call NMT_loss(loader, batch, outputs, attns)")


def NMTCriterion(vocabSize, opt):
    return _NMTCriterion(vocabSize)


class NMT_loss(nn.Module):
    """P/misc/criterion.py:161-205.  `outputs` must come from models.NMT_Models.NMTModel.forward, which already ran the
    generator, the loss and the `score` counters on the device (outputs.uic_loss / outputs.uic_stats)."""

    def __init__(self, opt, generator, crit, eval=False):
        super(NMT_loss, self).__init__()
        for name in ('lambda_coverage', 'lambda_fertility', 'lambda_exhaust'):
            if getattr(opt, name, 0):
                raise NotImplementedError("opt.%s is outside the MI355X NMT hot path" % name)
        self.generator = generator
        self.crit = crit
        self.batch_size = getattr(opt, 'batch_size', None)
        self._total_stats = Statistics()
        self._report_stats = Statistics()
        self._pending = []

    # The counters of a batch live on the device until somebody looks at the statistics: reading them inside forward()
    # (the reference's loss.data[0], criterion.py:199-203) would stall the host between the forward and the backward
    # enqueue of every step and leave the GPU idle meanwhile.
    def _flush(self):
        for loss_t, stats_t, reset in self._pending:
            num_correct, num_words = [int(x) for x in stats_t.tolist()]            # one small D2H
            stats = Statistics(float(loss_t), num_words, num_correct)
            self._total_stats.update(stats)
            self._report_stats.update(stats)
            if reset:
                self._total_stats = Statistics()
                self._report_stats = Statistics()
        self._pending = []

    @property
    def total_stats(self):
        self._flush()
        return self._total_stats

    @total_stats.setter
    def total_stats(self, v):
        self._flush()
        self._total_stats = v

    @property
    def report_stats(self):
        self._flush()
        return self._report_stats

    @report_stats.setter
    def report_stats(self, v):
        self._flush()
        self._report_stats = v

    def forward(self, loader, batch, outputs, attns):
        if not hasattr(outputs, 'uic_loss'):
            raise RuntimeError("NMT_loss needs the `outputs` tensor returned by the HIP NMTModel.forward (no eager fallback)")
        nmt_stats_reset = bool(loader is not None and loader.nmt_batchIdx > len(loader.nmt_trainData))
        loss_t = outputs.uic_loss
        self._pending.append((loss_t.detach(), outputs.uic_stats, nmt_stats_reset))
        if len(self._pending) > 64:          # nobody is reading: keep the backlog bounded
            self._flush()
        return loss_t
