"""Criteria of the captioner with the reference's call signatures (P/misc/criterion.py:104-159),
computed by libuic_hip.so on device tensors."""
import ctypes as C

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
