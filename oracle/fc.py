"""Plain-torch fp32 CPU restatement of the reference FC captioner (BASELINE config 1).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Restates ``FCModel_NMT`` + ``LSTMCore``
(``P/models/FCModel_NMT.py:21-51, 54-217``); pinned by ``tests/golden/fc_*.npz`` generated from the reference.

state_dict keys: img_embed.{weight [E,Dfc], bias}, core.i2h.{weight [5H,E], bias}, core.h2h.{weight [5H,H], bias},
embed.weight [V1,E], logit.{weight [V1,H], bias}.

Dropout: ``drop`` is None or {'out': [S,N,H]} multiplicative masks for the S executed core steps -- LSTMCore drops
``next_h`` BEFORE it becomes the recurrent state (:47-50).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


def lstm_core(W: Weights, xt: Tensor, h: Tensor, c: Tensor, mask: Optional[Tensor] = None):
    """LSTMCore.forward, P/models/FCModel_NMT.py:32-51 (maxout LSTM)."""
    H = h.shape[1]
    s = F.linear(xt, W["core.i2h.weight"], W["core.i2h.bias"]) + F.linear(h, W["core.h2h.weight"], W["core.h2h.bias"])
    sig = torch.sigmoid(s[:, :3 * H])
    i, f, o = sig[:, :H], sig[:, H:2 * H], sig[:, 2 * H:3 * H]
    g = torch.max(s[:, 3 * H:4 * H], s[:, 4 * H:5 * H])
    c2 = f * c + i * g
    h2 = o * torch.tanh(c2)
    if mask is not None:
        h2 = h2 * mask
    return h2, c2


def forward_logprobs(W: Weights, fc_feats: Tensor, seq: Tensor, drop=None):
    """FCModel_NMT._forward, P/models/FCModel_NMT.py:89-124.  seq = labels [N, L+2]; returns [N, L+1, V1]
    (``outputs[:, 1:]``), zero-filled after the early break (:115-116)."""
    N = fc_feats.shape[0]
    S = seq.shape[1]
    H = W["core.h2h.weight"].shape[1]
    V1 = W["logit.weight"].shape[0]
    h, c = torch.zeros(N, H), torch.zeros(N, H)
    outs = []
    for i in range(S):
        if i == 0:
            xt = F.linear(fc_feats, W["img_embed.weight"], W["img_embed.bias"])
        else:
            if i >= 2 and int(seq[:, i - 1].sum()) == 0:
                break
            xt = W["embed.weight"][seq[:, i - 1]]
        h, c = lstm_core(W, xt, h, c, None if drop is None else drop["out"][i])
        outs.append(F.log_softmax(F.linear(h, W["logit.weight"], W["logit.bias"]), dim=1))
    out = torch.stack(outs, 1)
    if out.shape[1] < S:
        out = torch.cat([out, torch.zeros(N, S - out.shape[1], V1)], 1)
    return out[:, 1:]


def sample(W: Weights, fc_feats: Tensor, seq_length: int, sample_max: int = 1, temperature: float = 1.0,
           forced_tokens: Optional[Tensor] = None, generator=None):
    """FCModel_NMT._sample (beam_size = 1), P/models/FCModel_NMT.py:164-217.  Returns [N, L+1] tensors.
    Quirks kept: the embedding of the RAW sampled token (before the finished-row masking) feeds the next step,
    and the all-finished break happens before anything is written for that step."""
    N = fc_feats.shape[0]
    H = W["core.h2h.weight"].shape[1]
    h, c = torch.zeros(N, H), torch.zeros(N, H)
    seq = torch.zeros(N, seq_length + 1, dtype=torch.long)
    seq_logp = torch.zeros(N, seq_length + 1)
    logp = None
    unfinished = None
    for t in range(seq_length + 2):
        if t == 0:
            xt = F.linear(fc_feats, W["img_embed.weight"], W["img_embed.bias"])
        else:
            if t == 1:
                it = torch.zeros(N, dtype=torch.long)
            elif sample_max:
                lp, it = torch.max(logp, 1)
            else:
                if forced_tokens is not None:
                    it = forced_tokens[:, t - 2].clone()
                else:
                    prob = torch.exp(logp if temperature == 1.0 else logp / temperature)
                    it = torch.multinomial(prob, 1, generator=generator).view(-1)
                lp = logp.gather(1, it.unsqueeze(1)).view(-1)
            xt = W["embed.weight"][it]
        if t >= 2:
            unfinished = (it > 0) if t == 2 else unfinished & (it > 0)
            if int(unfinished.sum()) == 0:
                break
            seq[:, t - 2] = it * unfinished.long()
            seq_logp[:, t - 2] = lp
        h, c = lstm_core(W, xt, h, c)
        logp = F.log_softmax(F.linear(h, W["logit.weight"], W["logit.bias"]), dim=1)
    return seq, seq_logp


def xe_loss_and_grads(W: Weights, fc_feats, labels, masks, drop=None):
    from .topdown import lm_criterion
    Wg = {k: v.detach().clone().requires_grad_(True) for k, v in W.items()}
    logp = forward_logprobs(Wg, fc_feats, labels, drop)
    loss = lm_criterion(logp, labels[:, 1:], masks[:, 1:])
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Wg.items()}
    return loss.detach(), grads, logp.detach()


def init_weights(V1: int, E: int, H: int, Dfc: int, seed: int = 0) -> Weights:
    g = torch.Generator().manual_seed(seed)

    def u(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    return {
        "img_embed.weight": u((E, Dfc), Dfc), "img_embed.bias": u((E,), Dfc),
        "core.i2h.weight": u((5 * H, E), E), "core.i2h.bias": u((5 * H,), E),
        "core.h2h.weight": u((5 * H, H), H), "core.h2h.bias": u((5 * H,), H),
        "embed.weight": (torch.rand(V1, E, generator=g) * 2 - 1) * 0.1,
        "logit.weight": (torch.rand(V1, H, generator=g) * 2 - 1) * 0.1, "logit.bias": torch.zeros(V1),
    }


def sample_beam(W: Weights, fc_feats: Tensor, seq_length: int, beam_size: int, decoding_constraint: int = 0, max_ppl: int = 0):
    """FCModel_NMT._sample_beam (P/models/FCModel_NMT.py:136-162) over CaptionModel.beam_search
    (``oracle.topdown.beam_search_core``): two warm-up core steps (image embedding, then <bos>), then the search with
    get_logprobs_state = embed -> core -> log_softmax(logit) (:126-134).  Returns [N, L] tensors."""
    from .topdown import beam_search_core
    N = fc_feats.shape[0]
    H = W["core.h2h.weight"].shape[1]
    B, L = beam_size, seq_length
    seq = torch.zeros(N, L, dtype=torch.long)
    seq_logp = torch.zeros(N, L)

    def step_fn(it, state):
        h, c = lstm_core(W, W["embed.weight"][it], state[0][0], state[1][0])
        return F.log_softmax(F.linear(h, W["logit.weight"], W["logit.bias"]), dim=1), (h.unsqueeze(0), c.unsqueeze(0))
    for k in range(N):
        xt = F.linear(fc_feats[k:k + 1], W["img_embed.weight"], W["img_embed.bias"]).expand(B, -1)
        h, c = lstm_core(W, xt, torch.zeros(B, H), torch.zeros(B, H))
        logprobs, state = step_fn(torch.zeros(B, dtype=torch.long), (h.unsqueeze(0), c.unsqueeze(0)))
        done = beam_search_core(step_fn, logprobs, state, L, B, decoding_constraint, max_ppl)
        seq[k] = done[0]["seq"]
        seq_logp[k] = done[0]["logps"]
    return seq, seq_logp
