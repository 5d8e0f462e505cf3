"""Plain-torch fp32 CPU restatement of the reference TopDown captioner hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function cites the
reference lines it restates; ``P/`` = ``/root/reference/pivot_based_eccv2018/``.

Weights travel as a dict keyed by the reference's ``state_dict`` names
(``P/models/AttModel.py:56-92, 422-428, 530-536, 686-690``):

    embed.0.weight [V1,E]           fc_embed.0.{weight [H,Dfc], bias}
    att_embed.0.{weight [H,D], bias}   (use_bn=0; with use_bn>=1 the Linear is
                                        att_embed.1 and att_embed.0 is BN1d(D);
                                        use_bn=2 adds BN1d(H) at att_embed.4)
    ctx2att.{weight [A,H], bias}    logit.{weight [V1,H], bias}
    core.att_lstm.{weight_ih [4H,E+2H], weight_hh [4H,H], bias_ih, bias_hh}
    core.lang_lstm.{weight_ih [4H,2H], weight_hh [4H,H], bias_ih, bias_hh}
    core.attention.h2att.{weight [A,H], bias}
    core.attention.alpha_net.{weight [1,A], bias [1]}

Dropout is explicit: ``drop`` is None (eval mode / p = 0) or a dict of
multiplicative masks already scaled by 1/(1-p):
    'embed' [T,N,E]   'fc' [N,H]   'att' [N,R,H]   'out' [T,N,H]
so a device kernel that exports its own masks can be checked in training mode.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


def _att_linear_key(use_bn: int) -> str:
    # P/models/AttModel.py:79-84 -- BN1d (if any) sits in front of the Linear
    return "att_embed.1" if use_bn else "att_embed.0"


def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor,
              b_ih: Tensor, b_hh: Tensor) -> Tuple[Tensor, Tensor]:
    """nn.LSTMCell as used at P/models/AttModel.py:426-427,434,441.

    gates = W_ih x + b_ih + W_hh h + b_hh, chunk order (i, f, g, o);
    c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c').
    """
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, dim=1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c2 = f * c + i * g
    h2 = o * torch.tanh(c2)
    return h2, c2


def attention(W: Weights, h: Tensor, att_feats: Tensor, p_att_feats: Tensor,
              att_masks: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """Attention.forward, P/models/AttModel.py:538-558.  Returns (att_res, weight)."""
    att_h = F.linear(h, W["core.attention.h2att.weight"], W["core.attention.h2att.bias"])
    dot = torch.tanh(p_att_feats + att_h.unsqueeze(1))                 # :543-546
    dot = F.linear(dot, W["core.attention.alpha_net.weight"],
                   W["core.attention.alpha_net.bias"]).squeeze(2)      # :547-549
    weight = F.softmax(dot, dim=1)                                     # :551
    if att_masks is not None:                                          # :552-554
        weight = weight * att_masks.float()
        weight = weight / weight.sum(1, keepdim=True)
    att_res = torch.bmm(weight.unsqueeze(1), att_feats).squeeze(1)     # :555-556
    return att_res, weight


def prepare_feature(W: Weights, fc_feats: Tensor, att_feats: Tensor,
                    att_masks: Optional[Tensor], drop=None, use_bn: int = 0,
                    training: bool = False, bn_eps: float = 1e-5):
    """AttModel._prepare_feature + clip_att + pack_wrapper, P/models/AttModel.py:30-53,99-117.

    pack_wrapper runs att_embed only on the first ``len_n = att_masks[n].sum()``
    regions of row n and zero-pads the rest; clip_att truncates R to max(len).
    """
    if att_masks is not None:                                          # clip_att :99-105
        max_len = int(att_masks.long().sum(1).max())
        att_feats = att_feats[:, :max_len].contiguous()
        att_masks = att_masks[:, :max_len].contiguous()
    fc = torch.relu(F.linear(fc_feats, W["fc_embed.0.weight"], W["fc_embed.0.bias"]))
    if drop is not None:
        fc = fc * drop["fc"]
    N, R, D = att_feats.shape
    lk = _att_linear_key(use_bn)
    if att_masks is not None:
        lens = att_masks.long().sum(1)
        valid = (torch.arange(R)[None, :] < lens[:, None])            # packed rows
    else:
        valid = torch.ones(N, R, dtype=torch.bool)
    x = att_feats[valid]                                               # [sum len, D]
    if use_bn:
        if training and "att_embed.0.num_batches_tracked" in W:
            W["att_embed.0.num_batches_tracked"] += 1
        x = F.batch_norm(x, W.get("att_embed.0.running_mean"), W.get("att_embed.0.running_var"),
                         W["att_embed.0.weight"], W["att_embed.0.bias"], training, 0.1, bn_eps)
    y = torch.relu(F.linear(x, W[lk + ".weight"], W[lk + ".bias"]))
    if drop is not None:
        y = y * drop["att"][:, :R][valid]
    if use_bn == 2:
        if training and "att_embed.4.num_batches_tracked" in W:
            W["att_embed.4.num_batches_tracked"] += 1
        y = F.batch_norm(y, W.get("att_embed.4.running_mean"), W.get("att_embed.4.running_var"),
                         W["att_embed.4.weight"], W["att_embed.4.bias"], training, 0.1, bn_eps)
    att = torch.zeros(N, R, y.shape[1], dtype=y.dtype)
    att = att.masked_scatter(valid.unsqueeze(2).expand_as(att), y) if att_masks is not None \
        else y.view(N, R, -1)
    p_att = F.linear(att, W["ctx2att.weight"], W["ctx2att.bias"])      # :115
    return fc, att, p_att, att_masks


def core_step(W: Weights, xt: Tensor, fc: Tensor, att: Tensor, p_att: Tensor, state,
              att_masks: Optional[Tensor], out_mask: Optional[Tensor] = None):
    """TopDownCore.forward, P/models/AttModel.py:430-446.  state = (h[2,N,H], c[2,N,H])."""
    prev_h = state[0][1]
    x1 = torch.cat([prev_h, fc, xt], 1)                                # :432
    h_att, c_att = lstm_cell(x1, state[0][0], state[1][0],
                             W["core.att_lstm.weight_ih"], W["core.att_lstm.weight_hh"],
                             W["core.att_lstm.bias_ih"], W["core.att_lstm.bias_hh"])
    att_res, alpha = attention(W, h_att, att, p_att, att_masks)        # :436
    x2 = torch.cat([att_res, h_att], 1)                                # :438
    h_lang, c_lang = lstm_cell(x2, state[0][1], state[1][1],
                               W["core.lang_lstm.weight_ih"], W["core.lang_lstm.weight_hh"],
                               W["core.lang_lstm.bias_ih"], W["core.lang_lstm.bias_hh"])
    out = h_lang if out_mask is None else h_lang * out_mask            # :443
    new_state = (torch.stack([h_att, h_lang]), torch.stack([c_att, c_lang]))
    return out, new_state, dict(h_att=h_att, c_att=c_att, alpha=alpha, att_res=att_res,
                                h_lang=h_lang, c_lang=c_lang)


def embed(W: Weights, it: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """self.embed = Embedding + ReLU + Dropout, P/models/AttModel.py:73-75,160."""
    xt = torch.relu(W["embed.0.weight"][it])
    return xt if mask is None else xt * mask


def _logit_indices(W):
    return sorted(int(k.split(".")[1]) for k in W if k.startswith("logit.") and k.endswith(".weight") and k.count(".") == 2)


def logit_final_key(W, what: str = "weight") -> str:
    """state_dict key of the vocabulary layer: 'logit.weight', or 'logit.{3(n-1)}.weight' when logit_layers = n > 1."""
    return "logit." + what if "logit.weight" in W else "logit.%d.%s" % (_logit_indices(W)[-1], what)


def logit_final_weight(W) -> Tensor:
    """The vocabulary layer's weight [V1, H]: `logit.weight`, or the last Linear of the Sequential when logit_layers > 1."""
    return W["logit.weight"] if "logit.weight" in W else W["logit.%d.weight" % _logit_indices(W)[-1]]


def logit_layer(W: Weights, x: Tensor, masks=None) -> Tensor:
    """self.logit (P/models/AttModel.py:86-91): one Linear, or logit_layers - 1 blocks [Linear(H, H), ReLU, Dropout(0.5)]
    followed by the vocabulary Linear.  ``masks``: per hidden block, the multiplicative dropout mask [N, H] (train mode)."""
    if "logit.weight" in W:
        return F.linear(x, W["logit.weight"], W["logit.bias"])
    idx = _logit_indices(W)
    for j, i in enumerate(idx[:-1]):
        x = torch.relu(F.linear(x, W["logit.%d.weight" % i], W["logit.%d.bias" % i]))
        if masks is not None:
            x = x * masks[j]
    return F.linear(x, W["logit.%d.weight" % idx[-1]], W["logit.%d.bias" % idx[-1]])


def logprobs_step(W: Weights, it: Tensor, fc, att, p_att, att_masks, state,
                  embed_mask=None, out_mask=None, logit_masks=None):
    """AttModel.get_logprobs_state, P/models/AttModel.py:158-165."""
    xt = embed(W, it, embed_mask)
    out, state, aux = core_step(W, xt, fc, att, p_att, state, att_masks, out_mask)
    logp = F.log_softmax(logit_layer(W, out, logit_masks), dim=1)
    return logp, state, aux


def forward_logprobs(W: Weights, fc_feats: Tensor, att_feats: Tensor, seq: Tensor,
                     att_masks: Optional[Tensor] = None, drop=None, use_bn: int = 0,
                     training: bool = False, return_aux: bool = False, ss=None):
    """AttModel._forward, P/models/AttModel.py:119-156.

    seq is labels [N, L+2]; returns log-probs [N, L+1, V1], zero-filled after the
    early break at the first all-zero label column i >= 1 (:148-151).

    Scheduled sampling (:130-143, training mode and step i >= 1 only): ``ss = {'prob': p}`` draws exactly as the
    reference does (``uniform_`` over the batch, then ``torch.multinomial`` of exp(previous log-probs) for EVERY row
    whenever at least one row is selected), from torch's global CPU generator, so seeding it reproduces the
    reference call for call.  ``ss = {'prob': p, 'mask': bool [T, N], 'tokens': int64 [N, T]}`` replays somebody
    else's decisions and draws (the device's) instead.  The chosen inputs are returned in aux['inputs'].
    """
    N = fc_feats.shape[0]
    T = seq.shape[1] - 1
    H = logit_final_weight(W).shape[1]
    V1 = logit_final_weight(W).shape[0]
    fc, att, p_att, masks = prepare_feature(W, fc_feats, att_feats, att_masks, drop, use_bn, training)
    state = (torch.zeros(2, N, H), torch.zeros(2, N, H))               # init_hidden :94-97
    outputs = []
    auxes = []
    inputs = []
    for i in range(T):
        it = seq[:, i].clone()
        if training and i >= 1 and ss is not None and ss["prob"] > 0.0:  # :130-143
            if "mask" in ss:
                sample_mask = ss["mask"][i].bool()
            else:
                sample_mask = torch.empty(N).uniform_(0, 1) < ss["prob"]
            if int(sample_mask.sum()) != 0:
                sample_ind = sample_mask.nonzero().view(-1)
                if "tokens" in ss:
                    drawn = ss["tokens"][:, i]
                else:
                    drawn = torch.multinomial(torch.exp(outputs[-1].detach()), 1).view(-1)
                it.index_copy_(0, sample_ind, drawn.index_select(0, sample_ind))
        if i >= 1 and int(seq[:, i].sum()) == 0:                       # :151
            break
        em = None if drop is None else drop["embed"][i]
        om = None if drop is None else drop["out"][i]
        lm = None if drop is None or "logit" not in drop else [m[i] for m in drop["logit"]]
        logp, state, aux = logprobs_step(W, it, fc, att, p_att, masks, state, em, om, lm)
        outputs.append(logp)
        auxes.append(aux)
        inputs.append(it)
    out = torch.stack(outputs, 1)
    if out.shape[1] < T:
        out = torch.cat([out, torch.zeros(N, T - out.shape[1], V1)], 1)
    if return_aux:
        return out, dict(fc=fc, att=att, p_att=p_att, steps=auxes, inputs=torch.stack(inputs, 1))
    return out


def lm_criterion(logp: Tensor, target: Tensor, mask: Tensor) -> Tensor:
    """LanguageModelCriterion.xe_loss, P/misc/criterion.py:143-150."""
    target = target[:, :logp.shape[1]]
    mask = mask[:, :logp.shape[1]]
    out = -logp.gather(2, target.unsqueeze(2)).squeeze(2) * mask
    return out.sum() / mask.sum()


def reward_criterion(logp: Tensor, seq: Tensor, reward: Tensor) -> Tensor:
    """RewardCriterion.forward, P/misc/criterion.py:117-124."""
    mask = (seq > 0).float()
    mask = torch.cat([mask.new_ones(mask.shape[0], 1), mask[:, :-1]], 1).reshape(-1)
    out = -logp.reshape(-1) * reward.reshape(-1) * mask
    return out.sum() / mask.sum()


def sample(W: Weights, fc_feats: Tensor, att_feats: Tensor, att_masks: Optional[Tensor],
           seq_length: int, sample_max: int = 1, temperature: float = 1.0,
           decoding_constraint: int = 0, use_bn: int = 0, generator=None,
           forced_tokens: Optional[Tensor] = None, drop=None):
    """AttModel._sample (beam_size = 1), P/models/AttModel.py:198-253.

    ``forced_tokens`` [N, L] replaces the multinomial draw so a device sampler's
    own draws can be scored by this oracle (sample_max = 0 only).  ``drop`` (training-mode sampling pass of
    the self-critical step, P/trainer.py:167) uses the same mask layout as forward_logprobs.
    """
    N = fc_feats.shape[0]
    H = logit_final_weight(W).shape[1]
    fc, att, p_att, masks = prepare_feature(W, fc_feats, att_feats, att_masks, drop, use_bn, False)
    state = (torch.zeros(2, N, H), torch.zeros(2, N, H))
    seq = torch.zeros(N, seq_length, dtype=torch.long)
    lps = []
    it = torch.zeros(N, dtype=torch.long)
    unfinished = None
    for t in range(seq_length + 1):
        em = None if drop is None or t >= seq_length else drop["embed"][t]
        om = None if drop is None or t >= seq_length else drop["out"][t]
        lm = None if drop is None or t >= seq_length or "logit" not in drop else [m[t] for m in drop["logit"]]
        logp, state, _ = logprobs_step(W, it, fc, att, p_att, masks, state, em, om, lm)
        if decoding_constraint and t > 0:                              # :220-223
            tmp = torch.zeros_like(logp)
            tmp.scatter_(1, seq[:, t - 1].unsqueeze(1), float("-inf"))
            logp = logp + tmp
        if t == seq_length:
            break
        if sample_max:
            lp, it = torch.max(logp, 1)                                # :229
        else:
            if forced_tokens is not None:
                it = forced_tokens[:, t].clone()
            else:
                prob = torch.exp(logp if temperature == 1.0 else logp / temperature)
                it = torch.multinomial(prob, 1, generator=generator).view(-1)
            lp = logp.gather(1, it.unsqueeze(1)).view(-1)
        unfinished = (it > 0) if t == 0 else unfinished & (it > 0)     # :242-246
        it = it * unfinished.long()
        seq[:, t] = it
        lps.append(lp)
        if int(unfinished.sum()) == 0:
            break
    seq_logp = torch.stack(lps, 1)
    if seq_logp.shape[1] < seq_length:
        seq_logp = torch.cat([seq_logp, torch.zeros(N, seq_length - seq_logp.shape[1])], 1)
    return seq, seq_logp


def beam_search_core(step_fn, logprobs: Tensor, state, seq_length: int, beam_size: int, decoding_constraint: int = 0,
                     max_ppl: int = 0):
    """CaptionModel.beam_search with group_size = 1 (P/models/CaptionModel.py:33-177) for ONE image.
    ``step_fn(it [B], state) -> (logprobs [B, V1], state)`` is the model's get_logprobs_state; ``state`` a tuple of
    [layers, B, H] tensors.  Returns the done beams, best first.

    Kept exactly: step 0 expands beam 0 only (:64-66); candidates are enumerated word-rank-major, beam-minor and sorted
    by joint log-prob with a stable sort (:67-74); the last vocabulary index gets -1000 (:133) and, with
    decoding_constraint, the previous word -inf (:130-131) BEFORE ranking, and the modified value is what is recorded as
    the step's log-prob (:98); a beam that emitted 0 -- or any beam at the last step -- is copied to the done list and
    its running sum set to -1000 but it keeps being expanded (:147-161); done beams are ranked by p (p / length with
    max_ppl) with a stable sort (:174-176)."""
    B, L = beam_size, seq_length
    beam_seq = torch.zeros(L, B, dtype=torch.long)
    beam_lp = torch.zeros(L, B)
    beam_sum = torch.zeros(B)
    done = []
    for t in range(L):
        lpf = logprobs.clone()
        if decoding_constraint and t > 0:
            lpf.scatter_(1, beam_seq[t - 1].unsqueeze(1), float("-inf"))
        lpf[:, -1] = lpf[:, -1] - 1000
        ys, ix = torch.sort(lpf, 1, True)
        cands = []
        rows = 1 if t == 0 else B
        for c in range(min(B, ys.shape[1])):
            for q in range(rows):
                cands.append(dict(c=int(ix[q, c]), q=q, p=float(beam_sum[q]) + float(ys[q, c]), r=float(lpf[q, ix[q, c]])))
        cands = sorted(cands, key=lambda x: -x["p"])
        new_state = tuple(x.clone() for x in state)
        prev_seq, prev_lp = beam_seq[:t].clone(), beam_lp[:t].clone()
        for vix in range(B):
            v = cands[vix]
            if t >= 1:
                beam_seq[:t, vix] = prev_seq[:, v["q"]]
                beam_lp[:t, vix] = prev_lp[:, v["q"]]
            for k in range(len(state)):
                new_state[k][:, vix] = state[k][:, v["q"]]
            beam_seq[t, vix] = v["c"]
            beam_lp[t, vix] = v["r"]
            beam_sum[vix] = v["p"]
        state = new_state
        for vix in range(B):
            if int(beam_seq[t, vix]) == 0 or t == L - 1:
                p = float(beam_sum[vix])
                done.append(dict(seq=beam_seq[:, vix].clone(), logps=beam_lp[:, vix].clone(),
                                 p=p / (t + 1) if max_ppl else p))
                beam_sum[vix] = -1000
        logprobs, state = step_fn(beam_seq[t].clone(), state)
    return sorted(done, key=lambda x: -x["p"])[:B]


def diverse_beam_search_core(step_fn, logprobs: Tensor, state, seq_length: int, beam_size: int, group_size: int,
                             diversity_lambda: float = 0.5, decoding_constraint: int = 0, max_ppl: int = 0):
    """CaptionModel.beam_search with group_size > 1 (P/models/CaptionModel.py:36-45,100-177) for ONE image: group g runs
    ``bdash = beam_size // group_size`` beams one step behind group g-1 (:125-127); before ranking, every word chosen
    at the same local step by a beam of an EARLIER group costs diversity_lambda per choosing beam (:36-45; the rows of
    the earlier groups' tables as re-threaded up to this moment), the recorded per-step log-prob is the un-penalised one
    (:72,98) while the running sum uses the penalised one (:71).  ``logprobs`` / ``state`` are the first step's for
    bdash rows (every group starts from the same replicated image).  Returns the concatenation of each group's done
    beams, best first within a group (:174-176) -- so entry 0 is the best beam of group 0, which never sees a penalty."""
    G, B, L = group_size, beam_size // group_size, seq_length
    seqs = [torch.zeros(L, B, dtype=torch.long) for _ in range(G)]
    lps = [torch.zeros(L, B) for _ in range(G)]
    sums = [torch.zeros(B) for _ in range(G)]
    done = [[] for _ in range(G)]
    states = [tuple(x.clone() for x in state) for _ in range(G)]
    lp_tab = [logprobs.clone() for _ in range(G)]
    for t in range(L + G - 1):
        for g in range(G):
            if not (g <= t <= L + g - 1):
                continue
            lt = t - g
            lpf = lp_tab[g].clone()
            if decoding_constraint and lt > 0:
                lpf.scatter_(1, seqs[g][lt - 1].unsqueeze(1), float("-inf"))
            lpf[:, -1] = lpf[:, -1] - 1000
            unaug = lpf.clone()
            for pg in range(g):
                for w in seqs[pg][lt].tolist():
                    lpf[:, w] = lpf[:, w] - diversity_lambda
            ys, ix = torch.sort(lpf, 1, True)
            cands = []
            rows = 1 if lt == 0 else B
            for c in range(min(B, ys.shape[1])):
                for q in range(rows):
                    cands.append(dict(c=int(ix[q, c]), q=q, p=float(sums[g][q]) + float(ys[q, c]), r=float(unaug[q, ix[q, c]])))
            cands = sorted(cands, key=lambda x: -x["p"])
            new_state = tuple(x.clone() for x in states[g])
            prev_seq, prev_lp = seqs[g][:lt].clone(), lps[g][:lt].clone()
            for vix in range(B):
                v = cands[vix]
                if lt >= 1:
                    seqs[g][:lt, vix] = prev_seq[:, v["q"]]
                    lps[g][:lt, vix] = prev_lp[:, v["q"]]
                for k in range(len(new_state)):
                    new_state[k][:, vix] = states[g][k][:, v["q"]]
                seqs[g][lt, vix] = v["c"]
                lps[g][lt, vix] = v["r"]
                sums[g][vix] = v["p"]
            states[g] = new_state
            for vix in range(B):
                if int(seqs[g][lt, vix]) == 0 or lt == L - 1:
                    p = float(sums[g][vix])
                    done[g].append(dict(seq=seqs[g][:, vix].clone(), logps=lps[g][:, vix].clone(),
                                        p=p / (lt + 1) if max_ppl else p, group=g))
                    sums[g][vix] = -1000
            lp_tab[g], states[g] = step_fn(seqs[g][lt].clone(), states[g])
    out = []
    for g in range(G):
        out += sorted(done[g], key=lambda x: -x["p"])[:B]
    return out


def sample_beam(W: Weights, fc_feats: Tensor, att_feats: Tensor, att_masks: Optional[Tensor], seq_length: int,
                beam_size: int, decoding_constraint: int = 0, max_ppl: int = 0, use_bn: int = 0, return_beams: bool = False,
                group_size: int = 1, diversity_lambda: float = 0.5):
    """AttModel._sample_beam (P/models/AttModel.py:167-196) over beam_search_core, image by image like the reference;
    the best finished beam per image is the result (:193-194).  group_size > 1: diverse_beam_search_core with
    beam_size // group_size beams per group (the replicated rows are chunked per group, CaptionModel.py:113-120)."""
    assert beam_size <= logit_final_weight(W).shape[0]
    N = fc_feats.shape[0]
    H = logit_final_weight(W).shape[1]
    fc, att, p_att, masks = prepare_feature(W, fc_feats, att_feats, att_masks, None, use_bn, False)
    B, L = beam_size // group_size, seq_length
    seq = torch.zeros(N, L, dtype=torch.long)
    seq_logp = torch.zeros(N, L)
    all_done = []
    for k in range(N):
        tfc = fc[k:k + 1].expand(B, -1)
        tatt = att[k:k + 1].expand(B, -1, -1).contiguous()
        tpatt = p_att[k:k + 1].expand(B, -1, -1).contiguous()
        tmask = masks[k:k + 1].expand(B, -1).contiguous() if masks is not None else None

        def step_fn(it, state):
            lp, st, _ = logprobs_step(W, it, tfc, tatt, tpatt, tmask, state)
            return lp, st
        state = (torch.zeros(2, B, H), torch.zeros(2, B, H))
        logprobs, state = step_fn(torch.zeros(B, dtype=torch.long), state)
        if group_size > 1:
            done = diverse_beam_search_core(step_fn, logprobs, state, L, beam_size, group_size, diversity_lambda,
                                            decoding_constraint, max_ppl)
        else:
            done = beam_search_core(step_fn, logprobs, state, L, B, decoding_constraint, max_ppl)
        seq[k] = done[0]["seq"]
        seq_logp[k] = done[0]["logps"]
        all_done.append(done)
    if return_beams:
        return seq, seq_logp, all_done
    return seq, seq_logp


def adam_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], m: Dict[str, Tensor],
              v: Dict[str, Tensor], step: int, lr: float, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam as built by Optim.create_optimizer, P/misc/optimizer.py:59-74
    (alpha, beta) = (0.9, 0.999), eps 1e-8, weight_decay 0.  ``step`` is 1-based.
    The captioner's gradient clipping is a no-op in the reference
    (generator consumed, P/misc/optimizer.py:78-79,92), so none is applied."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k, p in params.items():
        g = grads[k]
        m[k].mul_(beta1).add_(g, alpha=1 - beta1)
        v[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m[k], denom, value=-lr / bc1)


TRAINABLE_SKIP = ("running_mean", "running_var", "num_batches_tracked")


def xe_loss_and_grads(W: Weights, fc_feats, att_feats, labels, masks, att_masks=None,
                      drop=None, use_bn: int = 0, training: bool = True, ss=None):
    """One Trainer.train XE step up to backward(), P/trainer.py:164-165,172-173."""
    Wg = {k: (v.detach().clone().requires_grad_(True)
              if v.is_floating_point() and not k.endswith(TRAINABLE_SKIP) else v)
          for k, v in W.items()}
    logp = forward_logprobs(Wg, fc_feats, att_feats, labels, att_masks, drop, use_bn, training, ss=ss)
    loss = lm_criterion(logp, labels[:, 1:], masks[:, 1:])
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v))
             for k, v in Wg.items() if isinstance(v, Tensor) and v.requires_grad}
    return loss.detach(), grads, logp.detach()


def init_weights(V1: int, E: int, H: int, A: int, D: int, Dfc: int, seed: int = 0,
                 use_bn: int = 0) -> Weights:
    """Random weights with torch's default initialisers' scale (shape contract of
    P/models/AttModel.py:56-92; values are NOT the reference's RNG stream)."""
    g = torch.Generator().manual_seed(seed)

    def u(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    W: Weights = {}
    W["embed.0.weight"] = torch.randn(V1, E, generator=g)
    W["fc_embed.0.weight"] = u((H, Dfc), Dfc)
    W["fc_embed.0.bias"] = u((H,), Dfc)
    lk = _att_linear_key(use_bn)
    if use_bn:
        W["att_embed.0.weight"] = torch.ones(D)
        W["att_embed.0.bias"] = torch.zeros(D)
        W["att_embed.0.running_mean"] = torch.zeros(D)
        W["att_embed.0.running_var"] = torch.ones(D)
        W["att_embed.0.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    W[lk + ".weight"] = u((H, D), D)
    W[lk + ".bias"] = u((H,), D)
    if use_bn == 2:
        W["att_embed.4.weight"] = torch.ones(H)
        W["att_embed.4.bias"] = torch.zeros(H)
        W["att_embed.4.running_mean"] = torch.zeros(H)
        W["att_embed.4.running_var"] = torch.ones(H)
        W["att_embed.4.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    W["logit.weight"] = u((V1, H), H)
    W["logit.bias"] = u((V1,), H)
    W["ctx2att.weight"] = u((A, H), H)
    W["ctx2att.bias"] = u((A,), H)
    W["core.att_lstm.weight_ih"] = u((4 * H, E + 2 * H), H)
    W["core.att_lstm.weight_hh"] = u((4 * H, H), H)
    W["core.att_lstm.bias_ih"] = u((4 * H,), H)
    W["core.att_lstm.bias_hh"] = u((4 * H,), H)
    W["core.lang_lstm.weight_ih"] = u((4 * H, 2 * H), H)
    W["core.lang_lstm.weight_hh"] = u((4 * H, H), H)
    W["core.lang_lstm.bias_ih"] = u((4 * H,), H)
    W["core.lang_lstm.bias_hh"] = u((4 * H,), H)
    W["core.attention.h2att.weight"] = u((A, H), H)
    W["core.attention.h2att.bias"] = u((A,), H)
    W["core.attention.alpha_net.weight"] = u((1, A), A)
    W["core.attention.alpha_net.bias"] = u((1,), A)
    return W


def synthetic_batch(n_img: int, seq_per_img: int, R: int, D: int, V: int, L: int,
                    seed: int = 1234, ragged_regions: bool = False):
    """Synthetic batch with the layout of DataLoader.get_batch
    (P/misc/dataloader/dataloader.py:209-299), per BASELINE.md section 3."""
    g = torch.Generator().manual_seed(seed)
    att = torch.randn(n_img, R, D, generator=g).abs()
    att = att / att.norm(dim=2, keepdim=True)                          # norm_att_feat :310-311
    if ragged_regions:
        cnt = torch.randint(max(1, R // 4), R + 1, (n_img,), generator=g)
        cnt[0] = R
    else:
        cnt = torch.full((n_img,), R, dtype=torch.long)
    att_masks = (torch.arange(R)[None, :] < cnt[:, None]).float()
    att = att * att_masks.unsqueeze(2)
    fc = att.sum(1) / cnt[:, None].float()                             # make_bu_data.py:56
    N = n_img * seq_per_img
    rep = torch.arange(n_img).repeat_interleave(seq_per_img)           # dataloader.py:270-277
    lens = torch.randint(max(1, L // 2), L + 1, (N,), generator=g)
    toks = torch.randint(1, V + 1, (N, L), generator=g)
    labels = torch.zeros(N, L + 2, dtype=torch.long)
    pos = torch.arange(L)[None, :]
    labels[:, 1:L + 1] = toks * (pos < lens[:, None]).long()
    masks = (torch.arange(L + 2)[None, :] < (lens[:, None] + 2)).float()  # dataloader.py:283-286
    return dict(fc_feats=fc[rep].contiguous(), att_feats=att[rep].contiguous(),
                att_masks=att_masks[rep].contiguous(), labels=labels, masks=masks)
