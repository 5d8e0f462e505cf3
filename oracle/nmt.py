"""Plain-torch fp32 CPU restatement of the pivot NMT path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Restates, with explicit loops instead of cuDNN-style fused
calls: ``Embeddings`` / ``Encoder`` (packed bi-LSTM) / ``Decoder`` (input-feed StackedLSTM + dot GlobalAttention) /
``NMTModel.forward`` (``P/models/NMT_Models.py:27-72, 75-135, 137-271, 284-295, 414-420``),
``StackedLSTM`` (``O/modules/StackedRNN.py:20-34``), ``GlobalAttention.forward`` dotprod + softmax
(``O/modules/GlobalAttention.py:112-167``) and ``NMTCriterion`` / the generator (``P/misc/criterion.py:126-136,181-184``).
Pinned by ``tests/golden/nmt_*.npz`` (generated from those reference modules).

state_dict keys as in the reference: encoder.embeddings.{word_lut.weight, linear.weight, linear.bias},
encoder.rnn.{weight,bias}_{ih,hh}_l{k}[_reverse], decoder.embeddings.word_lut.weight,
decoder.rnn.layers.{k}.{weight,bias}_{ih,hh}, decoder.attn.linear_{in,out}.weight, generator.0.{weight,bias}.

Dropout (opt.dropout): ``drop`` is None or {'enc': [layers-1][S,B,H] (between encoder layers),
'dec': [layers-1][T-1,B,H] (between decoder layers), 'out': [T-1,B,H] (decoder output = next input feed)}.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .topdown import lstm_cell

Tensor = torch.Tensor
Weights = Dict[str, Tensor]
PAD = 0


def n_layers(W: Weights) -> int:
    return sum(1 for k in W if k.startswith("decoder.rnn.layers.") and k.endswith(".weight_ih"))


def encoder(W: Weights, src: Tensor, lengths: Tensor, drop=None):
    """Encoder.forward with pack/unpack semantics, P/models/NMT_Models.py:104-135.  src [S,B,1], lengths [1,B] sorted
    descending.  Returns (context [S,B,H], h [layers,B,H], c [layers,B,H]) with _fix_enc_hidden applied (:284-287)."""
    S, B = src.shape[0], src.shape[1]
    lens = lengths.view(-1).tolist()
    emb = W["encoder.embeddings.word_lut.weight"][src[:, :, 0]]
    x = torch.relu(F.linear(emb, W["encoder.embeddings.linear.weight"], W["encoder.embeddings.linear.bias"]))  # :65-67
    L = n_layers(W)
    hs, cs = [], []
    for l in range(L):
        outs = []
        for sfx, order in (("", range(S)), ("_reverse", range(S - 1, -1, -1))):
            w_ih, w_hh = W["encoder.rnn.weight_ih_l%d%s" % (l, sfx)], W["encoder.rnn.weight_hh_l%d%s" % (l, sfx)]
            b_ih, b_hh = W["encoder.rnn.bias_ih_l%d%s" % (l, sfx)], W["encoder.rnn.bias_hh_l%d%s" % (l, sfx)]
            Hd = w_hh.shape[1]
            h, c = torch.zeros(B, Hd), torch.zeros(B, Hd)
            out = [None] * S
            for s in order:
                nb = sum(1 for n in lens if n > s)              # active rows form a prefix (sorted lengths)
                h2, c2 = lstm_cell(x[s, :nb], h[:nb], c[:nb], w_ih, w_hh, b_ih, b_hh)
                h = torch.cat([h2, h[nb:]], 0)
                c = torch.cat([c2, c[nb:]], 0)
                out[s] = torch.cat([h2, torch.zeros(B - nb, Hd)], 0)    # unpack pads with zeros
            outs.append(torch.stack(out))
            hs.append(h)
            cs.append(c)
        x = torch.cat(outs, 2)
        if l + 1 < L and drop is not None:
            x = x * drop["enc"][l]
    context = x
    # hidden_t is [layers*2, B, Hd] ordered (l0 fwd, l0 bwd, l1 fwd, ...); _fix_enc_hidden concatenates fwd|bwd
    h = torch.stack([torch.cat([hs[2 * l], hs[2 * l + 1]], 1) for l in range(L)])
    c = torch.stack([torch.cat([cs[2 * l], cs[2 * l + 1]], 1) for l in range(L)])
    return context, h, c


def global_attention(W: Weights, q_in: Tensor, context_bsh: Tensor):
    """GlobalAttention.forward (dotprod, softmax, no mask), O/modules/GlobalAttention.py:112-167."""
    target = F.linear(q_in, W["decoder.attn.linear_in.weight"])
    scores = torch.bmm(context_bsh, target.unsqueeze(2)).squeeze(2)
    attn = F.softmax(scores, dim=1)
    c = torch.bmm(attn.unsqueeze(1), context_bsh).squeeze(1)
    out = torch.tanh(F.linear(torch.cat([c, q_in], 1), W["decoder.attn.linear_out.weight"]))
    return out, attn


def decoder(W: Weights, tgt_in: Tensor, context: Tensor, h0: Tensor, c0: Tensor, drop=None):
    """Decoder.forward, P/models/NMT_Models.py:183-271 (input feed, no context gate / coverage / copy)."""
    Tm1, B = tgt_in.shape
    H = context.shape[2]
    L = h0.shape[0]
    emb = W["decoder.embeddings.word_lut.weight"][tgt_in]                    # bare lut (feature_dicts=None, :160)
    ctx_b = context.transpose(0, 1)
    feed = torch.zeros(B, H)                                                  # init_input_feed :454-458
    h = [h0[l] for l in range(L)]
    c = [c0[l] for l in range(L)]
    outs, attns = [], []
    for t in range(Tm1):
        x = torch.cat([emb[t], feed], 1)                                      # :248-249
        for l in range(L):                                                    # StackedLSTM.forward
            h[l], c[l] = lstm_cell(x, h[l], c[l], W["decoder.rnn.layers.%d.weight_ih" % l], W["decoder.rnn.layers.%d.weight_hh" % l],
                                   W["decoder.rnn.layers.%d.bias_ih" % l], W["decoder.rnn.layers.%d.bias_hh" % l])
            x = h[l]
            if l + 1 < L and drop is not None:
                x = x * drop["dec"][l][t]
        out, attn = global_attention(W, x, ctx_b)
        if drop is not None:
            out = out * drop["out"][t]                                        # self.dropout(attn_output) :258
        feed = out
        outs.append(out)
        attns.append(attn)
    return torch.stack(outs), torch.stack(attns)


def forward(W: Weights, src: Tensor, tgt: Tensor, lengths: Tensor, drop=None):
    """NMTModel.forward, P/models/NMT_Models.py:414-420: returns (outputs [T-1,B,H], attn [T-1,B,S], context)."""
    context, h, c = encoder(W, src, lengths, drop)
    outputs, attn = decoder(W, tgt[:-1], context, h, c, drop)
    return outputs, attn, context, h, c


def nmt_loss(W: Weights, outputs: Tensor, tgt: Tensor):
    """generator + NMTCriterion (sum of NLL, PAD weight 0), P/misc/criterion.py:126-136,181-184; plus the
    accuracy counters of NMT_loss.score (:175-179)."""
    scores = F.log_softmax(F.linear(outputs.reshape(-1, outputs.shape[2]), W["generator.0.weight"], W["generator.0.bias"]), dim=1)
    target = tgt[1:].reshape(-1)
    nonpad = target.ne(PAD)
    loss = -(scores.gather(1, target.unsqueeze(1)).squeeze(1) * nonpad.float()).sum()
    num_correct = int((scores.max(1)[1].eq(target) & nonpad).sum())
    return loss, scores, num_correct, int(nonpad.sum())


def loss_and_grads(W: Weights, src, tgt, lengths, drop=None):
    Wg = {k: v.detach().clone().requires_grad_(True) for k, v in W.items()}
    outputs, attn, context, h, c = forward(Wg, src, tgt, lengths, drop)
    loss, scores, nc, nw = nmt_loss(Wg, outputs, tgt)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Wg.items()}
    return loss.detach(), grads, dict(outputs=outputs.detach(), attn=attn.detach(), context=context.detach(),
                                      h=h.detach(), c=c.detach(), scores=scores.detach(), num_correct=nc, num_words=nw)


def translate_batch(W: Weights, src: Tensor, beam_size: int = 15, max_steps: int = 100):
    """NMTModel.translateBatch (P/models/NMT_Models.py:322-395) with the fork's Beam (O/Beam.py:21-117), n_best = 1.

    Kept exactly: the encoder runs WITHOUT lengths (:327: PAD positions go through both LSTM directions and the final
    states are those of the last / first array position), source padding is masked in the attention only (:345,352);
    every beam starts as [BOS, PAD, ...] and the first step expands slot 0 only (Beam.py:37-38,66-69); a step takes
    the top `beam_size` of the flattened beam x word scores (:73); a sentence is done when its TOP hypothesis ends in
    EOS (:85-87) but its beam KEEPS ADVANCING until every sentence of the batch is done (translateBatch calls advance
    for all beams in every iteration, :372-376), and the hypothesis is read out after the last iteration by walking the
    back-pointers from the best final score (:384-392, Beam.py:93-117) -- so its length is the number of iterations.

    src [S, B, 1] int64.  Returns (hyp [B, n_iter] int64, scores [B], attn [B, n_iter, S] with PAD columns zeroed and
    packed to the left like the reference's index_select over the valid positions)."""
    BOS, EOS = 2, 3
    S, B = src.shape[0], src.shape[1]
    K = beam_size
    context, h0, c0 = encoder(W, src, torch.full((1, B), S, dtype=torch.long))     # lengths=None: every position is "valid"
    L = h0.shape[0]
    H = context.shape[2]
    # beam-major replication like `.repeat(1, beamSize, 1)`: row k * B + b
    ctx_b = context.repeat(1, K, 1).transpose(0, 1)                                  # [K*B, S, H]
    h = [h0[l].repeat(K, 1) for l in range(L)]
    c = [c0[l].repeat(K, 1) for l in range(L)]
    feed = torch.zeros(K * B, H)
    pad_mask = src[:, :, 0].eq(PAD).t().repeat(K, 1)                                 # [K*B, S]
    scores = [torch.zeros(K) for _ in range(B)]
    next_ys = [[torch.full((K,), PAD, dtype=torch.long)] for _ in range(B)]
    for b in range(B):
        next_ys[b][0][0] = BOS
    prev_ks = [[] for _ in range(B)]
    attns = [[] for _ in range(B)]
    done = [False] * B
    for _ in range(max_steps):
        inp = torch.stack([next_ys[b][-1] for b in range(B)]).t().contiguous().view(-1)     # [K*B], row k*B+b
        x = torch.cat([W["decoder.embeddings.word_lut.weight"][inp], feed], 1)
        for l in range(L):
            h[l], c[l] = lstm_cell(x, h[l], c[l], W["decoder.rnn.layers.%d.weight_ih" % l], W["decoder.rnn.layers.%d.weight_hh" % l],
                                   W["decoder.rnn.layers.%d.bias_ih" % l], W["decoder.rnn.layers.%d.bias_hh" % l])
            x = h[l]
        target = F.linear(x, W["decoder.attn.linear_in.weight"])
        sc = torch.bmm(ctx_b, target.unsqueeze(2)).squeeze(2).masked_fill(pad_mask, float("-inf"))
        attn = F.softmax(sc, dim=1)
        cvec = torch.bmm(attn.unsqueeze(1), ctx_b).squeeze(1)
        feed = torch.tanh(F.linear(torch.cat([cvec, x], 1), W["decoder.attn.linear_out.weight"]))
        out = F.log_softmax(F.linear(feed, W["generator.0.weight"], W["generator.0.bias"]), dim=1)
        word = out.view(K, B, -1).transpose(0, 1)                                    # [B, K, V]
        att = attn.view(K, B, -1).transpose(0, 1)                                    # [B, K, S]
        active = 0
        for b in range(B):
            V = word.shape[2]
            lk = word[b] + scores[b].unsqueeze(1) if prev_ks[b] else word[b][0]
            best, ids = lk.reshape(-1).topk(K, 0, True, True)
            scores[b] = best
            pk = torch.div(ids, V, rounding_mode="floor")
            prev_ks[b].append(pk)
            next_ys[b].append(ids - pk * V)
            attns[b].append(att[b].index_select(0, pk))
            if int(next_ys[b][-1][0]) == EOS:
                done[b] = True
            if not done[b]:
                active += 1
            # beamUpdate_: re-thread this sentence's rows of every state tensor to the surviving parents
            rows = torch.arange(K) * B + b
            for t_ in h + c + [feed]:
                t_[rows] = t_[rows].index_select(0, pk)
        if not active:
            break
    n_iter = len(prev_ks[0])
    hyp = torch.zeros(B, n_iter, dtype=torch.long)
    out_scores = torch.zeros(B)
    out_attn = torch.zeros(B, n_iter, S)
    for b in range(B):
        sc_sorted, ks = torch.sort(scores[b], 0, True)
        k = int(ks[0])
        out_scores[b] = sc_sorted[0]
        valid = src[:, b, 0].ne(PAD).nonzero().view(-1)
        for j in range(n_iter - 1, -1, -1):
            hyp[b, j] = next_ys[b][j + 1][k]
            out_attn[b, j, :valid.numel()] = attns[b][j][k].index_select(0, valid)
            k = int(prev_ks[b][j][k])
    return hyp, out_scores, out_attn
