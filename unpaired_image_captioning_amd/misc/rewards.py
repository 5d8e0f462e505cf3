"""Self-critical reward (P/misc/rewards.py) on the MI355X: CIDEr-D of the sampled captions minus CIDEr-D of the greedy
baseline, scored ON THE DEVICE (csrc/cider.hip) so that neither caption set is copied to the host and no python
n-gram loop sits between the sampling pass and the policy-gradient step.

Same surface as the reference module: `init_scorer(cached_tokens)`, `array_to_str`, `get_self_critical_reward(model,
fc_feats, attri_feats, att_feats, att_masks, data, gen_result, opt)` -> float array [N, L].  The Trainer uses
`self_critical_reward_device`, which returns the device tensor without synchronising.

bleu_reward_weight > 0 (default 0, P/opts.py:154) adds the per-sentence BLEU-4 of the reference's Bleu(4) scorer
(rewards.py:70-75), also scored on the device (`uic_bleu_scores`).
"""
import ctypes as C
import os
import pickle

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream

CiderD_scorer = None
Bleu_scorer = None

SIGMA = 6.0     # CiderD(sigma=6.0), ciderD.py:19


class DeviceCiderD(object):
    """CiderD(df=...) (P/misc/cider/pyciderevalcap/ciderD/ciderD.py:19-50).  `document_frequency`: dict n-gram tuple
    (ints, or the strings of token ids the cached pickle holds) -> count, with `ref_len` as stored beside it -- or None
    for df_mode 'corpus', where both are recomputed from every batch's references (ciderD_scorer.py:177,199-203)."""

    def __init__(self, document_frequency=None, ref_len=None, device="cuda"):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.corpus = document_frequency is None
        self._pen = {}
        if not self.corpus:
            self.ref_len = float(ref_len)
            self.slot_keys, self.slot_vals, self.slots = self._table(document_frequency)

    def _table(self, df):
        items = [(k, v) for k, v in df.items() if 1 <= len(k) <= 4]      # the scorer only ever asks for 1..4-grams
        n = len(items)
        keys = np.full((max(n, 1), 4), -1, dtype=np.int32)
        vals = np.zeros(max(n, 1), dtype=np.float64)
        for j, (k, v) in enumerate(items):
            keys[j, :len(k)] = [int(t) for t in k]
            vals[j] = np.log(max(1.0, float(v)))                       # counts2vec, ciderD_scorer.py:128
        slots = int(self.lib.uic_ciderd_table_slots(n))
        sk = np.empty((slots, 4), dtype=np.int32)
        sv = np.empty(slots, dtype=np.float64)
        check(self.lib.uic_ciderd_table_build(keys.ctypes.data, vals.ctypes.data, n, sk.ctypes.data, sv.ctypes.data, slots),
              "ciderd_table_build")
        return torch.from_numpy(sk).to(self.device), torch.from_numpy(sv).to(self.device), slots

    def _penalty(self, half):
        if half not in self._pen:
            t = np.array([np.e ** (-(float(d) ** 2) / (2 * SIGMA ** 2)) for d in range(-half, half + 1)], dtype=np.float64)
            self._pen[half] = torch.from_numpy(t).to(self.device)          # :164, evaluated as the scorer evaluates it
        return self._pen[half]

    @staticmethod
    def _words(row):
        out = []
        for t in row:
            out.append(int(t))
            if int(t) == 0:
                break
        return out

    def _corpus_table(self, gts, n_hyp, batch_size, seq_per_img):
        """df_mode 'corpus' (:103-114,177): one document per hypothesis = the reference set of its image."""
        per_img = []
        for refs in gts:
            grams = set()
            for r in refs:
                w = self._words(r)
                for k in range(1, 5):
                    for i in range(len(w) - k + 1):
                        grams.add(tuple(w[i:i + k]))
            per_img.append(grams)
        df = {}
        for h in range(n_hyp):
            for g in per_img[h % batch_size // seq_per_img]:
                df[g] = df.get(g, 0.0) + 1.0
        return self._table(df) + (float(np.log(float(n_hyp))),)

    def scores(self, hyp, gts, batch_size, seq_per_img, refs=None):
        """hyp: device int64 [n_hyp, L]; gts: list (one per image) of integer arrays [n_caps, Lr] (data['gts']).
        Returns device f64 [n_hyp] -- enqueue only."""
        hyp = hyp.contiguous()
        n_hyp, L = hyp.shape
        n_img = len(gts)
        ref_d, start_d, Lr = refs if refs is not None else upload_references(gts, self.device)
        if self.corpus:
            sk, sv, slots, ref_len = self._corpus_table(gts, n_hyp, batch_size, seq_per_img)
        else:
            sk, sv, slots, ref_len = self.slot_keys, self.slot_vals, self.slots, self.ref_len
        half = max(L, Lr, 32)
        pen = self._penalty(half)
        out = torch.empty(n_hyp, dtype=torch.float64, device=self.device)
        check(self.lib.uic_ciderd_scores(ptr(hyp), n_hyp, L, batch_size, seq_per_img, ptr(ref_d), Lr, ptr(start_d), n_img,
                                         ptr(sk), ptr(sv), slots, C.c_double(ref_len), ptr(pen), half, ptr(out), stream()),
              "ciderd_scores")
        return out


def upload_references(gts, device):
    """data['gts'] (one integer array [n_caps, Lr] per image) -> (tokens [sum n_caps, Lr] i64, first row per image [n_img + 1]
    i32, Lr) on the device -- shared by the CIDEr-D and the BLEU kernel."""
    ref = np.concatenate([np.asarray(g).reshape(len(g), -1) for g in gts], 0).astype(np.int64)
    start = np.cumsum([0] + [len(g) for g in gts]).astype(np.int32)
    ref_d = torch.from_numpy(np.ascontiguousarray(ref)).to(device, non_blocking=True)
    start_d = torch.from_numpy(start).to(device, non_blocking=True)
    return ref_d, start_d, ref.shape[1]


def bleu4_scores_device(hyp, gts, batch_size, seq_per_img, refs=None):
    """Bleu(4).compute_score(gts, res)[1][3] (rewards.py:71-72) for hyp [n_hyp, L] i64 on the device -> f64 [n_hyp], enqueue only."""
    lib = _lib.load()
    hyp = hyp.contiguous()
    n_hyp, L = hyp.shape
    ref_d, start_d, Lr = refs if refs is not None else upload_references(gts, hyp.device)
    out = torch.empty(n_hyp, dtype=torch.float64, device=hyp.device)
    check(lib.uic_bleu_scores(ptr(hyp), n_hyp, L, batch_size, seq_per_img, ptr(ref_d), Lr, ptr(start_d), len(gts), ptr(out),
                              stream()), "bleu_scores")
    return out


def init_scorer(cached_tokens, device="cuda"):
    """rewards.init_scorer (P/misc/rewards.py:24-28): CiderD(df=cached_tokens); 'corpus' computes the document
    frequencies from every batch.  The pickle (scripts/prepro_ngrams.py:125-126) is read from data/<cached_tokens>.p
    like the reference does (ciderD_scorer.py:64-66), or from the path itself when it exists."""
    global CiderD_scorer
    if CiderD_scorer is not None:
        return CiderD_scorer
    if cached_tokens == "corpus":
        CiderD_scorer = DeviceCiderD(None, None, device)
        return CiderD_scorer
    path = cached_tokens if os.path.exists(cached_tokens) else os.path.join('data', cached_tokens + '.p')
    with open(path, 'rb') as f:
        pkl = pickle.load(f, encoding='latin1')             # written by python 2 in the reference's workflow
    CiderD_scorer = DeviceCiderD(pkl['document_frequency'], pkl['ref_len'], device)
    return CiderD_scorer


def array_to_str(arr):
    """P/misc/rewards.py:29-35."""
    out = ''
    for i in range(len(arr)):
        out += str(int(arr[i])) + ' '
        if arr[i] == 0:
            break
    return out.strip()


def self_critical_reward_device(scorer, gen_result, greedy_res, gts, cider_reward_weight=1.0, bleu_reward_weight=0.0):
    """reward [N, L] f32 on the device, no host synchronisation: score(sampled) - score(greedy) per caption row, repeated
    over the L positions, score = cider_reward_weight * CIDEr-D + bleu_reward_weight * BLEU-4 (P/misc/rewards.py:49-79).
    greedy_res may hold one row per image (the eval-mode decode of identical replicas is identical): it is expanded to
    the N caption rows."""
    N, L = gen_result.shape
    seq_per_img = N // len(gts)
    refs = upload_references(gts, gen_result.device)
    lib = _lib.load()
    reward = torch.empty(N, L, dtype=torch.float32, device=gen_result.device)
    n_g = greedy_res.shape[0]
    if n_g != N and bleu_reward_weight <= 0 and not getattr(scorer, 'corpus', False) and n_g % len(gts) == 0:
        # one greedy row per image and a document-frequency table that does not depend on the hypothesis count: every distinct
        # greedy caption is scored ONCE and its score repeated (the same f64 operations per hypothesis: identical scores)
        s = torch.cat([scorer.scores(gen_result, gts, N, seq_per_img, refs),
                       scorer.scores(greedy_res, gts, n_g, n_g // len(gts), refs).repeat_interleave(N // n_g)])
        check(lib.uic_ciderd_reward(ptr(s), N, L, float(cider_reward_weight), ptr(reward), stream()), "ciderd_reward")
        return reward
    if n_g != N:
        greedy_res = greedy_res.repeat_interleave(N // n_g, 0)
    hyp = torch.cat([gen_result, greedy_res], 0)
    if bleu_reward_weight > 0:
        s = float(bleu_reward_weight) * bleu4_scores_device(hyp, gts, N, seq_per_img, refs)
        if cider_reward_weight > 0:                   # :76, in that order, f64
            s = float(cider_reward_weight) * scorer.scores(hyp, gts, N, seq_per_img, refs) + s
        weight = 1.0
    else:
        s = scorer.scores(hyp, gts, N, seq_per_img, refs)
        weight = float(cider_reward_weight)
    check(lib.uic_ciderd_reward(ptr(s), N, L, weight, ptr(reward), stream()), "ciderd_reward")
    return reward


def get_self_critical_reward(model, fc_feats, attri_feats, att_feats, att_masks, data, gen_result, opt):
    """P/misc/rewards.py:37-81: greedy baseline in eval mode, then the reward as a float array [N, L]."""
    if CiderD_scorer is None:
        raise RuntimeError("call init_scorer(opt.cached_tokens) first (P/trainer.py:158)")
    model.eval()
    with torch.no_grad():
        greedy_res, _ = model(fc_feats, attri_feats, att_feats, att_masks=att_masks, mode='sample')
    model.train()
    w = float(getattr(opt, 'cider_reward_weight', 1))
    bw = float(getattr(opt, 'bleu_reward_weight', 0))
    if w <= 0 and bw <= 0:
        return np.zeros(tuple(gen_result.shape), dtype=np.float64)
    r = self_critical_reward_device(CiderD_scorer, gen_result.detach(), greedy_res, data['gts'], w, bw)
    return r.double().cpu().numpy()
