"""CPU restatement of the reference's per-sentence BLEU (the `bleu_reward_weight` half of the self-critical reward) --
TEST INFRASTRUCTURE ONLY.  Follows P/AI_Challenger/Evaluation/caption_eval/coco_caption/pycxevalcap/bleu/bleu_scorer.py
(precook :23-34, cook_refs :36-59, cook_test :61-88, BleuScorer.compute_score :199-262 with option='closest' as
bleu.py:41 calls it) on integer token rows; python floats (f64) in the reference's order.

Pinned: tests/golden/bleu_*.npz hold what the reference's scorer and its get_self_critical_reward returned
(tests/golden/make_golden_bleu.py); tests/test_oracle_bleu.py compares bit for bit.
"""
import math


def words(row):
    """array_to_str(...).split() (P/misc/rewards.py:29-35): tokens up to and including the first 0."""
    out = []
    for t in row:
        out.append(int(t))
        if int(t) == 0:
            break
    return out


def _counts(w, n=4):
    c = {}
    for k in range(1, n + 1):
        for i in range(len(w) - k + 1):
            g = tuple(w[i:i + k])
            c[g] = c.get(g, 0) + 1
    return c


def sentence_bleu(hyp_row, ref_rows, n=4):
    """[BLEU-1 .. BLEU-n] of one hypothesis against its references (bleu_list[k][i] of compute_score)."""
    small, tiny = 1e-9, 1e-15
    test = words(hyp_row)
    testlen = len(test)
    reflens, maxcounts = [], {}
    for r in ref_rows:
        w = words(r)
        reflens.append(len(w))
        for g, c in _counts(w, n).items():
            maxcounts[g] = max(maxcounts.get(g, 0), c)
    reflen = min((abs(l - testlen), l) for l in reflens)[1]                # 'closest' (:76,:187)
    guess = [max(0, testlen - k + 1) for k in range(1, n + 1)]
    correct = [0] * n
    for g, c in _counts(test, n).items():
        correct[len(g) - 1] += min(maxcounts.get(g, 0), c)
    out = []
    bleu = 1.
    for k in range(n):
        bleu *= (float(correct[k]) + tiny) / (float(guess[k]) + small)
        out.append(bleu ** (1. / (k + 1)))
    ratio = (testlen + tiny) / (reflen + small)
    if ratio < 1:
        for k in range(n):
            out[k] *= math.exp(1 - 1 / ratio)
    return out


def bleu4_scores(hyp, gts, batch_size, seq_per_img):
    """BLEU-4 of every row of hyp [n_hyp, L] against the references of image (h % batch_size) // seq_per_img (rewards.py:59)."""
    return [sentence_bleu(hyp[h], gts[h % batch_size // seq_per_img])[3] for h in range(len(hyp))]
