"""TEST INFRASTRUCTURE -- CPU restatement of the self-critical reward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(unpaired_image_captioning_amd/misc/rewards.py over csrc/cider.hip) never does.

Restates, on integer token arrays instead of strings of token ids:
  * CIDEr-D as the reference's py2 scorer computes it (P/misc/cider/pyciderevalcap/ciderD/ciderD_scorer.py:13-209,
    P/misc/cider/pyciderevalcap/ciderD/ciderD.py:26-50) in python floats (IEEE double), same operation order;
  * get_self_critical_reward (P/misc/rewards.py:29-81) for bleu_reward_weight = 0.

PINNED: tests/test_oracle_ciderd.py checks it against tests/golden/ciderd_*.npz, produced by the reference's own
scorer classes (tests/golden/make_golden_cider.py runs them through the standard lib2to3 fixers in memory: the
files are python 2 -- xrange, dict.iteritems).

Kept exactly, because they shape the numbers:
  * a caption's words are its token ids up to AND INCLUDING the first 0 (array_to_str, rewards.py:29-35);
  * "length" is the number of BIGRAMS (counts2vec adds term_freq when n == 1 with n = len(ngram) - 1, :134-135);
  * with a cached document-frequency file, ref_len is used as stored in the pickle (:66; scripts/prepro_ngrams.py:125
    stores the raw image COUNT, not its log), with df_mode "corpus" it is log(number of hypotheses) (:177);
  * tf-idf weight = tf * (ref_len - log(max(1, df))) (:128-132); similarity = sum over the hypothesis' n-grams of
    min(w_hyp, w_ref) * w_ref / (|hyp| |ref|), times exp(-(len_hyp - len_ref)^2 / (2 sigma^2)) (:141-164);
  * score = mean over n = 1..4, mean over references, times 10 (:181-197).
"""
import math
from collections import defaultdict

import numpy as np


def caption_words(row):
    """array_to_str (P/misc/rewards.py:29-35) on integers: tokens up to and including the first 0."""
    out = []
    for t in row:
        out.append(int(t))
        if int(t) == 0:
            break
    return out


def precook(words, n=4):
    """ciderD_scorer.py:13-28: n-gram -> term frequency, insertion order = (order k, first position)."""
    counts = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return counts


def corpus_document_frequency(ref_sets):
    """compute_doc_freq (ciderD_scorer.py:103-114): one 'document' per entry of ref_sets (= per hypothesis)."""
    df = defaultdict(float)
    for refs in ref_sets:
        for ngram in set(g for ref in refs for g in precook(ref)):
            df[ngram] += 1
    return df


def counts2vec(cnts, df, ref_len, n=4):
    vec = [dict() for _ in range(n)]
    length = 0
    norm = [0.0 for _ in range(n)]
    for ngram, tf in cnts.items():
        logdf = float(np.log(max(1.0, df.get(ngram, 0.0))))
        k = len(ngram) - 1
        vec[k][ngram] = float(tf) * (ref_len - logdf)
        norm[k] += pow(vec[k][ngram], 2)
        if k == 1:
            length += tf
    norm = [float(np.sqrt(x)) for x in norm]
    return vec, norm, length


def sim(vec_hyp, vec_ref, norm_hyp, norm_ref, length_hyp, length_ref, n=4, sigma=6.0):
    delta = float(length_hyp - length_ref)
    val = [0.0 for _ in range(n)]
    for k in range(n):
        for ngram in vec_hyp[k]:
            r = vec_ref[k].get(ngram, 0.0)
            val[k] += min(vec_hyp[k][ngram], r) * r
        if norm_hyp[k] != 0 and norm_ref[k] != 0:
            val[k] /= (norm_hyp[k] * norm_ref[k])
        assert not math.isnan(val[k])
        val[k] *= float(np.e ** (-(delta ** 2) / (2 * sigma ** 2)))
    return val


def ciderd_scores(hyps, ref_sets, df=None, ref_len=None, n=4, sigma=6.0):
    """CiderD.compute_score (ciderD.py:26-50 -> ciderD_scorer.py:198-209).  hyps: list of word lists; ref_sets: per
    hypothesis, a list of reference word lists.  df None -> df_mode 'corpus'.  Returns (mean, per-hypothesis array)."""
    if df is None:
        df = corpus_document_frequency(ref_sets)
        ref_len = float(np.log(float(len(ref_sets))))
    scores = []
    for hyp, refs in zip(hyps, ref_sets):
        vec, norm, length = counts2vec(precook(hyp, n), df, ref_len, n)
        score = np.array([0.0 for _ in range(n)])
        for ref in refs:
            vec_ref, norm_ref, length_ref = counts2vec(precook(ref, n), df, ref_len, n)
            score += np.array(sim(vec, vec_ref, norm, norm_ref, length, length_ref, n, sigma))
        score_avg = np.mean(score)
        score_avg /= len(refs)
        score_avg *= 10.0
        scores.append(score_avg)
    return float(np.mean(np.array(scores))), np.array(scores)


def self_critical_reward(gen_result, greedy_res, gts, df=None, ref_len=None, cider_reward_weight=1.0):
    """get_self_critical_reward (P/misc/rewards.py:37-81) after the greedy decode, bleu_reward_weight = 0.
    gen_result / greedy_res: int arrays [N, L]; gts: list (one per image) of int arrays [n_caps, L]."""
    gen_result = np.asarray(gen_result)
    greedy_res = np.asarray(greedy_res)
    batch_size = gen_result.shape[0]
    seq_per_img = batch_size // len(gts)
    hyps = [caption_words(gen_result[i]) for i in range(batch_size)] + [caption_words(greedy_res[i]) for i in range(batch_size)]
    refs = [[caption_words(r) for r in gts[i]] for i in range(len(gts))]
    ref_sets = [refs[i % batch_size // seq_per_img] for i in range(2 * batch_size)]
    _, scores = ciderd_scores(hyps, ref_sets, df, ref_len)
    scores = cider_reward_weight * scores
    scores = scores[:batch_size] - scores[batch_size:]
    return np.repeat(scores[:, np.newaxis], gen_result.shape[1], 1)
