"""Pin oracle/ciderd.py against golden vectors produced by the reference's CIDEr-D scorer and get_self_critical_reward
(tests/golden/make_golden_cider.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import ciderd as OC

CASES = ["ciderd_cached", "ciderd_corpus", "ciderd_logreflen", "ciderd_real_shape"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    gts = [z["gts_tok"][z["gts_start"][i]:z["gts_start"][i + 1]] for i in range(len(z["gts_start"]) - 1)]
    df = ref_len = None
    if "df_keys" in z.files:
        df = {tuple(int(t) for t in k if t >= 0): float(c) for k, c in zip(z["df_keys"], z["df_count"])}
        ref_len = float(z["ref_len"])
    return z, gts, df, ref_len


@pytest.mark.parametrize("name", CASES)
def test_ciderd_scores_and_reward_match_reference(name):
    z, gts, df, ref_len = load_case(name)
    N, S = z["gen"].shape[0], int(z["seq_per_img"])
    hyps = [OC.caption_words(r) for r in z["gen"]] + [OC.caption_words(r) for r in z["greedy"]]
    refs = [[OC.caption_words(r) for r in g] for g in gts]
    mean, scores = OC.ciderd_scores(hyps, [refs[i % N // S] for i in range(2 * N)], df, ref_len)
    # same IEEE-double operations in the same order as the reference: bit-identical
    assert np.array_equal(scores, z["scores"]), np.abs(scores - z["scores"]).max()
    assert mean == float(z["mean"])
    reward = OC.self_critical_reward(z["gen"], z["greedy"], gts, df, ref_len)
    assert reward.shape == z["reward"].shape and np.array_equal(reward, z["reward"])
    assert (z["scores"] > 0).any() and np.abs(z["reward"]).max() > 0.1


def test_fixture_covers_the_edge_cases():
    z, gts, df, ref_len = load_case("ciderd_cached")
    assert (z["gen"][1] == 0).all()                                  # a hypothesis that ends at once: the single word "0"
    assert any((r != 0).all() for r in z["gen"])                     # one that fills all L positions (no terminating 0)
    w = OC.caption_words(z["greedy"][2])
    assert len(set(w[:-1])) == 1 and len(w) > 3                      # a repeated word: term frequencies > 1
    assert np.array_equal(z["greedy"][0], gts[0][-1])                # a hypothesis equal to one of its references
    hyp_ngrams = set(OC.precook(OC.caption_words(z["gen"][0])))
    assert any(g not in df for g in hyp_ngrams)                      # n-grams missing from the cached table (df -> log 1 = 0)
    assert ref_len == 52.0                                           # the raw image count, as prepro_ngrams.py stores it
