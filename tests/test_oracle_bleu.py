"""oracle/bleu.py against the reference's own BLEU scorer (tests/golden/bleu_*.npz, tests/golden/make_golden_bleu.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import bleu as OB
from oracle import ciderd as OC

CASES = ["bleu_tiny", "bleu_real_shape"]


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    gts = [z["gts_tok"][z["gts_start"][i]:z["gts_start"][i + 1]] for i in range(len(z["gts_start"]) - 1)]
    return z, gts, int(z["seq_per_img"])


@pytest.mark.parametrize("name", CASES)
def test_sentence_bleu_is_bit_identical(name):
    z, gts, S = load(name)
    hyp = np.concatenate([z["gen"], z["greedy"]], 0)
    N = len(z["gen"])
    for h in range(2 * N):
        got = OB.sentence_bleu(hyp[h], gts[h % N // S])
        assert got == list(z["bleu"][:, h]), h
    assert OB.bleu4_scores(hyp, gts, N, S) == list(z["bleu"][3])


@pytest.mark.parametrize("name", CASES)
def test_mixed_reward_matches_get_self_critical_reward(name):
    z, gts, S = load(name)
    hyp = np.concatenate([z["gen"], z["greedy"]], 0)
    N, L = z["gen"].shape
    bleu = np.array(OB.bleu4_scores(hyp, gts, N, S))
    refs = [[OC.caption_words(r) for r in g] for g in gts]
    _, cider = OC.ciderd_scores([OC.caption_words(r) for r in hyp], [refs[h % N // S] for h in range(2 * N)])   # df 'corpus'
    for tag in ("mix", "bleu_only", "bleu_heavy"):
        cw, bw = z["weights_" + tag]
        scores = (cw * cider if cw > 0 else 0) + bw * bleu
        want = z["reward_" + tag]
        got = np.repeat((scores[:N] - scores[N:])[:, None], L, 1)
        assert np.array_equal(got, want), tag
