#!/usr/bin/env python3
"""Golden vectors for the BLEU half of the self-critical reward (`bleu_reward_weight > 0`, P/misc/rewards.py:70-75) from the
REFERENCE's own scorer: AI_Challenger/Evaluation/caption_eval/coco_caption/pycxevalcap/bleu/{bleu,bleu_scorer}.py
(python 2: run through lib2to3 in memory like the CIDEr-D scorer, see make_golden_cider.py) and the reference's
get_self_critical_reward with both scorers live (CIDEr-D in 'corpus' mode).  Build container only.

    python tests/golden/make_golden_bleu.py        # writes tests/golden/bleu_*.npz
"""
import argparse
import contextlib
import io
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_cider as mc

B = os.path.join(mc.P, "AI_Challenger", "Evaluation", "caption_eval", "coco_caption", "pycxevalcap", "bleu")


def make_case(name, rewards, ciderd_mod, Bleu, seed, V, L, n_img, S):
    g = np.random.default_rng(seed)

    def caps(n):
        r = mc.random_captions(g, n, L, V)
        return np.where(r > 0, np.minimum(r, g.integers(1, V + 1, r.shape)), 0)
    gts = [caps(int(g.integers(1, 6))) for _ in range(n_img)]
    N = n_img * S
    gen, greedy = caps(N), caps(N)
    for i in range(0, N, 2):                       # hypotheses that share long n-grams with a reference
        ref = gts[i // S][0]
        gen[i, :L - 2] = ref[:L - 2]
    gen[1] = 0                                      # ends at once
    greedy[0] = gts[0][-1]                          # an exact copy of a reference
    greedy[2, :] = greedy[2, 0]                     # one word repeated (clipped counts)
    greedy[2, L // 2:] = 0

    class FakeModel(object):
        def eval(self): pass
        def train(self): pass
        def __call__(self, *a, **k):
            return torch.from_numpy(greedy), None
    rewards.CiderD_scorer = ciderd_mod.CiderD(df="corpus")
    rewards.Bleu_scorer = Bleu(4)
    z = torch.zeros(1)
    data = {"gts": gts}
    out = dict(gen=gen, greedy=greedy, gts_tok=np.concatenate(gts, 0),
               gts_start=np.cumsum([0] + [len(x) for x in gts]).astype(np.int32), seq_per_img=np.array(S))
    with contextlib.redirect_stdout(io.StringIO()):                        # the scorer prints its totals (verbose=1)
        for tag, cw, bw in (("mix", 1.0, 0.5), ("bleu_only", 0, 1.0), ("bleu_heavy", 0.25, 2.0)):
            opt = argparse.Namespace(cider_reward_weight=cw, bleu_reward_weight=bw)
            out["reward_" + tag] = np.asarray(rewards.get_self_critical_reward(FakeModel(), z, z, z, z, data, torch.from_numpy(gen), opt),
                                              dtype=np.float64)
            out["weights_" + tag] = np.array([cw, bw], dtype=np.float64)
        res = {i: [" ".join(mc.words(gen[i] if i < N else greedy[i - N]))] for i in range(2 * N)}
        gts_d = {i: [" ".join(mc.words(r)) for r in gts[i % N // S]] for i in range(2 * N)}
        corpus_bleu, per_sentence = Bleu(4).compute_score(gts_d, res)
    out["bleu"] = np.asarray(per_sentence, dtype=np.float64)               # [4, 2N]: BLEU-1 .. BLEU-4 of every hypothesis
    out["corpus_bleu"] = np.asarray(corpus_bleu, dtype=np.float64)
    assert np.array_equal(out["reward_bleu_only"][:, 0], out["bleu"][3][:N] - out["bleu"][3][N:])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "BLEU-4 range [%.4f, %.4f], corpus %s" % (out["bleu"][3].min(), out["bleu"][3].max(), np.round(corpus_bleu, 4)))


def main():
    scorer_mod, ciderd_mod, rewards = mc.load_reference()
    for name in ("coco_caption", "coco_caption.pycxevalcap", "coco_caption.pycxevalcap.bleu"):
        sys.modules[name].__path__ = []
    sys.modules["nltk"].__path__ = []
    mc.load_py2("coco_caption.pycxevalcap.bleu.bleu_scorer", os.path.join(B, "bleu_scorer.py"))
    bleu_mod = mc.load_py2("coco_caption.pycxevalcap.bleu.bleu", os.path.join(B, "bleu.py"))
    make_case("bleu_tiny", rewards, ciderd_mod, bleu_mod.Bleu, 21, V=10, L=8, n_img=4, S=3)
    make_case("bleu_real_shape", rewards, ciderd_mod, bleu_mod.Bleu, 22, V=60, L=16, n_img=6, S=5)


if __name__ == "__main__":
    main()
