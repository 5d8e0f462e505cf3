#!/usr/bin/env python3
"""Golden vectors for the self-critical reward from the REFERENCE's own scorer:
``misc/cider/pyciderevalcap/ciderD/{ciderD,ciderD_scorer}.py`` and ``misc/rewards.py::get_self_critical_reward``.

Build-container only (needs /root/reference; nothing is written there).  The scorer files are python 2 (``xrange``,
``dict.iteritems``): they are run through the standard library's ``lib2to3`` fixers IN MEMORY (four lines change) and
executed as modules; ``rewards.py`` is py3-clean and is imported as is, with stub modules for what it imports but never
calls here (``nltk`` behind ``misc.utils``, the coco-caption ``Bleu`` scorer -- bleu_reward_weight is 0) and a fake
captioner that returns a preset greedy decode.  The cached document-frequency table is built the way
``scripts/prepro_ngrams.py:69-125`` builds it (n-gram -> number of images whose references contain it; ``ref_len`` =
the raw image COUNT, which is what that script stores and the scorer then uses un-logged).

    python tests/golden/make_golden_cider.py        # writes tests/golden/ciderd_*.npz
"""
import argparse
import os
import sys
import types
import warnings
from collections import defaultdict

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

P = "/root/reference/pivot_based_eccv2018"
C = os.path.join(P, "misc", "cider", "pyciderevalcap", "ciderD")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_py2(modname, path):
    """Execute a python-2 source file as module `modname` after lib2to3's standard fixers (in memory)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        rt = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
        src = str(rt.refactor_string(open(path).read() + "\n", path))
    mod = types.ModuleType(modname)
    mod.__file__ = path
    mod.__package__ = modname.rpartition(".")[0]       # lib2to3 turns the py2 implicit relative import into `from .x import`
    sys.modules[modname] = mod
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


def load_reference():
    for name in ("pyciderevalcap", "pyciderevalcap.ciderD"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    scorer = load_py2("pyciderevalcap.ciderD.ciderD_scorer", os.path.join(C, "ciderD_scorer.py"))
    ciderd = load_py2("pyciderevalcap.ciderD.ciderD", os.path.join(C, "ciderD.py"))
    for name in ("pyciderevalcap", "pyciderevalcap.ciderD", "coco_caption", "coco_caption.pycxevalcap",
                 "coco_caption.pycxevalcap.bleu", "coco_caption.pycxevalcap.bleu.bleu", "nltk", "nltk.translate",
                 "nltk.translate.bleu_score"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["coco_caption.pycxevalcap.bleu.bleu"].Bleu = type("Bleu", (), {"__init__": lambda self, n: None})
    sys.modules["nltk.translate.bleu_score"].SmoothingFunction = object
    sys.path.insert(0, P)
    import misc.rewards as rewards
    return scorer, ciderd, rewards


def words(row):
    out = []
    for t in row:
        out.append(str(int(t)))
        if int(t) == 0:
            break
    return out


def random_captions(g, n, L, V, p_short=0.7):
    """Label-style rows: tokens 1..V, zero padded; some rows fill all L positions (no terminating 0)."""
    rows = np.zeros((n, L), dtype=np.int64)
    for i in range(n):
        ln = L if g.random() > p_short else int(g.integers(1, L))
        rows[i, :ln] = g.integers(1, V + 1, ln)
    return rows


def make_case(name, scorer_mod, ciderd_mod, rewards, seed, V, L, n_img, S, mode, n_corpus=50, zipf=True):
    g = np.random.default_rng(seed)

    def caps(n):
        r = random_captions(g, n, L, V)
        if zipf:                       # skew towards small ids so that n-grams repeat within and across captions
            r = np.where(r > 0, np.minimum(r, g.integers(1, V + 1, r.shape)), 0)
        return r
    gts = [caps(int(g.integers(2, 6))) for _ in range(n_img)]
    N = n_img * S
    gen = caps(N)
    greedy = caps(N)
    # make some hypotheses share n-grams with their references, some end at once, some repeat a word
    for i in range(0, N, 2):
        ref = gts[i // S][0]
        gen[i, :L // 2] = ref[:L // 2]
    gen[1] = 0
    greedy[2, :] = greedy[2, 0]
    greedy[2, L // 2:] = 0
    greedy[0] = gts[0][-1]

    CiderD = ciderd_mod.CiderD
    sc = CiderD(df="corpus")
    out = {}
    if mode != "corpus":
        corpus = [caps(int(g.integers(2, 6))) for _ in range(n_corpus)] + gts[: n_img // 2]
        crefs = [[scorer_mod.precook(" ".join(words(r))) for r in img] for img in corpus]
        df = defaultdict(float)
        for refs in crefs:                                   # prepro_ngrams.py:69-80
            for ngram in set(ng for ref in refs for ng in ref):
                df[ngram] += 1
        ref_len = float(len(corpus)) if mode == "cached" else float(np.log(float(len(corpus))))
        sc.cider_scorer.df_mode = "cached"                   # what CiderScorer.__init__ does after reading the pickle (:64-67)
        sc.cider_scorer.ref_len = ref_len
        sc.cider_scorer.document_frequency = df
        keys = np.full((len(df), 4), -1, dtype=np.int32)
        cnt = np.zeros(len(df), dtype=np.float64)
        # (rows in a fixed order -- by n-gram -- so that the fixture regenerates bit for bit: the dict's own order follows
        # python's per-process string hashing)
        for j, (k, v) in enumerate(sorted(df.items(), key=lambda kv: tuple(int(x) for x in kv[0]))):
            keys[j, :len(k)] = [int(x) for x in k]
            cnt[j] = v
        out["df_keys"], out["df_count"], out["ref_len"] = keys, cnt, np.array(ref_len)

    class FakeModel(object):
        def eval(self): pass
        def train(self): pass
        def __call__(self, *a, **k):
            assert k.get("mode") == "sample"
            return torch.from_numpy(greedy), None
    rewards.CiderD_scorer = sc
    opt = argparse.Namespace(cider_reward_weight=1.0, bleu_reward_weight=0)
    data = {"gts": gts}
    z = torch.zeros(1)
    reward = rewards.get_self_critical_reward(FakeModel(), z, z, z, z, data, torch.from_numpy(gen), opt)
    # the scorer level, same inputs as get_self_critical_reward assembles (rewards.py:51-63)
    res = [{"image_id": i, "caption": [" ".join(words(gen[i] if i < N else greedy[i - N]))]} for i in range(2 * N)]
    gts_d = {i: [" ".join(words(r)) for r in gts[i % N // S]] for i in range(2 * N)}
    mean, scores = sc.compute_score(gts_d, res)
    assert np.allclose(reward[:, 0], scores[:N] - scores[N:], rtol=0, atol=0)
    out.update(gen=gen, greedy=greedy, gts_tok=np.concatenate(gts, 0),
               gts_start=np.cumsum([0] + [len(x) for x in gts]).astype(np.int32), seq_per_img=np.array(S),
               scores=np.asarray(scores, dtype=np.float64), mean=np.array(mean), reward=np.asarray(reward, dtype=np.float64))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) mean CIDEr-D %.6f, reward range [%.4f, %.4f]" % (path, os.path.getsize(path) / 1024, mean,
                                                                             reward.min(), reward.max()))


def main():
    scorer_mod, ciderd_mod, rewards = load_reference()
    make_case("ciderd_cached", scorer_mod, ciderd_mod, rewards, 1, V=12, L=8, n_img=4, S=3, mode="cached")
    make_case("ciderd_corpus", scorer_mod, ciderd_mod, rewards, 2, V=12, L=8, n_img=4, S=3, mode="corpus")
    make_case("ciderd_logreflen", scorer_mod, ciderd_mod, rewards, 3, V=9, L=7, n_img=3, S=2, mode="cached_log", n_corpus=30)
    make_case("ciderd_real_shape", scorer_mod, ciderd_mod, rewards, 4, V=60, L=16, n_img=6, S=5, mode="cached", n_corpus=200)


if __name__ == "__main__":
    main()
