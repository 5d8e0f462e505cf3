"""BASELINE configs[2], the eval_pivot.py surface, as ONE path: image features -> im2zh TopDown captioner with beam search
(AttModel._sample_beam, P/models/AttModel.py:167-196) -> pivot caption -> zh->en NMT translateBatch (beam 15,
P/models/NMT_Models.py:322-395) -> target caption, at the real widths (36 x 2048 features, hidden 512, caption vocabulary
9 487 + 1, NMT vocabularies 50 004, 2 layers), against the two oracles chained the same way.  f32: tokens must agree
exactly at both stages; bf16: the final beam scores must agree (near-tied candidates may swap)."""
import argparse

import pytest
import torch

from oracle import nmt as ON
from oracle import topdown as O
from test_gpu_nmt import build as build_nmt, recipe_weights
from test_gpu_topdown import build_model

pytestmark = pytest.mark.gpu

V, E, H, A, D, L, R = 9487, 512, 512, 512, 2048, 16, 36
CAP = dict(V=V, E=E, H=H, A=A, D=D, L=L)
STEPS = 9                                           # translator iterations (the oracle's (15 B) x 50 004 GEMM per step runs on the host)


def nmt_cfg(B):
    return dict(layers=2, H=512, W=512, B=B, S=L, T=2, Vs=50004, Vt=50004)


def to_source(seq):
    """pivot caption tokens -> NMT source ids, time-major [S, B, 1]: ids + 4 skip the NMT special tokens, 0 = PAD after the
    caption's end, no empty source sentence (tools/pivot_decode_bench.py)."""
    src = torch.where(seq > 0, seq + 4, torch.zeros_like(seq)).t().contiguous().unsqueeze(2)
    src[0] = torch.where(src[0] == 0, torch.full_like(src[0], 5), src[0])
    return src


@pytest.fixture(scope="module", params=[8, 64], ids=["B8", "B64"])       # 64 = BASELINE configs[2]'s batch
def weights(request):
    NMT = nmt_cfg(request.param)
    Wc = O.init_weights(V + 1, E, H, A, D, D, seed=3)
    Wc["logit.weight"] *= 30.0                      # a peaked word distribution ...
    Wc["logit.bias"][0] += 0.8                      # ... in which EOS (= 0) wins at once for the weakest images and never for the rest
    Wn = recipe_weights(NMT, 9, 0.25)
    Wn["generator.0.bias"][3] += 2.0               # sentences that end at different steps
    b = O.synthetic_batch(NMT["B"], 1, R, D, V, L, seed=21, ragged_regions=True)
    sc = 1 + 2.0 * (torch.arange(NMT["B"]) % 8).float()   # images of very different feature magnitude: the captions differ
    b["att_feats"] = b["att_feats"] * sc[:, None, None]
    b["fc_feats"] = b["fc_feats"] * sc[:, None]
    return Wc, Wn, b, NMT


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_pivot_decode_captioner_beam_then_translate(weights, dtype):
    Wc, Wn, b, NMT = weights
    nt = torch.get_num_threads()
    torch.set_num_threads(min(16, nt))
    cap = build_model(CAP, Wc, dtype).eval()
    nmt, _ = build_nmt(NMT, Wn, dtype)
    nmt.eval()
    with torch.no_grad():
        seq, seq_lp = cap(b["fc_feats"].cuda(), None, b["att_feats"].cuda(), b["att_masks"].cuda(), opt={"beam_size": 3}, mode="sample")
        src = to_source(seq)
        allHyp, allScores, allAttn, gold = nmt.translateBatch(argparse.Namespace(src=src, batchSize=NMT["B"]), max_steps=STEPS)
    torch.cuda.synchronize()
    # stage 1 against the oracle's beam search on the same features
    seq_o, lp_o = O.sample_beam(Wc, b["fc_feats"], b["att_feats"], b["att_masks"], L, 3)
    lens = (seq_o > 0).sum(1)
    assert len(set(lens.tolist())) > 1             # empty and full-length pivot captions in one batch
    if dtype == "f32":
        assert torch.equal(seq.cpu(), seq_o)
        assert (seq_lp.cpu() - lp_o).abs().max().item() < 1e-3
    else:
        same = (seq.cpu() == seq_o).all(1).float().mean().item()
        assert same >= 0.5, same
        # and the numbers of the device's OWN captions: the oracle, walked along the device's tokens, must give every recorded
        # step log-prob (the beam's -1000 bookkeeping entries aside) within the bf16 tolerance
        _, lp_f = O.sample(Wc, b["fc_feats"], b["att_feats"], b["att_masks"], L, sample_max=0, forced_tokens=seq.cpu())
        live = (seq_lp.cpu() > -100) & (torch.cat([torch.ones(seq.shape[0], 1, dtype=torch.bool), seq.cpu()[:, :-1] > 0], 1))
        # (logit.weight x 30 above: the logits span ~60 here, a bf16 rounding step of the hidden state moves one by 1.2e-2 measured)
        assert (seq_lp.cpu() - lp_f)[live].abs().max().item() < 2e-2
    # stage 2 against the oracle's translator ON THE DEVICE'S pivot captions (so a bf16 tie-swap in stage 1 is not counted twice)
    hyp_o, scores_o, attn_o = ON.translate_batch(Wn, src.cpu(), max_steps=STEPS)
    got = torch.stack([s_[0] for s_ in allScores]).cpu()
    assert len(allHyp[0][0]) == hyp_o.shape[1]
    if dtype == "f32":
        assert [h[0] for h in allHyp] == [[int(t) for t in row] for row in hyp_o]
        assert (got - scores_o).abs().max().item() < 2e-3 * max(1.0, scores_o.abs().max().item())
        for k in range(NMT["B"]):
            a = allAttn[k][0].cpu()
            assert (a - attn_o[k, :, :a.shape[1]]).abs().max().item() < 1e-3
    else:
        assert (got - scores_o).abs().max().item() < 0.15 * max(1.0, scores_o.abs().max().item())
        # (beam 15 over 50 004 near-tied words at random weights: bf16 swaps candidates, so token identity is only asked of
        # some sentences -- 3 of 8 measured -- while every final beam score must agree)
        n_same = sum(int(allHyp[k][0] == [int(t) for t in hyp_o[k]]) for k in range(NMT["B"]))
        assert n_same >= max(2, NMT["B"] // 8), n_same
    torch.set_num_threads(nt)
