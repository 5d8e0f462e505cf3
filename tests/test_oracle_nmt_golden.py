"""Pin oracle/nmt.py against golden vectors produced from the reference's NMT_Models / OpenNMT-fork modules."""
import pytest
import torch

from conftest import GOLDEN
from oracle import nmt as ON


def load(name):
    import numpy as np
    import os
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    W = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    I = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in::")}
    Out = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out::")}
    G = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad::")}
    return W, I, Out, G


def close(a, b, tol=1e-5):
    a, b = a.double(), b.double()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("name", ["nmt_tiny", "nmt_tiny_1layer", "nmt_odd"])
def test_nmt_forward_loss_grads(name):
    W, I, Out, G = load(name)
    loss, grads, aux = ON.loss_and_grads(W, I["src"], I["tgt"], I["lengths"])
    close(aux["context"], Out["context"])
    close(aux["h"], Out["enc_h"])
    close(aux["c"], Out["enc_c"])
    close(aux["outputs"], Out["outputs"])
    close(aux["attn"], Out["attn"])
    close(aux["scores"], Out["scores"])
    assert abs(loss.item() - float(Out["loss"])) < 1e-3
    assert aux["num_correct"] == int(Out["num_correct"]) and aux["num_words"] == int(Out["num_words"])
    assert set(G) == set(grads)
    for k in G:
        close(grads[k], G[k], 2e-5)
    # padding_idx: the PAD rows of both embedding tables receive no gradient (nn.Embedding(padding_idx=PAD))
    assert G["encoder.embeddings.word_lut.weight"][0].abs().max() == 0


@pytest.mark.parametrize("name", ["nmt_translate_tiny", "nmt_translate_odd", "nmt_translate_1layer", "nmt_translate_long"])
def test_nmt_translate_batch_matches_reference(name):
    """NMTModel.translateBatch + onmt Beam (beam 15): hypotheses token for token, final scores, attention of the winning
    hypothesis; incl. a case that runs all 100 steps and cases whose sentences finish at different steps."""
    W, I, Out, G = load(name)
    hyp, scores, attn = ON.translate_batch(W, I["src"])
    assert torch.equal(hyp, Out["hyp"]), (hyp, Out["hyp"])
    assert (scores.double() - Out["scores"]).abs().max().item() < 1e-3 * max(1.0, Out["scores"].abs().max().item())
    close(attn, Out["attn"], 1e-4)
