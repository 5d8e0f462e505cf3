"""The pivot NMT step over the target positions that are not PAD (uic_nmt_dims.tgt_live_rows): NMTCriterion's weight[PAD] = 0
(reference misc/criterion.py:126-136) makes the padded positions exact zeros, so generator, criterion and their gradients run
over the others only -- against the reference's golden vectors, the oracle and the step over all positions."""
import argparse

import numpy as np
import pytest
import torch


def test_tgt_live_positions_lists_the_non_pad_targets():
    from unpaired_image_captioning_amd.models.NMT_Models import tgt_live_positions
    rng = np.random.RandomState(1)
    T, B = 9, 7
    tgt = rng.randint(4, 50, size=(T, B)).astype(np.int64)
    tgt[0] = 2
    for b in range(B):
        n = rng.randint(3, T + 1)
        tgt[n - 1, b] = 3
        tgt[n:, b] = 0
    rows, count = tgt_live_positions(tgt)
    want = [t * B + b for t in range(T - 1) for b in range(B) if tgt[t + 1, b] != 0]
    assert count == len(want) and rows.dtype == torch.int32 and rows.numel() == (len(want) + 127) // 128 * 128
    assert rows[:count].tolist() == want and (rows[count:] == -1).all()
    rows0, count0 = tgt_live_positions(np.zeros((5, 3), dtype=np.int64))
    assert count0 == 0 and rows0.numel() == 0


def test_nmt_dims_layout_matches_the_header(tmp_path):
    import ctypes as C
    import os
    import subprocess
    from unpaired_image_captioning_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f for f, _ in _lib.NmtDims._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "uic_hip.h"\nint main(void) {\n' +
                   "".join('  printf("%%zu\\n", offsetof(uic_nmt_dims, %s));\n' % f for f in fields) +
                   '  printf("%zu\\n", sizeof(uic_nmt_dims));\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[:-1] == [getattr(_lib.NmtDims, f).offset for f in fields]
    assert out[-1] == C.sizeof(_lib.NmtDims)


def _run(model, crit, I, live):
    from unpaired_image_captioning_amd.models.NMT_Models import tgt_live_positions
    model.zero_grad()
    model._seed_counter = 5
    tgt = I["tgt"].cuda()
    if live == "rows":                                       # the caller's own list
        tgt.uic_live = tgt_live_positions(I["tgt"], tgt.device)
    elif live:                                               # the count only (what the Dataset attaches): the step compacts the targets
        from unpaired_image_captioning_amd.models.NMT_Models import tgt_live_count
        tgt.uic_live = (None, tgt_live_count(I["tgt"]))
    batch = argparse.Namespace(src=I["src"].cuda(), tgt=tgt, lengths=I["lengths"])
    outputs, attns, _, _ = model(batch.src, batch.tgt, batch.lengths, None)
    crit.report_stats = type(crit.report_stats)()             # (fresh counters for this call)
    loss = crit(None, batch, outputs, attns)
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), outputs, {k: p.grad.detach().clone() for k, p in model.named_parameters()}, (crit.report_stats.n_words, crit.report_stats.n_correct)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("form", ["rows", "device"])
@pytest.mark.parametrize("name", ["nmt_tiny", "nmt_tiny_1layer", "nmt_odd"])
def test_nmt_step_over_live_positions_vs_reference_golden(name, dtype, form):
    from test_gpu_nmt import GRAD_TOL, OUT_TOL, absmax, build, grads_close, load
    cfg, W, I, Out, G = load(name)
    model, crit = build(cfg, W, dtype)
    model.train()
    loss, outputs, grads, (n_words, n_correct) = _run(model, crit, I, live=form)
    assert absmax(outputs, Out["outputs"]) < OUT_TOL[dtype]
    assert abs(loss - float(Out["loss"])) < OUT_TOL[dtype] * int(Out["num_words"])
    assert n_words == int(Out["num_words"])
    if dtype == "f32":
        assert n_correct == int(Out["num_correct"])
    grads_close(grads, G, GRAD_TOL[dtype])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [dict(layers=2, H=256, W=192, B=24, S=21, T=17, Vs=900, Vt=1100),
                                 dict(layers=1, H=64, W=48, B=5, S=7, T=6, Vs=90, Vt=70),
                                 dict(layers=2, H=512, W=512, B=64, S=30, T=32, Vs=3000, Vt=5003)])
def test_nmt_step_over_live_positions_equals_the_step_over_all(cfg, dtype):
    """Same weights, batch and dropout seed, with the list and without: equal to summation order; the score counters equal."""
    from test_gpu_nmt import build, random_weights, synthetic
    W = random_weights(cfg, 11)
    I = synthetic(cfg, 5)
    assert (I["tgt"][1:] == 0).any()
    model, crit = build(cfg, W, dtype, dropout=0.3)
    model.train()
    l0, o0, g0, s0 = _run(model, crit, I, live=False)
    l1, o1, g1, s1 = _run(model, crit, I, live="rows")
    l2, o2, g2, s2 = _run(model, crit, I, live="device")
    assert l2 == l1 and s2 == s1 and all(torch.equal(g1[k], g2[k]) for k in g1)      # (the device's list is the host's)
    assert torch.equal(o0, o1)                                    # (the forward pass up to the generator is the same launches)
    assert abs(l1 - l0) <= (2e-6 if dtype == "f32" else 2e-5) * max(1.0, abs(l0))
    assert s0 == s1
    floor = 1e-3 * max(float(v.abs().max()) for v in g0.values())
    # (bf16: d outputs differ in f32 summation order, the bf16 stores behind them flip last bits -- 2^-9 per element -- and the
    # BPTT carries that through 30 steps: worst entry measured 4.3e-3 of the tensor's largest)
    tol = 2e-5 if dtype == "f32" else 1e-2
    for k in g0:
        err = (g1[k].double() - g0[k].double()).abs().max().item()
        assert err <= tol * max(g0[k].abs().max().item(), floor), (k, err, g0[k].abs().max().item())


@pytest.mark.gpu
def test_dataset_attaches_the_list_and_opt_turns_it_off(tmp_path):
    """onmt_dataset_h5.Dataset(cuda=True) attaches the list to the target tensor; opt.live_positions = 0 makes NMTModel ignore it."""
    from test_gpu_nmt import build, random_weights, synthetic
    from unpaired_image_captioning_amd.models.NMT_Models import tgt_live_positions
    cfg = dict(layers=1, H=64, W=48, B=5, S=7, T=6, Vs=90, Vt=70)
    W = random_weights(cfg, 3)
    I = synthetic(cfg, 4)
    model, crit = build(cfg, W, "f32")
    model.train()
    l_live, _, g_live, _ = _run(model, crit, I, live=True)
    model.opt.live_positions = 0
    l_off, _, g_off, _ = _run(model, crit, I, live=True)       # (attribute present, switched off)
    l_all, _, g_all, _ = _run(model, crit, I, live=False)
    assert l_off == l_all
    for k in g_all:
        assert torch.equal(g_off[k], g_all[k]), k
    assert abs(l_live - l_all) <= 2e-6 * max(1.0, abs(l_all))
