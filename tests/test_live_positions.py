"""The list of live (step, row) positions (uic_topdown_batch.live_rows / live_count): the fused step's logit layer and criterion
run over the positions whose mask is not zero only.  LanguageModelCriterion multiplies every other position by zero
(reference misc/utils.py:62-73), so the step's loss and gradients are the ones of the full computation -- checked here against
the full computation of the library itself, against the reference's golden vectors and against the CPU oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_live_positions_lists_the_unmasked_positions_step_major():
    from unpaired_image_captioning_amd.topdown_engine import live_positions
    rng = np.random.RandomState(3)
    N, T = 37, 9
    masks = np.zeros((N, T + 1), dtype=np.float32)
    for n in range(N):
        masks[n, :rng.randint(1, T + 2)] = 1.0
    masks[5, :] = 0.0                                    # a row without a caption
    rows, count = live_positions(masks)
    assert rows.dtype == torch.int32 and count.dtype == np.int32 and count.shape == (T,)
    want = [t * N + n for t in range(T) for n in range(N) if masks[n, 1 + t] != 0]
    assert rows.numel() % 128 == 0 and rows.numel() - len(want) < 128
    assert rows[:len(want)].tolist() == want
    assert (rows[len(want):] == -1).all()
    assert count.tolist() == [int((masks[:, 1 + t] != 0).sum()) for t in range(T)]
    # nothing live: an empty list, not a failure
    rows0, count0 = live_positions(np.zeros((4, 6), dtype=np.float32))
    assert rows0.numel() == 0 and count0.tolist() == [0] * 5


def test_batch_struct_layout_matches_the_header(tmp_path):
    """ctypes mirror of uic_topdown_batch == the C compiler's layout of include/uic_hip.h (field offsets and size)."""
    from unpaired_image_captioning_amd import _lib
    fields = [f for f, _ in _lib.Batch._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "uic_hip.h"\nint main(void) {\n' +
                   "".join('  printf("%%zu\\n", offsetof(uic_topdown_batch, %s));\n' % f for f in fields) +
                   '  printf("%zu\\n", sizeof(uic_topdown_batch));\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[:-1] == [getattr(_lib.Batch, f).offset for f in fields]
    assert out[-1] == C.sizeof(_lib.Batch)
    assert fields[-2:] == ["live_rows", "live_count"]


@pytest.mark.gpu
def test_batch_struct_refuses_a_list_of_the_wrong_shape():
    from unpaired_image_captioning_amd.topdown_engine import TopDownEngine, live_positions
    eng = TopDownEngine.__new__(TopDownEngine)
    eng._pad_att = lambda a: a
    fc, att = torch.zeros(4, 8).cuda(), torch.zeros(4, 3, 8).cuda()
    labels = torch.zeros(4, 7, dtype=torch.int64).cuda()
    masks = torch.ones(4, 7).cuda()
    rows, count = live_positions(masks)
    assert rows.is_cuda
    b = eng.batch_struct(fc, att, None, labels, masks, live=(rows, count))
    assert b.live_rows == rows.data_ptr() and b.live_count[0] == 4
    b = eng.batch_struct(fc, att, None, labels, masks, live=(None, count))      # counts only: the step makes the list
    assert not b.live_rows and b.live_count[5] == 4
    with pytest.raises(ValueError):
        eng.batch_struct(fc, att, None, labels, masks, live=(rows, count[:-1]))
    with pytest.raises(ValueError):
        eng.batch_struct(fc, att, None, labels, masks, live=(rows[:64], count))
    with pytest.raises(ValueError):
        eng.batch_struct(fc, att, None, labels, masks, live=(rows.to(torch.int64), count))
    with pytest.raises(ValueError):
        eng.batch_struct(fc, att, None, labels, masks, live=(rows.cpu(), count))


# ---------------------------------------------------------------------------------------------------------------- GPU
def _step(model, batch, live, seed=11):
    from unpaired_image_captioning_amd.trainer import Trainer, xe_step
    b = {k: v for k, v in batch.items() if not k.startswith("live_")}
    if live == "rows":                                   # the caller's own list
        from unpaired_image_captioning_amd.topdown_engine import live_positions
        b["live_rows"], b["live_count"] = live_positions(b["masks"])
    elif live:                                           # counts only (what Trainer does): the step compacts the masks itself
        Trainer.attach_live(b)
        assert "live_rows" not in b
    model._seed_counter = seed                           # (the same dropout masks in every call)
    loss, grads = xe_step(model, b)
    torch.cuda.synchronize()
    return loss.item(), {k: v.detach().clone() for k, v in grads.items()}


def _same(g1, g0, tol):
    floor = 1e-3 * max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        err = (g1[k].double() - g0[k].double()).abs().max().item()
        assert err <= tol * max(g0[k].abs().max().item(), floor), (k, err, g0[k].abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("form", ["rows", "device"])
@pytest.mark.parametrize("name", ["topdown_tiny", "topdown_tiny_ragged", "topdown_tiny_earlybreak", "topdown_odd", "topdown_tiny_box"])
def test_step_over_live_positions_vs_reference_golden(name, dtype, form):
    from test_gpu_topdown import GRAD_TOL, LOGP_TOL, build_model, grads_close
    cfg, W, I, Out, G, X = load_golden(name)
    model = build_model(cfg, W, dtype)
    model.train()
    batch = {k: I[k].cuda() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks") if I.get(k) is not None}
    loss, grads = _step(model, batch, live=form)
    assert abs(loss - float(Out["loss"])) < (1e-4 if dtype == "f32" else LOGP_TOL[dtype])
    grads_close(grads, G, GRAD_TOL[dtype])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [dict(V=1000, E=96, H=160, A=128, D=200, L=9, n_img=14, S=5, R=50),
                                   dict(V=2047, E=128, H=128, A=128, D=256, L=12, n_img=40, S=5, R=36),
                                   dict(V=300, E=64, H=64, A=64, D=64, L=6, n_img=3, S=1, R=7),
                                   # 1050 rows x 17 steps = 17 850 positions: past the one-round-trip list kernel's 16 384
                                   dict(V=300, E=64, H=64, A=64, D=64, L=16, n_img=210, S=5, R=7)])
def test_step_over_live_positions_equals_the_step_over_all(shape, dtype):
    """Same weights, same batch, same dropout seed: the step with the list and the step without it agree to summation order (the
    weight gradient sums the same products over fewer, reordered rows; d hdrop rows are the same dot products)."""
    from oracle import topdown as O
    from test_gpu_topdown import make_opt
    from unpaired_image_captioning_amd import models
    torch.manual_seed(5)
    model = models.setup(make_opt(shape, dtype, drop=0.5, seed=4)).cuda().train()
    b = O.synthetic_batch(shape["n_img"], shape["S"], shape["R"], shape["D"], shape["V"], shape["L"], seed=9, ragged_regions=True)
    batch = {k: v.cuda() for k, v in b.items()}
    assert (b["masks"][:, 1:] == 0).any()
    l0, g0 = _step(model, batch, live=False)
    l1, g1 = _step(model, batch, live="rows")
    l2, g2 = _step(model, batch, live="device")
    assert abs(l1 - l0) < (2e-6 if dtype == "f32" else 2e-5) * max(1.0, abs(l0))
    _same(g1, g0, 2e-5 if dtype == "f32" else 2e-3)
    # (the device's list is the host's -- ascending positions -- but with it the step also knows the captions' lengths and its
    # attention accumulation sums over each row's own steps only: the same sums, paired differently in the last bits)
    assert abs(l2 - l1) <= 1e-6 * max(1.0, abs(l1))
    _same(g2, g1, 2e-5 if dtype == "f32" else 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_arbitrary_masks_holes_empty_steps_and_fractional_weights(dtype):
    """The list is defined by mask != 0, not by caption lengths: masks with holes, a decode step without a single live row (an
    empty chunk of the logit layer), fractional mask weights and a row without any live position give the step over all positions."""
    from oracle import topdown as O
    from test_gpu_topdown import make_opt
    from unpaired_image_captioning_amd import models
    shape = dict(V=500, E=64, H=64, A=64, D=96, L=11, n_img=9, S=5, R=12)
    torch.manual_seed(8)
    model = models.setup(make_opt(shape, dtype, drop=0.5, seed=2)).cuda().train()
    b = O.synthetic_batch(shape["n_img"], shape["S"], shape["R"], shape["D"], shape["V"], shape["L"], seed=3)
    g = torch.Generator().manual_seed(4)
    T = b["masks"].shape[1] - 1
    m = (torch.rand(b["masks"].shape, generator=g) < 0.6).float() * (0.25 + torch.rand(b["masks"].shape, generator=g))
    m[:, 1 + 4] = 0.0                                     # decode step 4: nothing live
    m[:, 1 + 5] = 0.0                                     # ... nor step 5 (steps 4-7 are one chunk: partly empty)
    m[7, :] = 0.0                                         # a row that contributes nothing
    m[:, 1 + T - 1] = 0.0
    m[3, 1 + T - 1] = 1.0                                 # the last step: one live row (the chunk in front of the BPTT loop)
    b["labels"][3, 1:] = torch.randint(1, shape["V"], (T,), generator=g)     # (so that every step is run: no early break)
    b["masks"] = m
    batch = {k: v.cuda() for k, v in b.items()}
    assert model._steps_to_run(batch["labels"]) == T
    l0, g0 = _step(model, batch, live=False)
    l1, g1 = _step(model, batch, live="rows")
    l2, g2 = _step(model, batch, live="device")
    assert abs(l1 - l0) < (2e-6 if dtype == "f32" else 2e-5) * max(1.0, abs(l0))
    _same(g1, g0, 2e-5 if dtype == "f32" else 2e-3)
    assert abs(l2 - l1) <= 1e-6 * max(1.0, abs(l1))
    _same(g2, g1, 2e-5 if dtype == "f32" else 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_batch_without_a_live_position(dtype):
    """Every mask zero: the list is empty, nothing of the logit layer runs; every gradient is zero and the loss is 0 / 0 as the
    reference's criterion gives it (a vocabulary of 501 words: the cleared gradient tensors are not multiples of 16 bytes)."""
    from oracle import topdown as O
    from test_gpu_topdown import make_opt
    from unpaired_image_captioning_amd import models
    shape = dict(V=500, E=64, H=64, A=64, D=96, L=7, n_img=4, S=5, R=9)
    torch.manual_seed(8)
    model = models.setup(make_opt(shape, dtype, drop=0.0, seed=2)).cuda().train()
    b = O.synthetic_batch(shape["n_img"], shape["S"], shape["R"], shape["D"], shape["V"], shape["L"], seed=3)
    b["masks"] = torch.zeros_like(b["masks"])
    batch = {k: v.cuda() for k, v in b.items()}
    for form in ("rows", "device"):
        loss, grads = _step(model, batch, live=form)
        assert loss != loss                               # NaN
        for k, g in grads.items():
            assert float(g.abs().max()) == 0.0, (form, k)


@pytest.mark.gpu
def test_full_size_step_over_live_positions_bf16():
    """BASELINE config 2 (640 caption rows, 9488 words): with and without the list."""
    from oracle import topdown as O
    from test_gpu_topdown import build_model
    V, E, H, A, D, L = 9487, 512, 512, 512, 2048, 16
    Wt = O.init_weights(V + 1, E, H, A, D, D, seed=7)
    b = O.synthetic_batch(128, 5, 36, D, V, L, seed=1234)
    batch = {k: v.cuda() for k, v in b.items()}
    model = build_model(dict(V=V, E=E, H=H, A=A, D=D, L=L), Wt, "bf16").eval()
    l0, g0 = _step(model, batch, live=False)
    l1, g1 = _step(model, batch, live=True)
    assert abs(l1 - l0) < 2e-5 * abs(l0)
    _same(g1, g0, 2e-3)


@pytest.mark.gpu
def test_list_is_ignored_where_the_step_cannot_use_it():
    """Scheduled sampling and hidden logit blocks compute every position (include/uic_hip.h): the list changes nothing there."""
    from test_gpu_topdown import build_model
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_logit2")
    model = build_model(cfg, W, "f32").eval()
    batch = {k: I[k].cuda() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks") if I.get(k) is not None}
    l0, g0 = _step(model, batch, live=False)
    l1, g1 = _step(model, batch, live=True)
    assert l0 == l1
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


@pytest.mark.gpu
def test_trainer_attaches_the_list_to_host_batches_and_trains_the_same():
    """Trainer.to_device builds the list from the host masks; opt.live_positions = 0 turns it off; both trajectories agree."""
    import argparse
    from oracle import topdown as O
    from unpaired_image_captioning_amd.trainer import Trainer
    cfg = dict(V=500, E=64, H=64, A=64, D=96, L=8)

    def run(live):
        torch.manual_seed(1)
        opt = argparse.Namespace(vocab_size=cfg["V"], input_encoding_size=cfg["E"], rnn_size=cfg["H"], num_layers=1, drop_prob_lm=0.0,
                                 seq_length=cfg["L"], fc_feat_size=cfg["D"], att_feat_size=cfg["D"], att_hid_size=cfg["A"], use_bn=0,
                                 logit_layers=1, caption_model="topdown", compute_dtype="f32", seed=3, learning_rate=5e-4,
                                 optim="adam", optim_alpha=0.9, optim_beta=0.999, optim_epsilon=1e-8, weight_decay=0, grad_clip=0.1,
                                 seq_per_img=1, live_positions=live)
        tr = Trainer(opt)
        losses = []
        for i in range(3):
            b = O.synthetic_batch(6, 5, 9, cfg["D"], cfg["V"], cfg["L"], seed=20 + i)
            data = {k: v.numpy() for k, v in b.items()}
            dev = tr.to_device(data, per_image=False)
            assert ("live_count" in dev) == bool(live) and "live_rows" not in dev
            losses.append(tr.train(data))
        return losses, {k: v.detach().clone() for k, v in tr.i2t_model.state_dict().items()}

    l0, w0 = run(0)
    l1, w1 = run(1)
    assert np.allclose(l0, l1, rtol=0, atol=2e-5)
    for k in w0:
        assert (w0[k] - w1[k]).abs().max().item() <= 2e-5, k
