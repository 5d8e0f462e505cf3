#!/usr/bin/env python3
"""Persistent BPTT kernel (csrc/rnn_bwd_persist.hip) against the per-step launch chain on the same inputs: every gradient
tensor and every per-step buffer the BPTT loop leaves behind, run-to-run determinism, time of the backward call alone, the
fused training step in both modes (with its timing marks), optional per-phase time stamps.

    python tools/rnn_bwd_probe.py [--n-img 128] [--dbg] [--ragged]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--n-img", type=int, default=128)
ap.add_argument("--dbg", action="store_true")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--ragged", action="store_true")
ap.add_argument("--no-step", action="store_true")
args = ap.parse_args()

import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd import models, trainer
from unpaired_image_captioning_amd.synthetic import synthetic_batch

lib = L.load()
c = CFG
torch.manual_seed(1234)
model = models.setup(make_opt("bf16", 1234)).cuda()
model.train()
eng = model.engine
batch = synthetic_batch(args.n_img, c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234, ragged_regions=args.ragged)
batch = {k: v.cuda() for k, v in batch.items()}
N, T, H, R = args.n_img * c["S"], c["L"] + 1, c["H"], c["R"]
t_run = model._steps_to_run(batch["labels"])
params = {k: v.detach() for k, v in model.param_dict().items()}
bf = torch.bfloat16
NAMES = [("dg1", (T, N, 4 * H), bf), ("dg2", (T, N, 4 * H), bf), ("datth", (T, N, H), bf), ("de", (T, N, R), torch.float32),
         ("dx2", (T, N, 3 * H), torch.float32)]
masks = batch["att_masks"] if args.ragged else None


def fwd():
    logp, ws, (d, w, b) = eng.forward(params, batch["fc_feats"], batch["att_feats"], masks, batch["labels"], t_run, True, 77,
                                      want_logprobs=False, masks=batch["masks"])
    eng.xe_loss(ws, d, b, t_run)
    return ws, d, w, b


# mode 3: BPTT as six launches per step (the default), 4: the persistent BPTT kernel
def set_mode(mode):
    eng.recurrence = (L.REC_BWD_PERSIST if mode >= 4 else 0) | (L.REC_STAMPS if args.dbg else 0)


def run(mode):
    set_mode(mode)
    ws, d, w, b = fwd()
    d = eng.dims(N, R, T, d.seq_per_img)
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    eng.backward(ws, d, w, b, t_run, True, 77, grads)
    torch.cuda.synchronize()
    out = {n: eng.workspace_tensor(ws, n, shp, dt)[:t_run].float().clone() for n, shp, dt in NAMES}
    out["dx2"] = out["dx2"][:, :, :H].contiguous()      # only the d att_res columns outlive the loop
    for k, g in grads.items():
        out["grad:" + k] = g.float().clone()
    dbg = eng.workspace_tensor(ws, "rnn_bwd_dbg", (256, T, 16), torch.int64).clone() if args.dbg and mode >= 4 else None
    eng.release(ws)
    return out, dbg


def time_backward(mode):
    set_mode(mode)
    ws, d, w, b = fwd()
    d = eng.dims(N, R, T, d.seq_per_img)
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    for _ in range(3):
        eng.backward(ws, d, w, b, t_run, True, 77, grads)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        eng.backward(ws, d, w, b, t_run, True, 77, grads)
    e1.record()
    torch.cuda.synchronize()
    eng.release(ws)
    return e0.elapsed_time(e1) / args.iters


def time_step(mode, steps=30):
    set_mode(mode)
    for _ in range(5):
        loss, _ = trainer.xe_step(model, batch)
        loss.item()
    L.check(lib.uic_topdown_step_marks(1, None))
    tot = [0.0] * 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    wall = 0.0
    for _ in range(steps):
        e0.record()
        loss, _ = trainer.xe_step(model, batch)
        e1.record()
        loss.item()
        torch.cuda.synchronize()
        wall += e0.elapsed_time(e1)
        m = (C.c_float * L.STEP_MARKS)()
        L.check(lib.uic_topdown_step_marks(1, m))
        for i in range(10):
            tot[i] += m[i]
    L.check(lib.uic_topdown_step_marks(0, None))
    return wall / steps, [x / steps for x in tot]


res = {"N": N, "t_run": t_run}
ref, _ = run(3)
print("status after chain:", L.persistent_status())
got, dbg = run(4)
again, _ = run(4)
st = L.persistent_status()
print("status after persistent BPTT:", st)
worst = 0.0
for k in ref:
    den = max(float(ref[k].abs().max()), 1e-30)
    rel = float((got[k] - ref[k]).abs().max()) / den
    l2 = float((got[k] - ref[k]).norm()) / max(float(ref[k].norm()), 1e-30)
    rep = torch.equal(got[k], again[k])
    worst = max(worst, l2)
    print("   %-32s max|diff|/max|ref| %.3e   L2 rel %.3e   max|ref| %.3e  repeatable %s" % (k, rel, l2, den, rep))
res["worst_l2_rel"] = worst
if dbg is not None:
    d = dbg[:, :t_run].double() * 10e-3          # 100 MHz ticks -> us
    names = ["A cell(lang)", "B arrive+wait1", "B gemm dx2", "B reduce+arrive", "C wait2(+slot)", "C attention", "D wait3", "D gemm+cell", "E wait4", "E gemm dx1"]
    for k in range(10):
        seg = d[:, 1:, k + 1] - d[:, 1:, k]
        print("   phase %-16s mean %6.2f us   median %6.2f   max over WGs (mean over t) %6.2f" %
              (names[k], seg.mean().item(), seg.median().item(), seg.max(dim=0)[0].mean().item()))
    step = d[:, :-1, 0] - d[:, 1:, 0]            # steps run backwards: stamp 0 of step t-1 comes after stamp 0 of step t
    print("   step   mean %6.2f us  (WG0, latest steps first: %s)" % (step.mean().item(), [round(x, 2) for x in step[0, -5:].tolist()]))
    res["step_us"] = float(step.mean())
    res["phases_us"] = {names[k]: float((d[:, 1:, k + 1] - d[:, 1:, k]).mean()) for k in range(10)}
for mode in (3, 4):
    ms = time_backward(mode)
    res["backward_ms_mode%d" % mode] = ms
    print("backward call (logit layer + BPTT + epilogue, one stream) mode %d: %.3f ms" % (mode, ms))
if not args.no_step:
    names = ["start", "prologue done", "recurrence done", "side: logit layer done", "BPTT starts", "BPTT done", "side: recurrent wgrads done",
             "main tail done", "side tail done", "joined"]
    for mode in (3, 4, 3, 4):
        ms, marks = time_step(mode)
        res.setdefault("step_ms_mode%d" % mode, []).append(ms)
        print("fused step mode %d: %.3f ms   " % (mode, ms) + "  ".join("%s %.3f" % (n.split(":")[-1].strip(), v) for n, v in zip(names[1:], marks[1:])))
print("status:", L.persistent_status())
eng.recurrence = 0
print(json.dumps(res))
