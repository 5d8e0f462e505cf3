#!/usr/bin/env python3
"""Persistent recurrence kernel (csrc/rnn_persist.hip) against the per-step launch chain on the same inputs:
activations of every decode step, log-probs, run-to-run determinism, forward time, optional per-phase time stamps.

    python tools/rnn_persist_probe.py [--dtype bf16] [--n-img 128] [--train 1] [--dbg]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--n-img", type=int, default=128)
ap.add_argument("--train", type=int, default=1)
ap.add_argument("--dbg", action="store_true")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--ragged", action="store_true")
args = ap.parse_args()

import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd import models
from unpaired_image_captioning_amd.synthetic import synthetic_batch

lib = L.load()
c = CFG
torch.manual_seed(1234)
model = models.setup(make_opt(args.dtype, 1234)).cuda()
model.train(bool(args.train))
eng = model.engine
batch = synthetic_batch(args.n_img, c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234, ragged_regions=args.ragged)
N, T, H, R = args.n_img * c["S"], c["L"] + 1, c["H"], c["R"]
t_run = model._steps_to_run(batch["labels"])
params = model.param_dict()
td = L.TORCH_DTYPE[L.dtype_id(args.dtype)]

NAMES = [("h_att", (T + 1, N, H), td), ("h_lang", (T + 1, N, H), td), ("c_att", (T + 1, N, H), torch.float32),
         ("c_lang", (T + 1, N, H), torch.float32), ("att_h", (T, N, H), torch.float32), ("alpha", (T, N, R), torch.float32),
         ("ctx", (T, N, H), td), ("hdrop", (T, N, H), td), ("gates1", (T, N, 4 * H), td), ("gates2", (T, N, 4 * H), td)]


# mode: 0 per-step launches, 1 persistent kernel, 2 persistent kernel with the SAFE exchange protocol
REC = {0: L.REC_FWD_CHAIN, 1: 0, 2: L.REC_SAFE}


def run(mode, want_lp=True):
    eng.recurrence = REC[mode] | (L.REC_STAMPS if args.dbg else 0)
    logp, ws, _ = eng.forward(params, batch["fc_feats"], batch["att_feats"], batch["att_masks"] if args.ragged else None,
                              batch["labels"], t_run, args.train, 77, want_logprobs=want_lp)
    out = {n: eng.workspace_tensor(ws, n, shp, dt)[: (t_run + 1 if shp[0] == T + 1 else t_run)].float().clone() for n, shp, dt in NAMES}
    if want_lp:
        out["logp"] = logp[:, :t_run].clone()
    dbg = None
    if args.dbg and mode:
        dbg = eng.workspace_tensor(ws, "rnn_dbg", (256, T, 16), torch.int64).clone()
    torch.cuda.synchronize()
    eng.release(ws)
    return out, dbg


def timeit(mode):
    eng.recurrence = REC[mode]
    for _ in range(3):
        _, ws, _ = eng.forward(params, batch["fc_feats"], batch["att_feats"], None, batch["labels"], t_run, args.train, 77, want_logprobs=False)
        eng.release(ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        _, ws, _ = eng.forward(params, batch["fc_feats"], batch["att_feats"], None, batch["labels"], t_run, args.train, 77, want_logprobs=False)
        eng.release(ws)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


res = {"dtype": args.dtype, "N": N, "t_run": t_run, "train": args.train}
ref, _ = run(0)
print("status after chain:", L.persistent_status())
for mode in (1, 2):
    got, dbg = run(mode)
    st = L.persistent_status()
    diffs = {k: float((got[k] - ref[k]).abs().max()) for k in ref}
    again, _ = run(mode)
    same = all(torch.equal(got[k], again[k]) for k in got)
    res["mode%d" % mode] = {"max_abs_diff_vs_chain": diffs, "bitwise_repeatable": same, "status": st}
    print("mode", mode, "status", st, "repeatable", same)
    for k, v in diffs.items():
        print("   %-8s max|diff| %.3e   (max|ref| %.3e)" % (k, v, float(ref[k].abs().max())))
    if dbg is not None and mode == 1:
        d = dbg[:, :t_run].double() * 10e-3          # 100 MHz ticks -> us
        names = ["lstm1", "bar1", "h2att", "bar2", "attn", "bar3", "lstm2", "bar4(next t0)"]
        for k in range(7):
            seg = d[:, :, k + 1] - d[:, :, k]
            print("   phase %-6s mean %6.2f us   median %6.2f   max over WGs (mean over t) %6.2f" %
                  (names[k], seg.mean().item(), seg.median().item(), seg.max(dim=0)[0].mean().item()))
        if args.dtype == "bf16":
            def seg(a, b):
                return (d[:, :, b] - d[:, :, a]).mean().item()
            print("   lstm1 (wave 0): loads+MFMA of own tile %.2f, cell %.2f, fifth tile %.2f" % (seg(0, 8), seg(8, 9), seg(9, 1)))
            print("   lstm2 (wave 0): to pass0 barrier %.2f, then per pass %s, tail %.2f" %
                  (seg(6, 11), [round(seg(11 + i, 12 + i), 2) for i in range(4)], seg(15, 7)))
        step = d[:, 1:, 0] - d[:, :-1, 0]
        print("   step   mean %6.2f us  (first WG0 steps: %s)" % (step.mean().item(), [round(x, 2) for x in step[0, :5].tolist()]))
        tot = (d[:, t_run - 1, 7] - d[:, 0, 0])
        print("   whole recurrence per WG: mean %.1f us  max %.1f us" % (tot.mean().item(), tot.max().item()))
        res["phases_us"] = {names[k]: float((d[:, :, k + 1] - d[:, :, k]).mean()) for k in range(7)}
        res["step_us"] = float(step.mean())
for mode in (0, 1, 2):
    ms = timeit(mode)
    res["fwd_ms_mode%d" % mode] = ms
    print("forward (prologue + recurrence + logits) mode %d: %.3f ms" % (mode, ms))
print("status:", L.persistent_status())
print(json.dumps(res))
