#!/usr/bin/env python3
"""Soak run: XE training steps, lone decode passes (persistent launches) and self-critical steps interleaved for a few minutes;
every persistent launch's status words are checked, losses must stay finite, greedy decoding must be repeatable.
    python tools/soak.py [--minutes 3]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--safe", action="store_true", help="the persistent kernels' placement-independent SAFE protocol (UIC_REC_SAFE)")
ap.add_argument("--recurrence", type=int, default=0, help="more UIC_REC_* bits for the engine (2: persistent BPTT, 16: early gradient order)")
args = ap.parse_args()

import numpy as np
import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer

c = CFG
tr = Trainer(make_opt("bf16", 1234))
tr.build_optimizer()
if args.safe:
    tr.i2t_model.engine.recurrence |= L.REC_SAFE
tr.i2t_model.engine.recurrence |= args.recurrence
batch = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=7)
dev = {k: v.cuda() for k, v in batch.items()}
data = {k: v.cpu().numpy() for k, v in batch.items()}
model = tr.i2t_model
t_end = time.time() + 60.0 * args.minutes
it = 0
n_xe = n_dec = n_sc = 0
st0 = L.persistent_status()
while time.time() < t_end:
    for _ in range(20):
        loss = tr.train(data)
        n_xe += 1
        assert np.isfinite(loss), loss
    model.eval()
    with torch.no_grad():
        g1, lp1 = model(dev["fc_feats"], None, dev["att_feats"], dev.get("att_masks"), opt={"sample_max": 1}, mode="sample")
        g2, lp2 = model(dev["fc_feats"], None, dev["att_feats"], dev.get("att_masks"), opt={"sample_max": 1}, mode="sample")
        s1, _ = model(dev["fc_feats"], None, dev["att_feats"], dev.get("att_masks"), opt={"sample_max": 0}, mode="sample")
    model.train()
    n_dec += 3
    assert torch.equal(g1, g2) and torch.equal(lp1, lp2), "greedy decoding is not repeatable"
    assert int(s1.max()) <= c["V"] and int(s1.min()) >= 0
    for persistent in (False, True):
        tr.persistent_decode = persistent
        loss = tr.train_self_critical(data, lambda d, s, g: np.where(s[:, :1] % 2 == 0, 1.0, -1.0).repeat(s.shape[1], 1).astype(np.float32))
        n_sc += 1
        assert np.isfinite(loss), loss
    tr.persistent_decode = False
    it += 1
torch.cuda.synchronize()
st1 = L.persistent_status()
print("soak (recurrence flags %d) %.1f min: %d XE steps, %d decode passes, %d self-critical steps; persistent launches %d (XCD-local) + %d (SAFE), time-outs %d; last XE loss %.4f"
      % (tr.i2t_model.engine.recurrence, args.minutes, n_xe, n_dec, n_sc, st1[1] - st0[1], st1[2] - st0[2], st1[0], tr.last_loss if tr.last_loss is not None else float("nan")))
assert st1[0] == 0
