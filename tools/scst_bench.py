#!/usr/bin/env python3
"""Trainer.train_self_critical at BASELINE configs[1] shapes (128 images x 5 sampled captions, device CIDEr-D reward), timed the
way the pivot NMT and XE steps are: many steps per variant, the variants alternating inside ONE process (boxes and hosts differ by
more than the effects), one host sync per step (the step's own loss.item()).
  resident    the batch already on the device (what bench.py's contract calls the step: inputs resident in HBM)
  host        the batch as host numpy arrays: + pinned staging and the PCIe transfer of 37.7 MB of features
  prefetch    host arrays, the NEXT batch shipped on the copy stream while this step computes (next_data=)
  persistent  resident, each decode pass as one persistent launch (sampling pass, then the greedy baseline)
    gpurun -- python tools/scst_bench.py [--steps 30] [--rounds 3]"""
import argparse, os, pickle, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer
from unpaired_image_captioning_amd.misc import rewards

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30); ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--only", default="")
ap.add_argument("--eos-bias", type=float, default=0.0, help="added to the end-of-caption logit's bias: 7 makes a caption sampled from the random "
                "weights end after ~9 words (P(end) ~ 0.1 per step), the length of a trained model's; 0 = captions run to full length")
a = ap.parse_args()
c = bench.CFG
torch.manual_seed(1234)
tr = Trainer(bench.make_opt("bf16", 1234)); tr.build_optimizer()
if a.eos_bias:
    import torch.nn as nn
    lg = tr.i2t_model.logit if isinstance(tr.i2t_model.logit, nn.Linear) else tr.i2t_model.logit[-1]
    with torch.no_grad():
        lg.bias[0] += a.eos_bias
    tr.lr = tr.i2t_current_lr = 0.0                       # (the bias is to stay where it is over the timed steps)
batch = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
host = {k: v.cpu().numpy() for k, v in batch.items()}
L_ = c["L"]
gts = [host["labels"][i * c["S"]:(i + 1) * c["S"], 1:L_ + 1].astype(np.int64) for i in range(c["n_img"])]
df = {}
for img in gts:
    grams = set()
    for r in img:
        w = rewards.DeviceCiderD._words(r)
        for k in range(1, 5):
            for i in range(len(w) - k + 1):
                grams.add(tuple(str(t) for t in w[i:i + k]))
    for ng in grams:
        df[ng] = df.get(ng, 0.0) + 1.0
pk = os.path.join(tempfile.gettempdir(), "uic_scst_bench-idxs.p")
with open(pk, "wb") as f:
    pickle.dump({"document_frequency": df, "ref_len": float(c["n_img"])}, f)
tr.opt.cached_tokens = pk
rewards.CiderD_scorer = None
host["gts"] = gts
# features once per image, as Trainer.to_device ships them (the reference replicates them per caption on the host)
per_img = dict(host)
for k in ("fc_feats", "att_feats", "att_masks"):
    if per_img.get(k) is not None and len(per_img[k]) == c["n_img"] * c["S"]:
        per_img[k] = per_img[k][::c["S"]]
resident = {k: (torch.from_numpy(np.ascontiguousarray(v)).cuda() if isinstance(v, np.ndarray) and k not in ("labels", "masks") else v) for k, v in per_img.items()}
hosts = [dict(per_img), dict(per_img)]


def step(kind, i):
    if kind == "resident":
        tr.persistent_decode = False
        return tr.train_self_critical(resident)
    if kind == "persistent":
        tr.persistent_decode = True
        try:
            return tr.train_self_critical(resident)
        finally:
            tr.persistent_decode = False
    if kind == "host":
        return tr.train_self_critical(hosts[i & 1])
    if kind == "prefetch":
        return tr.train_self_critical(hosts[i & 1], next_data=hosts[(i + 1) & 1])
    raise ValueError(kind)


kinds = [k for k in ("resident", "host", "prefetch", "persistent") if not a.only or k in a.only.split(",")]
res = {k: [] for k in kinds}
for k in kinds:
    for i in range(4):
        step(k, i)
torch.cuda.synchronize()
for r in range(a.rounds):
    for k in kinds:
        for i in range(3):
            step(k, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(k, i)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / a.steps * 1e3)
print("Trainer.train_self_critical, %d images x %d captions, bf16; ms per step (median of %d rounds of %d steps; all rounds)" % (c["n_img"], c["S"], a.rounds, a.steps))
for k in kinds:
    v = sorted(res[k])
    print("   %-11s %6.3f   %s" % (k, v[len(v) // 2], "  ".join("%.3f" % x for x in res[k])))
