#!/usr/bin/env python3
"""Isolated phase timings of the captioner step at the bench shapes (HIP events): forward API call (prologue + recurrence +
all logits, one stream), XE, backward API call (one stream), fused two-stream step, Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer, xe_step

c = bench.CFG
torch.manual_seed(1234)
tr = Trainer(bench.make_opt("bf16", 1234)); tr.build_optimizer()
batch = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
model = tr.i2t_model
model.defer_status_check = True          # timing loops: no host sync inside the decode calls
eng = model.engine
t_run = model._steps_to_run(batch["labels"])
pd = {k: v.detach() for k, v in model.param_dict().items()}
grads = tr.arena.grad_views


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


state = {}
def fwd():
    _, ws, (d, w, b) = eng.forward(pd, batch["fc_feats"], batch["att_feats"], batch["att_masks"], batch["labels"], t_run, True, 5,
                                   want_logprobs=False, masks=batch["masks"])
    state["x"] = (ws, d, w, b)
def fwd_rel():
    fwd(); eng.release(state["x"][0])
def loss():
    ws, d, w, b = state["x"]; eng.xe_loss(ws, d, b, t_run)
def bwd():
    ws, d, w, b = state["x"]; eng.backward(ws, d, w, b, t_run, True, 5, grads)
print("forward API (refresh + prologue + 17 steps + logits): %.3f ms" % timeit(fwd_rel))
fwd()
print("xe_loss: %.3f ms" % timeit(loss))
print("backward API (single stream): %.3f ms" % timeit(bwd))
eng.release(state["x"][0])
print("fused two-stream step (xe_step): %.3f ms" % timeit(lambda: xe_step(model, batch, t_run=t_run, grads=grads)))
print("separate calls (xe_step fused=False): %.3f ms" % timeit(lambda: xe_step(model, batch, t_run=t_run, grads=grads, fused=False)))
den = float(batch["masks"][:, 1:c["L"] + 2].sum().item())
print("whole trainer step: %.3f ms" % timeit(lambda: tr.train_device_batch(batch, t_run, den)))

# ---- self-critical branch (P/trainer.py:166-171): multinomial sampling pass (train mode), greedy baseline (eval mode),
# teacher-forced replay + backward with the reward weights
fc, att, am = batch["fc_feats"], batch["att_feats"], batch["att_masks"]
def sample_pass():
    model.train()
    with torch.no_grad():
        return model(fc, None, att, am, opt={'sample_max': 0}, mode='sample')
def greedy_pass():
    model.eval()
    with torch.no_grad():
        r = model(fc, None, att, am, opt={'sample_max': 1}, mode='sample')
    model.train()
    return r
def beam_pass():
    model.eval()
    with torch.no_grad():
        r = model(fc[::5], None, att[::5], am[::5], opt={'beam_size': 3}, mode='sample')
    model.train()
    return r
print("multinomial sampling pass, 640 rows x 16 steps: %.3f ms" % timeit(sample_pass))
print("greedy pass, 640 rows x 16 steps: %.3f ms" % timeit(greedy_pass))
print("beam-3 search, 128 images x 16 steps: %.3f ms" % timeit(beam_pass))
import numpy as np
data = {k: v.cpu().numpy() for k, v in batch.items()}
def scst():
    tr.train_self_critical(data, lambda d, s, g: np.ones(s.shape, dtype=np.float32))
print("Trainer.train_self_critical incl. H2D of the batch and the reward round trip: %.3f ms" % timeit(scst, iters=5, warm=2))

# the reference's own reward: CIDEr-D(sampled) - CIDEr-D(greedy) against 5 references per image, cached document frequencies
import pickle, tempfile
from unpaired_image_captioning_amd.misc import rewards
g = np.random.default_rng(1)
L_ = c["L"]
data["gts"] = [data["labels"][i * c["S"]:(i + 1) * c["S"], 1:L_ + 1].astype(np.int64) for i in range(c["n_img"])]
df = {}
for img in data["gts"]:
    grams = set()
    for r in img:
        w = rewards.DeviceCiderD._words(r)
        for k in range(1, 5):
            for i in range(len(w) - k + 1):
                grams.add(tuple(str(t) for t in w[i:i + k]))
    for ng in grams:
        df[ng] = df.get(ng, 0.0) + 1.0
pk = os.path.join(tempfile.gettempdir(), "uic_phase_times-idxs.p")
with open(pk, "wb") as f:
    pickle.dump({"document_frequency": df, "ref_len": float(c["n_img"])}, f)
tr.opt.cached_tokens = pk
rewards.CiderD_scorer = None
def scst_dev():
    tr.train_self_critical(data)
print("Trainer.train_self_critical with the device CIDEr-D reward (%d n-grams in the table), incl. H2D: %.3f ms" % (len(df), timeit(scst_dev, iters=5, warm=2)))
# decode passes of the step: the default (per-step launch chains, the greedy baseline on a second stream beside the sampling
# pass) against one persistent decode launch per pass (they cannot share the chip, so the two passes run one after the other)
from unpaired_image_captioning_amd import _lib as _L
tr.persistent_decode = True
print("  ... with persistent decode launches (sampling pass, then the greedy baseline): %.3f ms" % timeit(scst_dev, iters=5, warm=2))
tr.persistent_decode = False
st3 = {"cur": data, "nxt": dict(data)}
def scst_dev_prefetch():
    tr.train_self_critical(st3["cur"], next_data=st3["nxt"])
    st3["cur"], st3["nxt"] = st3["nxt"], st3["cur"]
print("  ... with the next batch shipped during the step (next_data=): %.3f ms" % timeit(scst_dev_prefetch, iters=5, warm=2))
hyp = torch.randint(0, c["V"], (2 * 640, L_), device="cuda")
sc = rewards.CiderD_scorer
print("CIDEr-D scores of 1280 captions x 5 references (kernel + reference upload): %.3f ms" % timeit(lambda: sc.scores(hyp, data["gts"], 640, c["S"])))

def xe_host():
    tr.train(data)
print("Trainer.train (XE) from host numpy incl. H2D (features once per image) and the loss.item() sync: %.3f ms" % timeit(xe_host, iters=10, warm=3))
data2 = dict(data)
state2 = {"cur": data, "nxt": data2}
def xe_host_prefetch():
    tr.train(state2["cur"], next_data=state2["nxt"])
    state2["cur"], state2["nxt"] = state2["nxt"], state2["cur"]
print("Trainer.train (XE) from host numpy with the NEXT batch shipped during the step (next_data=): %.3f ms" % timeit(xe_host_prefetch, iters=10, warm=3))
tr.opt.ship_replicated_features = 1
print("Trainer.train (XE) from host numpy, features replicated on the host as the reference ships them: %.3f ms" % timeit(xe_host, iters=5, warm=2))
