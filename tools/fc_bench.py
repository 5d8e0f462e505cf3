#!/usr/bin/env python3
"""BASELINE config 1 (the reference's own CPU-runnable case): FC caption model (FCModel_NMT + maxout LSTMCore) on 2048-d fc
features, 16 images x 5 captions = 80 rows, seq_len 16, hidden 512, V + 1 = 9 488.  One step = forward (model call, dense
log-probs like the reference) + LanguageModelCriterion + backward + Adam.  GPU (bf16 / f32) and, beside it, the CPU oracle
(oracle/fc.py, results-identical restatement of the reference) on this host."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unpaired_image_captioning_amd import models
from unpaired_image_captioning_amd.misc.criterion import LanguageModelCriterion

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=16); ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--cpu-threads", type=int, default=16)
a = ap.parse_args()
V, E, H, D, L, S = 9487, 512, 512, 2048, 16, 5
from oracle import fc as OF, topdown as O
b = O.synthetic_batch(a.images, S, 3, D, V, L, seed=3)
N = a.images * S
for dtype in ("bf16", "f32"):
    opt = argparse.Namespace(vocab_size=V, input_encoding_size=E, rnn_type="LSTM", rnn_size=H, num_layers=1, drop_prob_lm=0.5,
                             seq_length=L, fc_feat_size=D, caption_model="fc", compute_dtype=dtype)
    torch.manual_seed(1)
    m = models.setup(opt).cuda().train()
    optim = torch.optim.Adam(m.parameters(), lr=5e-4)
    fc, labels, masks = b["fc_feats"].cuda(), b["labels"].cuda(), b["masks"].cuda()
    crit = LanguageModelCriterion()

    def step():
        optim.zero_grad(set_to_none=True)
        loss = crit(m(fc, None, None, labels, None), labels[:, 1:], masks[:, 1:])
        loss.backward()
        optim.step()
        return loss
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print("fc %s: %.3f ms/step, %.0f captions/s (N=%d rows), loss %.3f" % (dtype, dt * 1e3, N / dt, N, loss.item()))

torch.set_num_threads(a.cpu_threads)
W = OF.init_weights(V + 1, E, H, D, seed=1)
Wp = {k: v.clone().requires_grad_(True) for k, v in W.items()}
opt_c = torch.optim.Adam(list(Wp.values()), lr=5e-4)
g = torch.Generator().manual_seed(0)


def cpu_step():
    opt_c.zero_grad(set_to_none=True)
    drop = dict(out=(torch.rand(L + 2, N, H, generator=g) >= 0.5).float() * 2.0)      # dropout 0.5 on next_h (FCModel_NMT.py:47-50)
    loss, grads, _ = OF.xe_loss_and_grads({k: v.detach() for k, v in Wp.items()}, b["fc_feats"], b["labels"], b["masks"], drop)
    for k, v in Wp.items():
        v.grad = grads[k]
    opt_c.step()
    return loss


cpu_step(); t0 = time.perf_counter()
for _ in range(5):
    cpu_step()
dt = (time.perf_counter() - t0) / 5
print("fc CPU oracle (%d threads): %.1f ms/step, %.0f captions/s" % (a.cpu_threads, dt * 1e3, N / dt))
