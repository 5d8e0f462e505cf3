#!/usr/bin/env python3
"""Timing of the pivot NMT training step at BASELINE config 3 shapes: zh->en 2-layer LSTM NMT,
batch 64, rnn_size = word_vec_size = 512, vocabularies 50 004, source / target lengths ~U{5..30}, dropout 0.3, Adam with
clip_grad_norm 5.  One step = zero_grad, NMTModel.forward (+ generator + NLL + counters), backward, Optim.step, with
the batch resident in HBM.  Prints sentences/s and target tokens/s."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unpaired_image_captioning_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--vocab", type=int, default=50004); ap.add_argument("--maxlen", type=int, default=30)
ap.add_argument("--all-positions", action="store_true", help="without the list of non-PAD target positions (uic_nmt_dims.tgt_live_rows)")
ap.add_argument("--ab", type=int, default=0, help="alternate blocks with / without the list this many times and print both medians")
a = ap.parse_args()
opt = argparse.Namespace(layers=2, rnn_size=512, word_vec_size=512, brnn=True, rnn_type="LSTM", dropout=0.3, input_feed=1,
                         position_encoding=False, coverage_attn=False, copy_attn=False, context_gate=None, attention_type="dot",
                         attn_transform="softmax", fertility=None, predict_fertility=False, guided_fertility=None,
                         supervised_fertility=None, lambda_coverage=0, lambda_fertility=0, lambda_exhaust=0, batch_size=a.batch,
                         compute_dtype=a.dtype, seed=1, nmt_train_flag=1, i2t_train_flag=0, nmt_learning_rate=1e-3,
                         nmt_max_grad_norm=5, param_init=0.1)
tr = Trainer(opt)
tr.build_nmt(a.vocab, a.vocab)
g = torch.Generator().manual_seed(3)
B, S, T = a.batch, a.maxlen, a.maxlen + 2
lengths = torch.sort(torch.randint(5, S + 1, (B,), generator=g), descending=True)[0]; lengths[0] = S
src = torch.randint(4, a.vocab, (S, B), generator=g)
for b in range(B):
    src[lengths[b]:, b] = 0
tl = torch.randint(7, T + 1, (B,), generator=g); tl[0] = T
tgt = torch.randint(4, a.vocab, (T, B), generator=g); tgt[0] = 2
for b in range(B):
    tgt[tl[b] - 1, b] = 3; tgt[tl[b]:, b] = 0
batch = argparse.Namespace(src=src.unsqueeze(2).cuda(), tgt=tgt.cuda(), lengths=lengths.view(1, -1))
if not a.all_positions:                  # (what the Dataset attaches where it assembles the batch, onmt_dataset_h5.py)
    from unpaired_image_captioning_amd.models.NMT_Models import tgt_live_count
    batch.tgt.uic_live = (None, tgt_live_count(tgt))
ntok = int((tgt[1:] != 0).sum())
for _ in range(15):                      # (the first process on a fresh box needs more than a few steps to reach its pace)
    tr.train_nmt(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    loss = tr.train_nmt(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print("nmt %s: %.3f ms/step, %.0f sentences/s, %.0f target tokens/s (B=%d, S=%d, T=%d, V=%d, %d target tokens), loss %.1f" % (
    a.dtype, dt * 1e3, B / dt, ntok / dt, B, S, T, a.vocab, ntok, loss))
if a.ab:
    import copy
    plain = argparse.Namespace(src=batch.src, tgt=batch.tgt.clone(), lengths=batch.lengths)      # (a tensor without the attribute)
    res = {"all": [], "live": []}
    for r in range(a.ab + 1):
        for name, bt in (("all", plain), ("live", batch)):
            for _ in range(3):
                tr.train_nmt(bt)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.steps):
                tr.train_nmt(bt)
            torch.cuda.synchronize()
            if r:
                res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
    for name, v in res.items():
        v = sorted(v)
        print("%-5s median %.3f ms  min %.3f  max %.3f" % (name, v[len(v) // 2], v[0], v[-1]))
