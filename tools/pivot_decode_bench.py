#!/usr/bin/env python3
"""BASELINE config 3 -- the eval_pivot.py surface: image -> (im2zh TopDown captioner, beam search) -> pivot caption ->
(zh->en 2-layer LSTM NMT, NMTModel.translateBatch: beam 15, <= 100 steps) -> target caption, batch 64, bf16, random-init
weights of the real shapes (36 x 2048 features, hidden 512, caption vocabulary 9 487, NMT vocabularies 50 004).
Random weights rarely emit EOS, so both searches run their full length (16 caption steps; the translator's step count is
set by --nmt-steps, default 30, a typical sentence length; the reference's cap is 100)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import bench
from unpaired_image_captioning_amd import models
from unpaired_image_captioning_amd.models import NMT_Models
from unpaired_image_captioning_amd.synthetic import synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--beam", type=int, default=3)
ap.add_argument("--nmt-steps", type=int, default=30); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
c = bench.CFG
torch.manual_seed(1)
cap = models.setup(bench.make_opt("bf16", 1)).cuda().eval()
V = 50004
opt = argparse.Namespace(layers=2, rnn_size=512, word_vec_size=512, brnn=True, rnn_type="LSTM", dropout=0.3, input_feed=1,
                         position_encoding=False, coverage_attn=False, copy_attn=False, context_gate=None, attention_type="dot",
                         attn_transform="softmax", fertility=None, predict_fertility=False, guided_fertility=None,
                         supervised_fertility=None, lambda_coverage=0, lambda_fertility=0, lambda_exhaust=0, batch_size=a.batch,
                         gpus=[0], compute_dtype="bf16", seed=1)
nmt = NMT_Models.NMTModel(opt, NMT_Models.Encoder(opt, V), NMT_Models.Decoder(opt, V), None, None, False)
nmt.generator = nn.Sequential(nn.Linear(512, V), nn.LogSoftmax(dim=-1))
nmt.cuda().eval()
b = synthetic_batch(a.batch, 1, c["R"], c["D"], c["V"], c["L"], seed=5)


def caption():
    with torch.no_grad():
        seq, _ = cap(b["fc_feats"], None, b["att_feats"], b["att_masks"], opt={"beam_size": a.beam}, mode="sample")
    return seq


def translate(seq):
    # pivot caption tokens -> NMT source ids (the reference maps words through the two dictionaries; ids + 4 skips the
    # NMT special tokens), time-major [S, B, 1], zero = PAD after the caption's end
    src = torch.where(seq > 0, seq + 4, torch.zeros_like(seq)).t().contiguous().unsqueeze(2)
    src[0] = torch.where(src[0] == 0, torch.full_like(src[0], 5), src[0])      # no empty source sentence
    return nmt.translateBatch(argparse.Namespace(src=src, batchSize=a.batch), max_steps=a.nmt_steps)


def timeit(fn, iters):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, r


t_cap, seq = timeit(caption, a.iters)
t_tr, _ = timeit(lambda: translate(seq), a.iters)
t_all, _ = timeit(lambda: translate(caption()), a.iters)
print("pivot decode, batch %d: captioner beam-%d %.2f ms, translateBatch (beam 15, %d steps) %.2f ms, joint %.2f ms = %.0f images/s"
      % (a.batch, a.beam, t_cap, a.nmt_steps, t_tr, t_all, a.batch / t_all * 1e3))
