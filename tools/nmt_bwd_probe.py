#!/usr/bin/env python3
"""Per-phase time stamps of the persistent decoder-BPTT launch of the pivot NMT step (csrc/nmt_persist.hip nmt_dec_bwd_kernel) at
BASELINE configs[2] shapes (batch 64, 2 layers, 512, target length 32): UIC_REC_STAMPS makes workgroup thread 0 write
s_memrealtime at every phase boundary; this prints the mean / median / max over workgroups per phase and the step time.
    gpurun -- python tools/nmt_bwd_probe.py"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--vocab", type=int, default=50004); ap.add_argument("--maxlen", type=int, default=30)
a = ap.parse_args()
opt = argparse.Namespace(layers=2, rnn_size=512, word_vec_size=512, brnn=True, rnn_type="LSTM", dropout=0.3, input_feed=1,
                         position_encoding=False, coverage_attn=False, copy_attn=False, context_gate=None, attention_type="dot",
                         attn_transform="softmax", fertility=None, predict_fertility=False, guided_fertility=None,
                         supervised_fertility=None, lambda_coverage=0, lambda_fertility=0, lambda_exhaust=0, batch_size=a.batch,
                         compute_dtype="bf16", seed=1, nmt_train_flag=1, i2t_train_flag=0, nmt_learning_rate=1e-3,
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
eng = tr.nmt_model.engine
eng.recurrence = L.REC_STAMPS
for _ in range(4):
    tr.train_nmt(batch)
torch.cuda.synchronize()
d = eng.dims(B, S, T)
(ws,) = eng._pool[(d.B, d.S, d.T, d.dtype)]
lib = L.load()
Td = T - 1
n = 256 * Td * 16
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def stamps(name):
    p = lib.uic_nmt_workspace_ptr(C.byref(d), C.c_void_p(ws.data_ptr()), name)
    assert p, "no %s in the workspace" % name
    host = torch.empty(n, dtype=torch.int64)
    assert hip.hipMemcpy(C.c_void_p(host.data_ptr()), C.c_void_p(p), n * 8, 2) == 0
    return host.view(256, Td, 16).double() * 10e-3          # 100 MHz ticks -> us


def report(title, st, names, first, lastt):
    print(title)
    mid = st[:, 1:Td - 1]                                 # (first / last steps: cold loads, no carry)
    tot = 0.0
    for i, nm in enumerate(names):
        ph = mid[:, :, i + 1] - mid[:, :, i]
        tot += ph.mean().item()
        print("   %-28s mean %6.2f us   median %6.2f   max over WGs (mean over t) %6.2f   min over WGs %6.2f" % (
            nm, ph.mean().item(), ph.median().item(), ph.mean(1).max().item(), ph.mean(1).min().item()))
    k = len(names)
    step = (mid[:, :, k] - mid[:, :, 0]).mean().item()
    print("   step   mean %6.2f us (sum of phases %.2f); launch: first stamp -> last stamp %.1f us for %d steps" % (
        step, tot, (st[:, lastt, k].max() - st[:, first, 0].min()).item(), Td))


report("decoder forward (nmt_dec_ws_kernel)", stamps(b"dec_fwd_dbg"),
       ["layer 0 cell", "barrier 1", "layer 1 cell", "barrier 2", "attention", "barrier 3", "linear_out + tanh", "barrier 4"], 0, Td - 1)
report("decoder BPTT (nmt_dec_bwd_kernel)", stamps(b"dec_bwd_dbg"),
       ["A: d_pre, d[c;q] GEMM", "barrier 1", "B: attention backward", "barrier 2", "C1: top cell backward", "barrier 3",
        "C2 + D1: d x GEMM, cell 0", "barrier 4", "D2: d feed GEMM", "barrier 5"], Td - 1, 0)
