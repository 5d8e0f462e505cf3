#!/usr/bin/env python3
"""Experiment: is the latency-bound recurrence faster as TWO independent half-batches on two HIP streams?
Runs forward + xe_loss + backward (the single-stream API calls) for N = 640 on one stream, and for two N = 320 halves
on two streams concurrently; same weights.  Prints ms per variant."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer

c = bench.CFG
torch.manual_seed(1234)
tr = Trainer(bench.make_opt("bf16", 1234)); tr.build_optimizer()
batch = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
model = tr.i2t_model
eng = model.engine
lib = eng.lib
t_run = model._steps_to_run(batch["labels"])
pd = {k: v.detach() for k, v in model.param_dict().items()}
T = batch["labels"].shape[1] - 1
R = batch["att_feats"].shape[1]
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2


def make(rows):
    sub = {k: v[rows].contiguous() for k, v in batch.items()}
    n = sub["labels"].shape[0]
    d = eng.dims(n, R, T)
    ws = eng.checkout(d, sub["fc_feats"].device)
    b = eng.batch_struct(sub["fc_feats"], sub["att_feats"], sub["att_masks"], sub["labels"], sub["masks"])
    grads = {k: torch.empty_like(v) for k, v in pd.items()}
    g = eng.weights_struct(grads)
    out = torch.empty(2, dtype=torch.float32, device="cuda")
    return dict(d=d, ws=ws, b=b, g=g, grads=grads, out=out, sub=sub)


full = make(slice(0, 640))
step = 640 // parts
halves = [make(slice(i * step, (i + 1) * step)) for i in range(parts)]
w = eng.refresh(pd, full["d"])
torch.cuda.synchronize()
der = L.ptr(eng._derived)


def run(x, s, what):
    if "f" in what:
        L.check(lib.uic_topdown_forward(C.byref(x["d"]), C.byref(w), der, C.byref(x["b"]), t_run, 1, 5, L.ptr(x["ws"].buf), None, s), "fwd")
    if "l" in what:
        L.check(lib.uic_topdown_xe_loss(C.byref(x["d"]), C.byref(x["b"]), t_run, L.ptr(x["ws"].buf), None, x["out"].data_ptr(),
                                        x["out"].data_ptr() + 4, s), "xe")
    if "b" in what:
        L.check(lib.uic_topdown_backward(C.byref(x["d"]), C.byref(w), der, C.byref(x["b"]), t_run, 1, 5, L.ptr(x["ws"].buf), None, None,
                                         C.byref(x["g"]), s), "bwd")


streams = [torch.cuda.Stream() for _ in range(parts)]
main = torch.cuda.current_stream()


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def one(what):
    run(full, main.cuda_stream, what)


def split(what):
    for st in streams:
        st.wait_stream(main)
    for x, st in zip(halves, streams):
        run(x, st.cuda_stream, what)
    for st in streams:
        main.wait_stream(st)


def split_interleaved(what):
    # enqueue phase by phase so neither stream's host enqueue starves the other
    for st in streams:
        st.wait_stream(main)
    for ph in what:
        for x, st in zip(halves, streams):
            run(x, st.cuda_stream, ph)
    for st in streams:
        main.wait_stream(st)


for what in ("f", "flb"):
    run(full, main.cuda_stream, "fl"); [run(x, main.cuda_stream, "fl") for x in halves]
    print("%-4s N=640 one stream          : %.3f ms" % (what, timeit(lambda: one(what))))
    print("%-4s %d x N=%d, %d streams       : %.3f ms" % (what, parts, step, parts, timeit(lambda: split(what))))
    print("%-4s %d x N=%d, phase-interleaved: %.3f ms" % (what, parts, step, timeit(lambda: split_interleaved(what))))
    print("%-4s 1 x N=%d alone            : %.3f ms" % (what, step, timeit(lambda: run(halves[0], main.cuda_stream, what))))
