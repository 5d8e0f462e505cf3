#!/usr/bin/env python3
"""What the overlapped gradient exchange costs the training step on ONE GPU, with a stand-in for RCCL (no multi-GPU node is
available to this build): every all-reduce of parallel_exchange.GradientExchange is replaced by uic_comm_proxy -- a few
workgroups (default 16, ~0.38 GB/ms) that stream the piece out and back on the stream the collective would run on, at the
points where uic_topdown_grad_ready_wait releases it.  The step is CU-time bound, so a co-resident comm kernel slows the BPTT
chain and the side GEMMs: this measures by how much, for the default gradient order and for opt.early_grads.

    python3 tools/comm_proxy.py [--workgroups 16] [--steps 30]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd.parallel_exchange import GradientExchange
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--workgroups", type=int, default=16)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--rec", default="0", help="extra uic_topdown_dims.recurrence bits (measurement knobs, csrc/uic_common.h)")
ap.add_argument("--overlap", type=int, default=3, help="how many leading pieces go to the communication stream (rest: after the step)")
ap.add_argument("--poll", action="store_true", help="uic_topdown_grad_ready_wait(group | UIC_GRAD_WAIT_POLL): polling kernels instead of barrier packets")
ap.add_argument("--skip", type=int, default=0, help="create (and use once) this many throw-away streams before the communication stream exists")
ap.add_argument("--no-comm-flag", action="store_true", help="clear UIC_REC_COMM_STREAM: the single-GPU stream layout beside the exchange")
ap.add_argument("--only", default="", help="default4: only the default order with the four overlapped pieces (for traces)")
args = ap.parse_args()
lib = L.load()


class ProxyExchange(GradientExchange):
    """Two 'ranks' as far as the Trainer can tell (it then takes the overlapped four-piece exchange), one GPU in fact."""
    def __init__(self, workgroups, pieces=True):
        GradientExchange.__init__(self, None)
        self.wg, self.pieces, self.scratch, self.moved = workgroups, pieces, None, 0
        self.stamps = None          # [(start event, end event, bytes)] of the current step's pieces when not None

    world_size = property(lambda self: 2)
    rank = property(lambda self: 0)

    def ranks_share_a_device(self):
        return False

    def _sum(self, t):
        nbytes = t.numel() * t.element_size()
        if nbytes < 4096:
            return                                     # the 1-float / 2-float sums: latency, not bandwidth
        if self.scratch is None or self.scratch.numel() < nbytes:
            self.scratch = torch.empty(nbytes + 256, dtype=torch.uint8, device=t.device)
        if self.stamps is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        L.check(lib.uic_comm_proxy(t.data_ptr(), L.ptr(self.scratch), nbytes - nbytes % 16, self.wg, torch.cuda.current_stream().cuda_stream))
        if self.stamps is not None:
            b.record()
            self.stamps.append((a, b, nbytes))
        self.moved += nbytes

    def allreduce_sum_overlapped(self, flat, splits, wait_group):
        if not self.pieces:
            return self.allreduce_sum(flat)
        if args.poll:
            wait_group = lambda raw, g: L.check(lib.uic_topdown_grad_ready_wait(raw, g | 0x100), "grad_ready_wait")
        return GradientExchange.allreduce_sum_overlapped(self, flat, list(splits)[:args.overlap], wait_group)


c = bench.CFG
batch_cpu = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
batch = {k: v.cuda() for k, v in batch_cpu.items()}
T = batch["labels"].shape[1] - 1
den = float(batch["masks"][:, 1:T + 1].sum().item())


def run(early, exchange):
    opt = bench.make_opt("bf16", 1234)
    opt.early_grads = int(early)
    tr = Trainer(opt, exchange=exchange) if exchange is not None else Trainer(opt)
    tr.build_optimizer()
    tr.i2t_model.engine.recurrence |= int(args.rec, 0)
    if args.no_comm_flag:
        tr.i2t_model.engine.recurrence &= ~L.REC_COMM_STREAM
    t_run = tr.i2t_model._steps_to_run(batch["labels"])
    for _ in range(5):
        tr.train_device_batch(batch, t_run, den)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_device_batch(batch, t_run, den)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    if exchange is not None and getattr(exchange, "pieces", False):
        # when the pieces run, un-traced: events around every piece of three more steps, relative to the step's start
        import ctypes as C
        names = ["start", "prologue", "recurrence", "logit layer", "BPTT starts", "BPTT done", "rec wgrads", "main tail", "side tail", "joined", "logit grads"]
        L.check(lib.uic_topdown_step_marks(1, None))
        for _ in range(3):
            exchange.stamps = []
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            tr.train_device_batch(batch, t_run, den)
            e1.record()
            torch.cuda.synchronize()
            ms_ = (C.c_float * L.STEP_MARKS)()
            L.check(lib.uic_topdown_step_marks(1, ms_))
            print("      marks: " + "  ".join("%s %.3f" % (n, v) for n, v in zip(names[1:], list(ms_)[1:])))
            print("      step %.3f ms; pieces (MB: start -> end ms): " % e0.elapsed_time(e1) +
                  "  ".join("%.1f: %.3f -> %.3f" % (n / 1e6, e0.elapsed_time(a), e0.elapsed_time(b)) for a, b, n in exchange.stamps))
        exchange.stamps = None
        L.check(lib.uic_topdown_step_marks(0, None))
    return ms


_dummies = []
for _ in range(args.skip):
    st_ = torch.cuda.Stream()
    with torch.cuda.stream(st_):
        torch.zeros(16, device="cuda").add_(1)
    _dummies.append(st_)
torch.cuda.synchronize()
print("comm stand-in: %d workgroups per collective; %d timed steps; 640 caption rows, bf16" % (args.workgroups, args.steps))
print("%-22s %12s %22s %22s" % ("gradient order", "no exchange", "4 overlapped pieces", "1 piece after the step"))
if args.only == "default4":
    print("default order, 4 overlapped pieces: %.3f ms" % run(False, ProxyExchange(args.workgroups, True)))
    sys.exit(0)
for name, early in (("default", False), ("early_grads", True)):
    base = run(early, None)
    ex4 = ProxyExchange(args.workgroups, True)
    t4 = run(early, ex4)
    ex1 = ProxyExchange(args.workgroups, False)
    t1 = run(early, ex1)
    print("%-22s %9.3f ms %13.3f ms (+%.3f) %13.3f ms (+%.3f)   [%.1f MB per step out and back]" % (
        name, base, t4, t4 - base, t1, t1 - base, ex4.moved / (args.steps + 5) / 1e6))
