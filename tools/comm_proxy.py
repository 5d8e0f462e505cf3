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
args = ap.parse_args()
lib = L.load()


class ProxyExchange(GradientExchange):
    """Two 'ranks' as far as the Trainer can tell (it then takes the overlapped four-piece exchange), one GPU in fact."""
    def __init__(self, workgroups, pieces=True):
        GradientExchange.__init__(self, None)
        self.wg, self.pieces, self.scratch, self.moved = workgroups, pieces, None, 0

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
        L.check(lib.uic_comm_proxy(t.data_ptr(), L.ptr(self.scratch), nbytes - nbytes % 16, self.wg, torch.cuda.current_stream().cuda_stream))
        self.moved += nbytes

    def allreduce_sum_overlapped(self, flat, splits, wait_group):
        if not self.pieces:
            return self.allreduce_sum(flat)
        return GradientExchange.allreduce_sum_overlapped(self, flat, splits, wait_group)


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
    t_run = tr.i2t_model._steps_to_run(batch["labels"])
    for _ in range(5):
        tr.train_device_batch(batch, t_run, den)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_device_batch(batch, t_run, den)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


print("comm stand-in: %d workgroups per collective; %d timed steps; 640 caption rows, bf16" % (args.workgroups, args.steps))
print("%-22s %12s %22s %22s" % ("gradient order", "no exchange", "4 overlapped pieces", "1 piece after the step"))
for name, early in (("default", False), ("early_grads", True)):
    base = run(early, None)
    ex4 = ProxyExchange(args.workgroups, True)
    t4 = run(early, ex4)
    ex1 = ProxyExchange(args.workgroups, False)
    t1 = run(early, ex1)
    print("%-22s %9.3f ms %13.3f ms (+%.3f) %13.3f ms (+%.3f)   [%.1f MB per step out and back]" % (
        name, base, t4, t4 - base, t1, t1 - base, ex4.moved / (args.steps + 5) / 1e6))
