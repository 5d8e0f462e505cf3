#!/usr/bin/env python3
"""What the data-parallel exchange costs the training step on ONE GPU, with stand-ins for RCCL (no multi-GPU node is available
to this build): every collective of parallel_exchange.GradientExchange is replaced by a few workgroups (default 16,
~0.38 GB/ms) that stream the bytes on the stream the collective would run on, at the points where the step releases it --
an all-reduce moves its bytes out and back (uic_comm_proxy, two passes), a reduce-scatter or an all-gather moves them once
(uic_comm_proxy_oneway).  The step is CU-time bound, so a co-resident communication kernel slows the BPTT chain and the side
GEMMs: this measures by how much, for

  * the SHARDED exchange (round 6, the default of a data-parallel Trainer): reduce-scatter of four gradient pieces, small
    all-reduce of the replicated tail, Adam on 1/world of the arena, all-gather of the bf16 weights beside the next prologue;
  * round 5's exchange (opt.allreduce_exchange): all-reduce in four overlapped pieces, Adam on everything.

    python3 tools/comm_proxy.py [--workgroups 16] [--steps 30] [--world 8]"""
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
ap.add_argument("--world", type=int, default=8, help="the world size the stand-in pretends (Adam runs on 1/world of the arena)")
ap.add_argument("--rec", default="0", help="extra uic_topdown_dims.recurrence bits (measurement knobs, csrc/uic_common.h)")
ap.add_argument("--no-next-den", action="store_true", help="do not carry the next batch's mask sum in the step's all-reduce (a 1-float "
                "collective then sits in front of every forward pass)")
ap.add_argument("--only", default="", help="sharded | allreduce | none: only that exchange / only the step without one (for traces)")
ap.add_argument("--f32", action="store_true")
ap.add_argument("--no-comm-flag", action="store_true", help="clear UIC_REC_COMM_STREAM: the single-GPU stream layout (chunk weight gradients on the "
                "third stream, 256 x 256 kernel) beside the exchange")
ap.add_argument("--early", action="store_true", help="opt.early_grads (UIC_REC_EARLY_GRADS): the embedding table and att_lstm.weight_ih final with the LSTM matrices")
ap.add_argument("--no-pipeline", action="store_true", help="opt.no_pipelined_logit_piece: the logit piece's Adam + all-gather with the others, after the step")
ap.add_argument("--half", action="store_true", help="opt.bf16_gradient_exchange: the pieces that are still on the wire when the step has joined travel as bf16")
ap.add_argument("--comm-stream", default="torch", choices=["torch", "high", "raw", "raw-early", "raw-low"],
                help="how the communication stream is made: torch = torch.cuda.Stream() (the Trainer's default); high = priority -1; raw = "
                     "hipStreamCreateWithFlags(non-blocking) wrapped as an ExternalStream, created after the library's side streams; "
                     "raw-early = the same, created before anything else touches the GPU; raw-low = lowest priority")
args = ap.parse_args()
lib = L.load()


class ProxyExchange(GradientExchange):
    """`world` ranks as far as the Trainer can tell, one GPU in fact; every collective is a stand-in that moves the same bytes."""
    def __init__(self, workgroups, world):
        GradientExchange.__init__(self, None)
        self.wg, self._world, self.scratch, self.moved = workgroups, world, None, 0
        self.stamps = None          # [(kind, start event, end event, bytes)] of the current step's collectives when not None

    world_size = property(lambda self: self._world)
    rank = property(lambda self: 0)

    def ranks_share_a_device(self):
        return False

    def _run(self, kind, t, passes):
        nbytes = t.numel() * t.element_size()
        if nbytes < 4096 and kind == "ar":
            nbytes = 4096                              # the small sums: a latency, not a bandwidth -- one tiny launch stands for it
        if self.scratch is None or self.scratch.numel() < nbytes + 256:
            self.scratch = torch.empty(max(nbytes + 256, 1 << 20), dtype=torch.uint8, device=t.device)
        if self.stamps is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        s = torch.cuda.current_stream().cuda_stream
        if passes == 2 and t.numel() * t.element_size() >= 4096:
            L.check(lib.uic_comm_proxy(t.data_ptr(), L.ptr(self.scratch), nbytes - nbytes % 16, self.wg, s))
        else:
            src = t.data_ptr() if t.numel() * t.element_size() >= 4096 else self.scratch.data_ptr() + (1 << 19)
            L.check(lib.uic_comm_proxy_oneway(src, L.ptr(self.scratch), nbytes - nbytes % 16, min(self.wg, max(1, nbytes // 4096)), s))
        if self.stamps is not None:
            b.record()
            self.stamps.append((kind, a, b, nbytes))
        self.moved += nbytes * passes

    def _sum(self, t):
        self._run("ar", t, 2)

    def _reduce_scatter(self, whole, mine):
        self._run("rs", whole, 1)

    def _all_gather(self, whole, mine):
        self._run("ag", whole, 1)


c = bench.CFG
batch_cpu = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
batch = Trainer.attach_live({k: v.cuda() for k, v in batch_cpu.items()})   # (as Trainer.to_device does for every real batch)
T = batch["labels"].shape[1] - 1
den = float(batch["masks"][:, 1:T + 1].sum().item())
NAMES = ["start", "prologue", "recurrence", "logit layer", "BPTT starts", "BPTT done", "rec wgrads", "main tail", "side tail", "joined", "logit grads"]


_raw_streams = []


def make_comm_stream(kind):
    if kind == "torch":
        return None
    if kind == "high":
        return torch.cuda.Stream(priority=-1)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    h = C.c_void_p()
    if kind == "raw-low":
        lo, hi = C.c_int(), C.c_int()
        hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
        rc = hip.hipStreamCreateWithPriority(C.byref(h), 1, lo.value)
    else:
        rc = hip.hipStreamCreateWithFlags(C.byref(h), 1)
    assert rc == 0, rc
    _raw_streams.append(h)
    return torch.cuda.ExternalStream(h.value)


_early = make_comm_stream("raw") if args.comm_stream == "raw-early" else None


def run(exchange, allreduce):
    opt = bench.make_opt("f32" if args.f32 else "bf16", 1234)
    opt.allreduce_exchange = int(allreduce)
    opt.allow_many_hw_queues = 1
    opt.early_grads = int(args.early)
    opt.no_pipelined_logit_piece = int(args.no_pipeline)
    opt.bf16_gradient_exchange = int(args.half)
    tr = Trainer(opt, exchange=exchange) if exchange is not None else Trainer(opt)
    tr.build_optimizer()
    tr.i2t_model.engine.recurrence |= int(args.rec, 0)
    if args.no_comm_flag:
        tr.i2t_model.engine.recurrence &= ~L.REC_COMM_STREAM
    if exchange is not None and args.comm_stream != "torch":
        if args.comm_stream != "raw-early":
            tr.train_device_batch(batch, tr.i2t_model._steps_to_run(batch["labels"]), den, None) if False else None
        tr._comm_stream = _early if args.comm_stream == "raw-early" else make_comm_stream(args.comm_stream)
        exchange._comm_stream = tr._comm_stream
    t_run = tr.i2t_model._steps_to_run(batch["labels"])
    nd = None if (args.no_next_den or exchange is None) else den
    for _ in range(10):
        tr.train_device_batch(batch, t_run, den, nd)
    torch.cuda.synchronize()
    # three timed rounds, the median reported: the first round of a process's FIRST Trainer can carry 0.3-1.5 ms per step of one-off
    # work that five warm-up steps did not cover (seen on the slower boxes of the pool: 3.2-4.4 ms for a 2.88 ms step)
    rounds = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.train_device_batch(batch, t_run, den, nd)
        torch.cuda.synchronize()
        rounds.append((time.perf_counter() - t0) / args.steps * 1e3)
    ms = sorted(rounds)[1]
    print("      rounds of %d steps: %s ms" % (args.steps, "  ".join("%.3f" % r for r in rounds)))
    if exchange is not None:
        # when the collectives run, un-traced: events around every one of three more steps, relative to the step's start
        import ctypes as C
        L.check(lib.uic_topdown_step_marks(1, None))
        for _ in range(3):
            exchange.stamps = []
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            tr.train_device_batch(batch, t_run, den, nd)
            e1.record()
            torch.cuda.synchronize()
            ms_ = (C.c_float * L.STEP_MARKS)()
            L.check(lib.uic_topdown_step_marks(1, ms_))
            print("      marks: " + "  ".join("%s %.3f" % (n, v) for n, v in zip(NAMES[1:], list(ms_)[1:])))
            print("      enqueue-to-enqueue %.3f ms; collectives (kind MB: start -> end ms after this step's first launch; "
                  "negative = the previous step's all-gather tail): " % e0.elapsed_time(e1) +
                  "  ".join("%s %.1f: %.3f -> %.3f" % (k, n / 1e6, e0.elapsed_time(a), e0.elapsed_time(b)) for k, a, b, n in exchange.stamps))
        exchange.stamps = None
        L.check(lib.uic_topdown_step_marks(0, None))
    return ms


print("comm stand-in: %d workgroups per collective; %d timed steps; 640 caption rows, %s; pretended world size %d; GPU_MAX_HW_QUEUES=%s" % (
    args.workgroups, args.steps, "f32" if args.f32 else "bf16", args.world, os.environ.get("GPU_MAX_HW_QUEUES")))
base = None
if args.only in ("", "none"):
    base = run(None, False)
    print("no exchange (single-GPU step)                          %9.3f ms" % base)
if args.only in ("", "sharded"):
    ex = ProxyExchange(args.workgroups, args.world)
    t = run(ex, False)
    print("SHARDED: 4 x reduce-scatter + small all-reduce + Adam/%d + 4 x all-gather   %9.3f ms%s   [%.1f MB moved per step]" % (
        args.world, t, " (+%.3f)" % (t - base) if base else "", ex.moved / (3 * args.steps + 13) / 1e6))
if args.only in ("", "allreduce"):
    ex = ProxyExchange(args.workgroups, args.world)
    t = run(ex, True)
    print("round 5: all-reduce in 4 overlapped pieces + Adam on everything          %9.3f ms%s   [%.1f MB moved per step]" % (
        t, " (+%.3f)" % (t - base) if base else "", ex.moved / (3 * args.steps + 13) / 1e6))
