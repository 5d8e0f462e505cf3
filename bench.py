#!/usr/bin/env python3
"""Benchmark of the hot path: train captions/sec of the TopDown attention-LSTM captioner
(BASELINE.json configs[1]: 36x2048 region features, hidden 512, 128 images x 5 captions = 640
caption rows per GPU, bf16 operands), synthetic data generated on the device.

One "step" = Trainer.train's work with inputs already resident in HBM: operand casts, feature
projection, 17-step teacher-forced unroll, logit GEMM + log-softmax + LanguageModelCriterion,
full BPTT, [N > 1: the sharded RCCL exchange -- reduce-scatter of the gradient pieces, Adam on the
rank's slice, all-gather of the bf16 weights], Adam, weight-copy refresh.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the persistent
recurrence kernel; `roofline_attention_step` = the stand-alone attention kernel, HBM-bound; `roofline_mfma` = the large
GEMM) and, at N=1, `cpu_baseline` (the CPU oracle timed on this host's cores).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Data-parallel runs: the step's streams, the communication stream and torch.distributed's own RCCL stream are more HIP streams
# than the four hardware queues ROCclr opens per priority by default, and on MI355X a process with MORE THAN FOUR busy
# hardware queues dispatches every dependent launch 1.5-2.5x slower (profiles/r05_*_queue_probe.txt: the step beside two
# one-wave kernels on two more streams, 3.08 -> 3.82 ms with the default, 3.20 with GPU_MAX_HW_QUEUES=2).  Read by the HIP
# runtime when it loads: set before torch is imported.  Single-GPU runs are left alone (no effect there: 3.076 vs 3.070 ms).
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CFG = dict(V=9487, E=512, H=512, A=512, D=2048, L=16, R=36, n_img=128, S=5)


def make_opt(dtype, seed, use_bn=0, att_feat_size=None):
    c = CFG
    return argparse.Namespace(vocab_size=c["V"], input_encoding_size=c["E"], rnn_size=c["H"], num_layers=1,
                              drop_prob_lm=0.5, seq_length=c["L"], fc_feat_size=c["D"], att_feat_size=att_feat_size or c["D"],
                              att_hid_size=c["A"], use_bn=use_bn, logit_layers=1, caption_model="topdown",
                              compute_dtype=dtype, seed=seed, i2t_learning_rate=5e-4, i2t_train_flag=1, seq_per_img=c["S"])


def _pmc_traffic(dtype):
    """HBM bytes per launch of the attention kernel from the committed PMC passes (tools/pmc_traffic.py), or None."""
    path = os.path.join(ROOT, "profiles", "attn_fwd_pmc_%s.json" % dtype)
    try:
        with open(path) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def attention_roofline(dtype_id, dtype_name, iters=64, pool=8):
    """Time the attention-step forward kernel alone, one HIP event pair per launch on the launch stream, at the bench
    shapes; algorithmic bytes per launch = N * (R*A + R*H + 2H + R) * sizeof.
    `achieved` / `frac` are HBM-COLD: the launches rotate through `pool` p_att/att pairs (377 MB in bf16, more than the
    256 MiB Infinity Cache), so every byte comes from HBM.  `us_per_launch_cache_resident` / `frac_cache_resident` re-read
    ONE pair (47 MB, Infinity-Cache resident) the way the 17 decode steps of a training step do -- not an HBM figure."""
    from unpaired_image_captioning_amd import _lib as L
    lib = L.load()
    c = CFG
    N, R, A, H = c["n_img"] * c["S"], c["R"], c["A"], c["H"]
    td = L.TORCH_DTYPE[dtype_id]
    g = torch.Generator(device="cuda").manual_seed(1)
    att_h = torch.randn(N, A, device="cuda", generator=g)
    p_atts = [torch.randn(N, R, A, device="cuda", generator=g).to(td) for _ in range(pool)]
    atts = [torch.randn(N, R, H, device="cuda", generator=g).abs().to(td) for _ in range(pool)]
    w = torch.randn(A, device="cuda", generator=g) * 0.05
    b = torch.zeros(1, device="cuda")
    alpha = torch.empty(N, R, device="cuda")
    ctx = torch.empty(N, H, device="cuda", dtype=td)

    def launch(k):
        L.check(lib.uic_attention_fwd(dtype_id, N, R, A, H, L.ptr(att_h), L.ptr(p_atts[k]), L.ptr(atts[k]), L.ptr(w), L.ptr(b),
                                      None, L.ptr(alpha), L.ptr(ctx), L.stream()))

    def timed(rotate):
        """Mean duration of ONE launch: an event pair around every launch (not a back-to-back average, whose launches overlap
        each other's ramps), on the stream the kernel is launched on."""
        for i in range(8):
            launch(i % pool if rotate else 0)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        # The host needs ~15 us to enqueue one (event, launch, event) triple through ctypes -- longer than the kernel runs -- so an
        # empty queue would let the GPU wait for the host INSIDE the event pair.  A ~2 ms spin kernel in front lets the host run
        # ahead; every pair then measures the device only: marker, kernel dispatch, kernel, marker (tools/micro/attn_variants.hip
        # reads the same 13.1 us from a C++ host).
        torch.cuda._sleep(int(5e6))
        for i in range(iters):
            ev[i][0].record()
            launch(i % pool if rotate else 0)
            ev[i][1].record()
        torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) for a, b in ev)
        return sum(d[iters // 8: iters - iters // 8]) / (iters - 2 * (iters // 8)) / 1e3      # trimmed mean, seconds
    def back_to_back(rotate, n=96):        # (96 enqueues take the host ~1.4 ms: inside the 2 ms head start the spin kernel gives it)
        """Duration per launch of n launches between ONE event pair (rotating buffers): consecutive launches overlap each other's
        ramp-up and tail, so this is the kernel's throughput, not its latency -- reported beside `frac`, never as it."""
        a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(5e6))
        a.record()
        for i in range(n):
            launch(i % pool if rotate else 0)
        b2.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b2) / n / 1e3
    cold_s, res_s = timed(True), timed(False)
    b2b_s = back_to_back(True)
    es = p_atts[0].element_size()
    bytes_per_launch = N * (R * A + R * H + 2 * H + R) * es
    achieved = bytes_per_launch / cold_s / 1e9
    traffic = _pmc_traffic(dtype_name)
    return {"bound": "hbm", "kernel": "attn_fwd_fast_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "note": "HBM-cold: the launches rotate through %d p_att/att pairs (%d MB > the 256 MiB Infinity Cache), one HIP event pair per launch" % (pool, pool * 2 * N * R * H * es // 2 ** 20),
            "traffic": traffic,
            "traffic_source": "profiles/attn_fwd_pmc_%s.json (committed rocprofv3 PMC passes of this kernel, NOT measured in this run)" % dtype_name if traffic else None,
            "bytes_per_launch": bytes_per_launch, "us_per_launch": round(cold_s * 1e6, 2),
            "us_per_launch_cache_resident": round(res_s * 1e6, 2),
            "frac_cache_resident": round(bytes_per_launch / res_s / 1e9 / HBM_PEAK_GBS, 4),
            "us_per_launch_back_to_back": round(b2b_s * 1e6, 2),
            "frac_back_to_back": round(bytes_per_launch / b2b_s / 1e9 / HBM_PEAK_GBS, 4)}


MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense peaks (MI355X_MICROARCH.md)


def recurrence_roofline(tr, batch, t_run, den_local, dtype_name, extra_steps=6):
    """The dominant single launch of the step since round 2: the persistent recurrence kernel (csrc/rnn_persist.hip), ONE
    launch = all t_run decode steps of all caption rows (att_lstm, h2att, attention, lang_lstm; AttModel.py:129-154).
    Duration: HIP timing events the fused step itself records on the stream the kernel is launched on
    (uic_topdown_step_marks: mark[2] - mark[1]), over `extra_steps` UNTIMED extra training steps run after the timed region.
    Algorithmic bytes per launch = SURVEY.md 8(d)'s attention-step figure (R*A + R*H + 2H + R) * sizeof per (row, step)
    x the N * t_run units of the launch.  Returns None when the step did not use the persistent kernel."""
    import ctypes as C
    from unpaired_image_captioning_amd import _lib as L
    lib = L.load()
    c = CFG
    before = L.persistent_status()
    L.check(lib.uic_topdown_step_marks(1, None))
    durs = []
    try:
        for _ in range(extra_steps):
            tr.train_device_batch(batch, t_run, den_local)
            ms = (C.c_float * L.STEP_MARKS)()
            L.check(lib.uic_topdown_step_marks(1, ms))
            durs.append((ms[2] - ms[1]) * 1e-3)
    finally:
        L.check(lib.uic_topdown_step_marks(0, None))
    after = L.persistent_status()
    if (after[1] + after[2]) - (before[1] + before[2]) < extra_steps:
        return None
    durs = sorted(durs)[1:-1] if len(durs) > 4 else durs
    dur = sum(durs) / len(durs)
    N, R, A, H = batch["labels"].shape[0], c["R"], c["A"], c["H"]
    es = 2 if dtype_name == "bf16" else 4
    units = N * t_run
    bytes_per_unit = (R * A + R * H + 2 * H + R) * es
    flops_per_unit = 2.0 * (2 * H * 4 * H + H * A + 3 * H * 4 * H) + 2.0 * (R * A + R * H)   # att_lstm (recurrent inputs), h2att, lang_lstm, scores + context
    achieved = units * bytes_per_unit / dur / 1e9
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "rnn_persist_pmc_%s.json" % dtype_name)) as f:
            traffic = json.load(f)["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    # "bound": neither roofline binds this launch -- it is a chain of t_run x 4 group exchanges.  achieved / peak /
    # frac stay the HBM figures of the task's contract (attention bytes per unit over the launch duration); the MFMA figures are
    # beside them.
    return {"bound": "latency", "kernel": "rnn_fwd_persist_%skernel (one launch = %d decode steps x %d caption rows: att_lstm, h2att, attention, lang_lstm)" % ("ws_" if dtype_name == "bf16" else "", t_run, N),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "hbm_frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": "profiles/rnn_persist_pmc_%s.json (committed rocprofv3 PMC passes of this kernel, NOT measured in this run)" % dtype_name if traffic else None,
            "units_per_launch": units, "bytes_per_unit": bytes_per_unit, "bytes_per_launch": units * bytes_per_unit,
            "us_per_launch": round(dur * 1e6, 1), "us_per_decode_step": round(dur * 1e6 / t_run, 2),
            "mfma_tflops": round(units * flops_per_unit / dur / 1e12, 1), "mfma_frac": round(units * flops_per_unit / dur / 1e12 / MFMA_PEAK_TFLOPS[dtype_name], 4),
            "note": "a chain of t_run dependent steps x 4 chip-wide exchanges: bound by exchange latency and the XCD-L2 operand broadcast (DESIGN.md 4), "
                    "neither by HBM nor by MFMA; the per-(row, step) attention bytes re-read p_att / att' from the 256 MiB Infinity Cache after the first step"}


def gemm_roofline(dtype_id, dtype_name, iters=32):
    """The dominant MFMA-bound GEMM of the step on its largest call: att_embed, [N R x D] x [H x D]^T = 23040 x 512 x 2048
    (P/models/AttModel.py:76-80,111) -- since round 4 the ping-pong kernel (csrc/gemm_pp.hip, 192-row tiles at this shape) in
    bf16, the 128 x 128 LDS-DMA kernel in f32.  Achieved TFLOP/s from one HIP event pair per launch against the dense MFMA
    peak of the operand dtype; a spin kernel in front of the timed launches lets the host run ahead, so that no pair contains
    the host's enqueue time (as attention_roofline)."""
    from unpaired_image_captioning_amd import _lib as L
    lib = L.load()
    c = CFG
    M, Nn, K = c["n_img"] * c["S"] * c["R"], c["H"], c["D"]
    td = L.TORCH_DTYPE[dtype_id]
    g = torch.Generator(device="cuda").manual_seed(2)
    Aop = torch.randn(M, K, device="cuda", generator=g).to(td)
    Bop = torch.randn(Nn, K, device="cuda", generator=g).to(td)
    Cout = torch.empty(M, Nn, device="cuda", dtype=td)
    bias = torch.zeros(Nn, device="cuda")

    def launch():
        L.check(lib.uic_linear(dtype_id, M, Nn, K, L.ptr(Aop), K, L.ptr(Bop), K, L.ptr(Cout), Nn, L.ptr(bias), 1, L.stream()))   # (+ bias, ReLU)
    for _ in range(4):
        launch()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda._sleep(int(5e6))
    for a, b in ev:
        a.record()
        launch()
        b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in ev)
    dur = sum(d[iters // 8: iters - iters // 8]) / (iters - 2 * (iters // 8)) / 1e3
    flops = 2.0 * M * Nn * K
    tf = flops / dur / 1e12
    kern = "uic_gemm_pp_kernel<3> (192 x 256 ping-pong tiles)" if dtype_name == "bf16" else "uic_gemm_glds_kernel (128 x 128 tiles)"
    return {"bound": "mfma", "kernel": "%s, att_embed: %d x %d x %d" % (kern, M, Nn, K),
            "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS[dtype_name], "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS[dtype_name], 4),
            "flops_per_launch": flops, "us_per_launch": round(dur * 1e6, 2), "traffic": None}


def cpu_baseline(threads=None):
    """The CPU oracle (results-identical restatement of the reference trainer step) on this host."""
    from oracle import topdown as O
    c = CFG
    if threads:
        torch.set_num_threads(threads)
    n_img = c["n_img"]                                        # the full 128 images x 5 captions = 640 rows
    torch.manual_seed(0)
    W = O.init_weights(c["V"] + 1, c["E"], c["H"], c["A"], c["D"], c["D"], seed=7)
    b = O.synthetic_batch(n_img, c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
    N = n_img * c["S"]
    T = c["L"] + 1
    g = torch.Generator().manual_seed(1)

    def masks():
        def m(*shape):
            return (torch.rand(*shape, generator=g) >= 0.5).float() * 2.0
        return dict(fc=m(N, c["H"]), att=m(N, c["R"], c["H"]), embed=m(T, N, c["E"]), out=m(T, N, c["H"]))
    P = {k: v.clone() for k, v in W.items()}
    m1 = {k: torch.zeros_like(v) for k, v in P.items()}
    v1 = {k: torch.zeros_like(v) for k, v in P.items()}
    times = []
    for step in range(1, 14):
        t0 = time.perf_counter()
        loss, grads, _ = O.xe_loss_and_grads(P, b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], masks())
        O.adam_step(P, grads, m1, v1, step, 5e-4)
        times.append(time.perf_counter() - t0)
    best = sum(times[3:]) / len(times[3:])
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(N / best, 1), "unit": "captions/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
            "cores_note": "threads used = the fastest setting measured on the GPU box's 256-thread host (8 -> 293, 16 -> 379, 32 -> 211, "
                          "64 -> 110, 128 -> 24 captions/s: torch's CPU kernels stop scaling at this problem size)",
            "sample": "%d images x %d captions = %d rows, fp32, 3 warm-up + 10 timed steps of fwd+loss+bwd+Adam "
                      "(oracle/topdown.py, torch CPU ops)" % (n_img, c["S"], N)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the secondary f32 (parity path) measurement")
    ap.add_argument("--features", default="per-caption", choices=["per-caption", "per-image"],
                    help="per-caption: the reference's batch dict (features replicated seq_per_img times, the headline); "
                         "per-image: features once per image, replication on the device")
    ap.add_argument("--att-feat-size", type=int, default=0,
                    help="region feature width (secondary measurement): 2053 = 2048 + 5 box features, the reference's default "
                         "use_box=1; the metric is quoted at 2048")
    ap.add_argument("--use-bn", type=int, default=0, help="opt.use_bn of the captioner (secondary measurement; the metric is quoted at 0)")
    ap.add_argument("--allreduce-exchange", action="store_true", help="N > 1: round 5's exchange (all-reduce of the arena in four pieces, Adam on "
                    "everything on every rank) instead of the sharded one")
    ap.add_argument("--early-grads", action="store_true",
                    help="UIC_REC_EARLY_GRADS (opt.early_grads): the order of the gradient work that has 62 %% of the gradient bytes final "
                         "0.18 ms before the step ends, for a step that is 4 %% longer on its own -- for N > 1 experiments; off for the headline")
    ap.add_argument("--all-positions", action="store_true", help="do not hand the step the list of unmasked positions (uic_topdown_batch.live_rows): "
                    "the logit layer and the criterion then compute the positions behind the captions' ends too (exact zeros; round 5's behaviour)")
    ap.add_argument("--long-run", type=int, default=200, help="steps of the secondary `long_run` figure (0: skip); the timed region "
                    "of the default command is 65 ms -- the boxes of the pool differ by more than a round's progress")
    ap.add_argument("--rows-per-gpu-probe", type=int, default=0,
                    help="also time the step at this many caption rows, e.g. 80 = the per-rank size of a strong-scaling run of the 640-row "
                         "batch over 8 GPUs (secondary key `strong_scaling_probe`, never `value`; off by default so that the default "
                         "command launches the 640-row kernels only and its rocprofv3 summary averages one problem size)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU oracle leg (0: min(host cores, 16), the fastest setting measured on the GPU box's 256-thread host: 8->293, 16->379, 32->211, 64->110, 128->24 captions/s)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here.  Fresh child processes, started BEFORE this
        # process makes any GPU call (counting devices does not initialise the GPU on this image); it only waits for them.
        import subprocess
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and os.environ.get("UIC_BENCH_SHARE_GPU") != "1":
            raise SystemExit("bench.py --gpus %d: this node has %d GPU(s)" % (args.gpus, n_dev))
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = 0
        for pr in procs:
            rc = max(rc, pr.wait())
        raise SystemExit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d ranks were launched; refusing to report a number for a different job" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the captioner hot path has no CPU fallback")
    # UIC_BENCH_SHARE_GPU=1 (functional test of the N > 1 path on a 1-GPU box): every rank uses device 0 and the
    # collectives go through gloo, since RCCL needs one GPU per rank.  Never set for a measurement.
    share = os.environ.get("UIC_BENCH_SHARE_GPU") == "1"
    torch.cuda.set_device(0 if share else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if share else "nccl", rank=rank, world_size=world)

    from unpaired_image_captioning_amd import _lib as L
    from unpaired_image_captioning_amd.synthetic import synthetic_batch
    from unpaired_image_captioning_amd.trainer import Trainer

    c = CFG
    torch.manual_seed(1234)                                    # identical initial weights on every rank
    Datt = args.att_feat_size or c["D"]
    opt_main = make_opt(args.dtype, 1234 + rank, args.use_bn, Datt)
    opt_main.early_grads = bool(args.early_grads)
    opt_main.allreduce_exchange = int(args.allreduce_exchange)
    tr = Trainer(opt_main)
    tr.build_optimizer()
    if share:
        # two processes on ONE GPU must not both run the persistent recurrence kernel (each wants every CU for itself and waits,
        # bounded, for the other: include/uic_hip.h): the functional N > 1 test on a 1-GPU box uses the per-step launch chain
        tr.i2t_model.engine.recurrence |= L.REC_FWD_CHAIN
    batch = synthetic_batch(c["n_img"], c["S"], c["R"], Datt, c["V"], c["L"], seed=1234 + rank)
    batch["fc_feats"] = batch["fc_feats"][:, :c["D"]].contiguous()
    N = c["n_img"] * c["S"]
    T = c["L"] + 1
    t_run = tr.i2t_model._steps_to_run(batch["labels"])
    den_local = float(batch["masks"][:, 1:T + 1].sum().item())
    # The list of unmasked (step, row) positions, made from the masks on the host like the step count and the mask sum above (the
    # loader builds the masks there; Trainer.to_device attaches the list to every real batch): the logit layer and the criterion
    # skip the positions behind the captions' ends.  --all-positions computes them all (their loss and gradient are exact zeros).
    if not args.all_positions:
        tr.attach_live(batch)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(bt):
        # (the synthetic batch is resident and the same every step: its mask sum is "the next batch's" too -- a data-parallel run
        # carries it in the step's small all-reduce instead of a collective in front of the next forward pass, as Trainer.train
        # does with next_data=)
        # (a full pass of Python's cyclic garbage collector over the interpreter's ~1e6 objects takes ~40 ms on these hosts -- 15 steps --
        # and falls wherever the allocation count happens to trip it: collect now and move what is alive out of the collector's
        # way, as a training loop that cares does once after start-up, instead of timing the collector.  In FRONT of the warm-up
        # steps: the GPU goes from them straight into the timed region)
        gc.collect()
        gc.freeze()
        for _ in range(args.warmup):
            tr.train_device_batch(bt, t_run, den_local, den_local)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ls = tr.train_device_batch(bt, t_run, den_local, den_local)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el, ls

    # the same images shipped once each (the S-fold replication of DataLoader.get_batch done on the device,
    # uic_topdown_dims.seq_per_img): identical results, reported beside the headline, never as `value`
    per_image = dict(batch)
    for k in ("fc_feats", "att_feats", "att_masks"):
        per_image[k] = batch[k][::c["S"]].contiguous()
    if args.features == "per-image":
        batch = per_image
    elapsed, loss = timed(batch)
    elapsed_img = None
    if args.features == "per-caption" and world == 1:
        elapsed_img, _ = timed(per_image)
    loss_val = float(loss.item())
    L.persistent_status()                                      # raises if a persistent-kernel spin timed out
    long_run = None
    if args.long_run > 0:
        # the same step over a window long enough to tell a 1 % change from noise (a secondary key; `value` is the contract's K steps)
        keep_steps = args.steps
        args.steps = args.long_run
        el_long, _ = timed(batch)
        args.steps = keep_steps
        long_run = {"steps": args.long_run, "ms_per_step": round(el_long / args.long_run * 1e3, 4),
                    "value": round(world * N * args.long_run / el_long, 1), "unit": "captions/s"}
    strong = None
    if world == 1 and args.rows_per_gpu_probe:
        # What one rank of a STRONG-scaling run (640 rows over 8 GPUs) would do per step: the same step on rows_per_gpu_probe
        # caption rows -- a latency chain at that size -- next to the host time of enqueueing it (a secondary key, never `value`).
        n_img_s = max(1, args.rows_per_gpu_probe // c["S"])
        small = {k: v[:n_img_s * c["S"]].contiguous() for k, v in batch.items() if not k.startswith("live_")}
        if not args.all_positions:
            tr.attach_live(small)
        den_s = float(small["masks"][:, 1:T + 1].sum().item())
        for _ in range(5):
            tr.train_device_batch(small, t_run, den_s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = 0.0
        for _ in range(20):
            h0 = time.perf_counter()
            tr.train_device_batch(small, t_run, den_s)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        el_s = time.perf_counter() - t0
        strong = {"rows_per_gpu": n_img_s * c["S"], "ms_per_step": round(el_s / 20 * 1e3, 3), "host_enqueue_ms_per_step": round(host / 20 * 1e3, 3),
                  "captions_per_s_per_gpu": round(n_img_s * c["S"] * 20 / el_s, 1),
                  "note": "1-GPU step at the per-rank size of a strong-scaling run (640 rows / 8 GPUs); no collective in it"}
    comm = None
    if world > 1:
        # exposed communication: the same step on every rank WITHOUT the exchange (same stream layout: UIC_REC_COMM_STREAM stays set),
        # timed rank-locally right after the measurement; what the collectives add to a step is ms_per_step minus this
        if not share and dist.get_backend() != "nccl":
            raise SystemExit("bench.py: %d ranks but the backend is %s, not RCCL: refusing to report a scaling number" % (world, dist.get_backend()))
        from unpaired_image_captioning_amd.parallel_exchange import GradientExchange

        class _Alone(GradientExchange):
            world_size = property(lambda self: 1)
            rank = property(lambda self: 0)

            def _sum(self, t):
                pass

        real = tr.exchange
        tr.exchange = _Alone()
        for _ in range(args.warmup):
            tr.train_device_batch(batch, t_run, den_local, den_local)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.train_device_batch(batch, t_run, den_local, den_local)
        torch.cuda.synchronize()
        alone_ms = (time.perf_counter() - t0) / args.steps * 1e3
        tr.exchange = real
        tr._next_den = None
        per_rank = [None] * world
        dist.all_gather_object(per_rank, round(alone_ms, 4))
        step_ms = elapsed / args.steps * 1e3
        comm = {"ms_per_step_without_exchange_per_rank": per_rank,
                "exposed_communication_ms_per_rank": [round(step_ms - a, 4) for a in per_rank],
                "gradient_bytes_per_step": int(tr.arena.grad.numel() * 4),
                "exchange": "sharded: reduce-scatter of %d gradient pieces + all-reduce of the replicated tail (%d bytes) + Adam on 1/%d of the arena + "
                            "all-gather of the %s weights (%d bytes) beside the next forward pass" % (
                                len(tr.arena.pieces), (tr.arena.scalars_off + 4 - tr.arena.repl_off) * 4, world,
                                "bf16" if tr.arena.w16 is not None else "f32", tr.arena.repl_off * (2 if tr.arena.w16 is not None else 4))
                if getattr(tr, "sharded", False) else "all-reduce in %d pieces" % (len(getattr(tr, "arena_splits", [])) + 1),
                "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                "note": "step time of each rank with the collectives removed (same stream layout), measured after the timed region; "
                        "exposed = ms_per_step - that"}
    rec_roof = recurrence_roofline(tr, batch, t_run, den_local, args.dtype)   # untimed extra steps, every rank (collectives stay matched)
    f32_line = None
    if world == 1 and args.dtype != "f32" and not args.no_f32:
        # the reference's own precision (exact-f32 MFMA parity path): same step, same batch, a few iterations
        del tr
        torch.manual_seed(1234)
        tr32 = Trainer(make_opt("f32", 1234, args.use_bn, Datt))
        tr32.build_optimizer()
        for _ in range(2):
            tr32.train_device_batch(batch, t_run, den_local)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            l32 = tr32.train_device_batch(batch, t_run, den_local)
        torch.cuda.synchronize()
        e32 = time.perf_counter() - t0
        f32_line = {"value": round(N * 5 / e32, 1), "unit": "captions/s", "ms_per_step": round(e32 / 5 * 1e3, 3), "steps": 5,
                    "warmup": 2, "final_loss": round(float(l32.item()), 4), "note": "same step with f32 operands (exact-f32 MFMA), the reference's precision"}
        del tr32

    if rank == 0:
        dtype_id = L.dtype_id(args.dtype)
        out = {
            "metric": "train captions/sec, TopDown-attn LSTM on 36x2048 feats, batch 128, 1/2/4/8 GPU",
            "value": round(world * N * args.steps / elapsed, 1),
            "unit": "captions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic" if not share else "synthetic (UIC_BENCH_SHARE_GPU functional test: ranks share one GPU, gloo; NOT a measurement)",
            "config": {"workload": "BASELINE configs[1]: TopDown attention LSTM, 128 images x 5 captions = 640 caption "
                                   "rows per GPU, R=36, D=2048, H=E=A=512, V+1=9488, 17 decode steps, dropout 0.5, "
                                   "XE loss + BPTT + Adam", "rows_per_gpu": N, "parallelism": "dp%d" % world, "use_bn": args.use_bn, "att_feat_size": Datt,
                       "features": args.features, "early_grads": bool(args.early_grads),
                       "positions": "all" if args.all_positions else "unmasked (%d of %d; counted per step on the host from the masks, outside the timed region)"
                       % (int(batch["live_count"][:t_run].sum()), t_run * N)},
            "final_loss": round(loss_val, 4),
            "rccl_ranks": dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else (1 if world == 1 else 0),
            "roofline_mfma": gemm_roofline(dtype_id, args.dtype),
        }
        # `roofline`: the dominant single launch of the timed region -- the persistent recurrence kernel when the step uses it;
        # the stand-alone attention-step kernel (decode paths, per-step launch chain) is reported beside it
        att_roof = attention_roofline(dtype_id, args.dtype)
        if rec_roof is not None:
            out["roofline"] = rec_roof
            out["roofline_attention_step"] = att_roof
        else:
            out["roofline"] = att_roof
        if f32_line is not None:
            out["f32"] = f32_line
        if long_run is not None:
            out["long_run"] = long_run
        if comm is not None:
            out["communication"] = comm
            if not share and out["rccl_ranks"] != world:
                raise SystemExit("bench.py: rccl_ranks %s != --gpus %d" % (out["rccl_ranks"], world))
        if strong is not None:
            out["strong_scaling_probe"] = strong
        if elapsed_img is not None:
            out["per_image_features"] = {"value": round(world * N * args.steps / elapsed_img, 1), "unit": "captions/s",
                                         "ms_per_step": round(elapsed_img / args.steps * 1e3, 3),
                                         "note": "same step, features shipped once per image (dims.seq_per_img = 5)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_threads or min(os.cpu_count() or 1, 16))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
