#!/usr/bin/env python3
"""Can the fused training step be captured into a hipGraph and replayed?  (the C ABI is enqueue-only "so hipGraph
capture works"; VERDICT round 4: untested.)  RESULT, round 5 (ROCm 7.2, MI355X): NO -- the multi-stream capture is accepted
launch by launch (an un-joined side stream is reported as hipErrorStreamCaptureUnjoined, which is how the ordering of the
gradient-group flag kernels was found and fixed), but hipStreamEndCapture then dies with a segmentation fault inside the
runtime.  csrc/topdown.hip says so; this probe is the evidence and the way to re-check on a newer ROCm.  Captures uic_topdown_refresh_weights + uic_topdown_xe_train_step -- the caller's stream
plus the library's four side streams, forked and joined by events -- in relaxed mode, replays the graph and compares every
gradient bit for bit with the eager step.    gpurun -- python tools/graph_capture_probe.py [--full]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler
import torch

faulthandler.enable()


def capture_and_replay(model, batch, replays=3, verbose=True):
    """Returns (eager grads, [replayed grads, ...], eager ms, replay ms)."""
    from unpaired_image_captioning_amd import _lib as L
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    eng = model.engine
    pd = {k: v.detach() for k, v in model.param_dict().items()}
    fc, att, am, labels, masks = batch["fc_feats"], batch["att_feats"], batch.get("att_masks"), batch["labels"], batch["masks"]
    t_run = model._steps_to_run(labels)
    seed = 4242

    def step(grads):
        return eng.xe_train_step(pd, fc, att, am, labels, masks, t_run, True, seed, grads, keep_workspace=True)

    ref = {k: torch.zeros_like(v) for k, v in pd.items()}
    for _ in range(2):                                       # warm-up: one-time attribute calls, the allocator's cache
        out, ws = step(ref)
        torch.cuda.synchronize()
        eng.release(ws)
    t0 = time.perf_counter()
    out_ref, ws = step(ref)
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) * 1e3
    eng.release(ws)
    loss_ref = out_ref.clone()

    got = {k: torch.zeros_like(v) for k, v in pd.items()}
    cs = torch.cuda.Stream()
    graph, gexec = C.c_void_p(), C.c_void_p()
    torch.cuda.synchronize()
    with torch.cuda.stream(cs):
        rc = hip.hipStreamBeginCapture(C.c_void_p(cs.cuda_stream), 2)              # hipStreamCaptureModeRelaxed
        assert rc == 0, "hipStreamBeginCapture: %d" % rc
        print("capture begun", flush=True)
        try:
            out, ws = step(got)
            print("step enqueued into the capture", flush=True)
        finally:
            rc = hip.hipStreamEndCapture(C.c_void_p(cs.cuda_stream), C.byref(graph))
        print("hipStreamEndCapture -> %d" % rc, flush=True)
        assert rc == 0 and graph.value, "hipStreamEndCapture: %d" % rc
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    rc = hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0)
    print("hipGraphInstantiate -> %d" % rc, flush=True)
    assert rc == 0, "hipGraphInstantiate: %d" % rc
    nnodes = C.c_size_t(0)
    hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
    hip.hipGraphDestroy.argtypes = [C.c_void_p]
    hip.hipGraphGetNodes(graph, None, C.byref(nnodes))
    if verbose:
        print("captured %d graph nodes" % nnodes.value)
    results = []
    replay_ms = 0.0
    for _ in range(replays):
        for g in got.values():
            g.fill_(float("nan"))
        out.fill_(float("nan"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = hip.hipGraphLaunch(gexec, C.c_void_p(cs.cuda_stream))
        assert rc == 0, "hipGraphLaunch: %d" % rc
        cs.synchronize()
        replay_ms = (time.perf_counter() - t0) * 1e3
        results.append(({k: v.clone() for k, v in got.items()}, out.clone()))
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    torch.cuda.synchronize()
    eng.release(ws)
    return (ref, loss_ref), results, eager_ms, replay_ms, nnodes.value


def capture_single_stream(model, batch, replays=5, rows=80):
    """Round 6: the SINGLE-stream form of the step -- uic_topdown_forward + uic_topdown_xe_loss +
    uic_topdown_backward with the launch-chain recurrence (UIC_REC_FWD_CHAIN: no persistent launch, whose gate waits on an event
    of an earlier, un-captured launch), everything on the capturing stream -- at the per-rank size of a strong-scaling run.
    Captured once, replayed, every gradient compared bit for bit with the eager calls.  The dropout seed is a launch parameter:
    a replay repeats the captured step's masks, so this measures what a replayed step costs, not a usable training loop
    (target 7)."""
    from unpaired_image_captioning_amd import _lib as L
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    eng = model.engine
    eng.recurrence = L.REC_FWD_CHAIN
    S = 5
    n = rows // S * S
    small = {k: v[:n].contiguous() for k, v in batch.items()}
    pd = {k: v.detach() for k, v in model.param_dict().items()}
    fc, att, am, labels, masks = small["fc_feats"], small["att_feats"], small.get("att_masks"), small["labels"], small["masks"]
    t_run = model._steps_to_run(labels)
    R, T = att.shape[1], labels.shape[1] - 1 if False else labels.shape[1] - 1
    seed = 4242

    def step(grads):
        _, ws, (d, w, bs) = eng.forward(pd, fc, att, am, labels, t_run, True, seed, want_logprobs=False, masks=masks)
        loss = eng.xe_loss(ws, d, bs, t_run)
        eng.backward(ws, d, w, bs, t_run, True, seed, grads)
        return loss, ws

    ref = {k: torch.zeros_like(v) for k, v in pd.items()}
    for _ in range(3):
        loss_ref, ws = step(ref)
        torch.cuda.synchronize()
        eng.release(ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        loss_ref, ws = step(ref)
        eng.release(ws)
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / 20 * 1e3
    loss_ref = loss_ref.clone() if torch.is_tensor(loss_ref) else loss_ref
    ref = {k: v.clone() for k, v in ref.items()}

    got = {k: torch.zeros_like(v) for k, v in pd.items()}
    cs = torch.cuda.Stream()
    graph, gexec = C.c_void_p(), C.c_void_p()
    torch.cuda.synchronize()
    with torch.cuda.stream(cs):
        loss_w, ws = step(got)                              # once on this stream, eagerly: its allocations and attribute calls happen here
        cs.synchronize()
        eng.release(ws)
        rc = hip.hipStreamBeginCapture(C.c_void_p(cs.cuda_stream), 2)              # hipStreamCaptureModeRelaxed
        assert rc == 0, "hipStreamBeginCapture: %d" % rc
        try:
            loss_g, ws = step(got)
        finally:
            rc = hip.hipStreamEndCapture(C.c_void_p(cs.cuda_stream), C.byref(graph))
        print("hipStreamEndCapture -> %d" % rc, flush=True)
        assert rc == 0 and graph.value, "hipStreamEndCapture: %d" % rc
    rc = hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0)
    assert rc == 0, "hipGraphInstantiate: %d" % rc
    nnodes = C.c_size_t(0)
    hip.hipGraphGetNodes(graph, None, C.byref(nnodes))
    bad_all = []
    for _ in range(2):
        for g in got.values():
            g.fill_(float("nan"))
        torch.cuda.synchronize()
        assert hip.hipGraphLaunch(gexec, C.c_void_p(cs.cuda_stream)) == 0
        cs.synchronize()
        bad_all.append([k for k in ref if not torch.equal(ref[k], got[k])])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays * 10):
        assert hip.hipGraphLaunch(gexec, C.c_void_p(cs.cuda_stream)) == 0
    cs.synchronize()
    replay_ms = (time.perf_counter() - t0) / (replays * 10) * 1e3
    eng.release(ws)
    eng.recurrence = 0
    print("single-stream step at %d caption rows (forward + loss + backward, launch-chain recurrence): eager %.3f ms per step, "
          "graph replay %.3f ms, %d nodes; tensors that differ from the eager step in two replays: %s" % (
              n, eager_ms, replay_ms, nnodes.value, [b or "none" for b in bad_all]))


if __name__ == "__main__":
    from bench import CFG, make_opt
    from unpaired_image_captioning_amd import models
    from unpaired_image_captioning_amd.synthetic import synthetic_batch
    c = CFG
    model = models.setup(make_opt("bf16", 1234)).cuda()
    model.train()
    batch = {k: v.cuda() for k, v in synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1).items()}
    if "--single-stream" in sys.argv:
        capture_single_stream(model, batch)
        sys.exit(0)
    (ref, loss_ref), results, eager_ms, replay_ms, nn = capture_and_replay(model, batch)
    for i, (g, out) in enumerate(results):
        bad = [k for k in ref if not torch.equal(ref[k], g[k])]
        print("replay %d: loss %s (eager %s), tensors that differ from the eager step: %s" % (i, out.tolist(), loss_ref.tolist(), bad or "none"))
    print("eager step (enqueue + run) %.3f ms, graph replay %.3f ms, %d nodes" % (eager_ms, replay_ms, nn))
