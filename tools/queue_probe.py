#!/usr/bin/env python3
"""How many busy hardware queues the fused step tolerates: the step alone, beside ONE long one-wave kernel on another stream (a
queue that stays non-empty for the whole step and uses no resources to speak of), and beside one such kernel on each of two
streams.    python tools/queue_probe.py [recurrence bits]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd import models, trainer
from unpaired_image_captioning_amd.synthetic import synthetic_batch
import ctypes as C
lib = L.load()
c = CFG
model = models.setup(make_opt("bf16", 1234)).cuda()
model.train()
model.engine.recurrence = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
batch = trainer.Trainer.attach_live({k: v.cuda() for k, v in synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1).items()})
extra = [torch.cuda.Stream() for _ in range(2)]
names = ["start", "prologue", "recurrence", "logit layer", "BPTT starts", "BPTT done", "rec wgrads", "main tail", "side tail", "joined", "logit grads"]
for n_busy in (0, 1, 2, 0):
    for it in range(8):
        if it == 4:
            L.check(lib.uic_topdown_step_marks(1, None))
        torch.cuda.synchronize()
        for st in extra[:n_busy]:
            with torch.cuda.stream(st):
                torch.cuda._sleep(int(4e-3 * 2.0e9))          # ~4 ms of one wave spinning
        t0 = time.perf_counter()
        loss, g = trainer.xe_step(model, batch)
        loss.item()
        dt = (time.perf_counter() - t0) * 1e3
    ms = (C.c_float * L.STEP_MARKS)()
    L.check(lib.uic_topdown_step_marks(1, ms))
    L.check(lib.uic_topdown_step_marks(0, None))
    torch.cuda.synchronize()
    print("%d busy extra queue(s): step wall %.3f ms; " % (n_busy, dt) + "  ".join("%s %.3f" % (n, v) for n, v in zip(names[1:], list(ms)[1:])))
