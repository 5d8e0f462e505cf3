#!/usr/bin/env python3
"""Run the captioner forward (feature projection, ONE persistent recurrence launch, logit layer) a few times at the bench
shapes, for PMC passes over the persistent kernel:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT/fetch -- python3 /root/repo/tools/rnn_kernel_only.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d OUT/write -- python3 /root/repo/tools/rnn_kernel_only.py
    python3 tools/pmc_traffic.py OUT rnn_fwd_persist profiles/rnn_persist_pmc_bf16.json
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L
from unpaired_image_captioning_amd import models
from unpaired_image_captioning_amd.synthetic import synthetic_batch

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
c = CFG
torch.manual_seed(1234)
model = models.setup(make_opt(dtype, 1234)).cuda()
model.train()
eng = model.engine
batch = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
t_run = model._steps_to_run(batch["labels"])
params = model.param_dict()
for _ in range(16):
    _, ws, _ = eng.forward(params, batch["fc_feats"], batch["att_feats"], None, batch["labels"], t_run, True, 77, want_logprobs=False)
    eng.release(ws)
torch.cuda.synchronize()
print("status", L.persistent_status())
