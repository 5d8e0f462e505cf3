"""Two ranks sharing ONE GPU over gloo (functional probe, not a measurement): per-step enqueue / total time of Trainer.train_device_batch\nwith the gradient exchange in one piece (plain), two pieces (one) or four (four).  torchrun --nproc-per-node 2 tools/dp_probe.py four"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from unpaired_image_captioning_amd.synthetic import synthetic_batch
from unpaired_image_captioning_amd.trainer import Trainer
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
c = bench.CFG
torch.manual_seed(1234)
tr = Trainer(bench.make_opt("bf16", 1234 + rank)); tr.build_optimizer()
batch = synthetic_batch(c["n_img"], c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234 + rank)
t_run = tr.i2t_model._steps_to_run(batch["labels"])
den = float(batch["masks"][:, 1:c["L"] + 2].sum().item())
mode = sys.argv[1]
if mode == "plain":
    tr.arena_splits = []
elif mode == "one":
    tr.arena_splits = tr.arena_splits[-1:]
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_device_batch(batch, t_run, den)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    if rank == 0: print("%s step %d: enqueue %.1f ms, total %.1f ms" % (mode, it, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
dist.destroy_process_group()
