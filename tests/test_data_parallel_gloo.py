"""World-size-2 data-parallel exchange on CPU (gloo): summing the per-rank gradients that were scaled by
1 / (global mask sum) reproduces the single-process whole-batch gradient and loss of the oracle."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import topdown as O
    from unpaired_image_captioning_amd.parallel_exchange import GradientExchange
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    ex = GradientExchange()
    assert ex.world_size == world and ex.rank == rank
    lo, hi = ex.shard_images(cfg["n_img"])
    rows = slice(lo * cfg["S"], hi * cfg["S"])                 # shard by image: the S replicas stay together
    sub = {k: I[k][rows] for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}
    T = sub["labels"].shape[1] - 1
    den_local = float(sub["masks"][:, 1:T + 1].sum())
    inv_den = ex.global_inv_den(den_local, torch.device("cpu"))
    # local step of the checker with the GLOBAL denominator (what uic_topdown_xe_loss(inv_den=...) computes)
    loss_l, grads_l, _ = O.xe_loss_and_grads(W, sub["fc_feats"], sub["att_feats"], sub["labels"], sub["masks"], sub["att_masks"])
    scale = den_local * float(inv_den)
    names = list(grads_l.keys())
    flat = torch.cat([(grads_l[k] * scale).reshape(-1) for k in names])
    ex.allreduce_sum(flat)
    loss = ex.allreduce_sum_scalar(loss_l * scale)
    if rank == 0:
        torch.save({"flat": flat, "loss": loss, "names": names, "shapes": [tuple(grads_l[k].shape) for k in names]},
                   os.path.join(out_dir, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "dp.pt"))
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    assert abs(float(res["loss"]) - float(Out["loss"])) < 1e-5
    off = 0
    for k, shape in zip(res["names"], res["shapes"]):
        n = 1
        for s in shape:
            n *= s
        g = res["flat"][off:off + n].view(shape)
        off += n
        ref = G[k]
        assert (g - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), k


def test_single_rank_exchange_is_identity():
    from unpaired_image_captioning_amd.parallel_exchange import GradientExchange
    ex = GradientExchange()
    assert ex.world_size == 1 and ex.rank == 0
    assert ex.global_inv_den(12.0, torch.device("cpu")) is None
    x = torch.arange(4.0)
    assert ex.allreduce_sum(x) is x and torch.equal(ex.allreduce_sum_scalar(x), x)
    assert ex.shard_images(128) == (0, 128)


def _sc_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import topdown as O
    from unpaired_image_captioning_amd.parallel_exchange import GradientExchange
    from unpaired_image_captioning_amd.trainer import Trainer
    ex = GradientExchange()
    logp, seq, reward = _sc_case()
    n = seq.shape[0] // world
    rows = slice(rank * n, (rank + 1) * n)
    lp = logp[rows].clone().requires_grad_(True)
    # the rank's RewardCriterion mean, rescaled to its share of the whole batch's mask sum (Trainer.train_self_critical)
    loss = O.reward_criterion(lp, seq[rows], reward[rows]) * ex.global_share(Trainer._reward_mask_sum(seq[rows]))
    loss.backward()
    total = ex.allreduce_sum_scalar(loss.detach().reshape(1))
    gathered = [torch.zeros_like(lp.grad) for _ in range(world)]
    dist.all_gather(gathered, lp.grad)
    if rank == 0:
        torch.save({"loss": total, "grad": torch.cat(gathered)}, os.path.join(out_dir, "sc.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _sc_case():
    g = torch.Generator().manual_seed(21)
    N, L = 12, 7
    logp = -torch.rand(N, L, generator=g) * 3
    seq = torch.randint(1, 50, (N, L), generator=g)
    for n in range(N):
        seq[n, (1 + n % 2 if n < N // 2 else 4 + n % 3):] = 0      # unequal lengths: the first half's captions are much shorter
    reward = torch.randn(N, 1, generator=g).repeat(1, L)
    return logp, seq, reward


def test_two_rank_self_critical_loss_uses_the_whole_batch_mask_sum(tmp_path):
    """RewardCriterion across ranks (P/misc/criterion.py:117-122 on the gathered batch, P/trainer.py:168-170): per-rank means
    rescaled by mask_sum_rank / mask_sum_all and then SUMMED equal the single-process loss and gradient -- and the plain average
    of the per-rank means (what round 4 did) does not."""
    from oracle import topdown as O
    world = 2
    mp.spawn(_sc_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "sc.pt"))
    logp, seq, reward = _sc_case()
    lp = logp.clone().requires_grad_(True)
    ref = O.reward_criterion(lp, seq, reward)
    ref.backward()
    assert abs(float(res["loss"]) - float(ref)) < 1e-6
    assert (res["grad"] - lp.grad).abs().max().item() < 1e-7
    n = seq.shape[0] // 2
    naive = 0.5 * (O.reward_criterion(logp[:n], seq[:n], reward[:n]) + O.reward_criterion(logp[n:], seq[n:], reward[n:]))
    assert abs(float(naive) - float(ref)) > 1e-3      # the shards really weigh differently
