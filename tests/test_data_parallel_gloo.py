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


# ---------------------------------------------------------------------------------------------------------------------------
# The sharded exchange (round 6): arena layout, ownership and the collectives' call pattern on CPU.  The Adam kernel itself is HIP
# (tests/test_gpu_dp2.py runs the real step); here a plain-torch Adam stands in for it, on exactly the ranges the arena hands out.

def _torch_adam(p, g, m, v, ranges, lr, step, w16=None):
    b1, b2, eps = 0.9, 0.999, 1e-8
    for lo, hi in ranges:
        m[lo:hi].mul_(b1).add_(g[lo:hi], alpha=1 - b1)
        v[lo:hi].mul_(b2).addcmul_(g[lo:hi], g[lo:hi], value=1 - b2)
        denom = v[lo:hi].sqrt() / (1 - b2 ** step) ** 0.5 + eps
        p[lo:hi].sub_(lr / (1 - b1 ** step) * m[lo:hi] / denom)
        if w16 is not None and hi <= w16.numel():
            w16[lo:hi].copy_(p[lo:hi])


def _tiny_module(W):
    import torch.nn as nn
    m = nn.Module()
    for k, v in W.items():
        m.register_parameter(k.replace(".", "_"), nn.Parameter(v.clone()))
    return m


def _sharded_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import topdown as O
    from unpaired_image_captioning_amd.misc.optimizer import FlatArena
    from unpaired_image_captioning_amd.parallel_exchange import GradientExchange
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    ex = GradientExchange()
    lo, hi = ex.shard_images(cfg["n_img"])
    rows = slice(lo * cfg["S"], hi * cfg["S"])
    sub = {k: I[k][rows] for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}
    T = sub["labels"].shape[1] - 1
    den_local = float(sub["masks"][:, 1:T + 1].sum())
    inv_den = ex.global_inv_den(den_local, torch.device("cpu"))
    loss_l, grads_l, _ = O.xe_loss_and_grads(W, sub["fc_feats"], sub["att_feats"], sub["labels"], sub["masks"], sub["att_masks"])
    scale = den_local * float(inv_den)
    key = lambda k: k.replace(".", "_")
    mats = [k for k, v in W.items() if v.dim() == 2 and v.shape[0] > 1]
    pieces = [[key(k) for k in mats if k.startswith("logit.")], [key(k) for k in mats if "lstm" in k],
              [key(k) for k in mats if k.startswith(("embed.", "fc_embed."))], [key(k) for k in mats if k.startswith(("att_embed.", "ctx2att.", "core.attention."))]]
    module = _tiny_module(W)
    a = FlatArena(module, [key(k) for k in W], world=world, rank=rank, pieces=pieces, operand_dtype=torch.bfloat16)
    assert len(a.pieces) == 4 and all(n % (64 * world) == 0 for _, n in a.pieces) and a.w16 is not None and a.w16.numel() == a.numel
    assert set(a.replicated) == {key(k) for k in W if k not in mats} and a.numel == a.scalars_off + FlatArena.SCALAR_SLOTS
    assert torch.equal(a.w16[:a.repl_off].float(), a.flat[:a.repl_off].bfloat16().float())            # built from the masters
    for k, g in grads_l.items():
        a.grad_views[key(k)].copy_(g * scale)
    a.scalars[0] = float(loss_l) * scale
    a.scalars[1] = 0.0
    a.scalars[3] = 17.0 + rank                       # "the next batch's mask sum" of this rank
    # -- the exchange, as FlatArena.sharded_step sequences it
    for i in range(len(a.pieces)):
        ex.reduce_scatter(a.grad, *a.pieces[i])
    ex._sum(a.grad[a.repl_off:a.scalars_off + 4])
    owned = a.owned_ranges()
    assert len(owned) == 5 and owned[-1] == (a.repl_off, a.repl_end)
    _torch_adam(a.flat, a.grad, a.exp_avg, a.exp_avg_sq, owned, 5e-3, 1, a.w16)
    mine_before = [a.flat[l:h].clone() for l, h in owned]
    for i in reversed(range(len(a.pieces))):
        ex.all_gather(a.w16, *a.pieces[i])
    # the f32 masters of the OTHER rank's slices are still the old ones (stale) until the explicit gather
    other = a.shard(0, rank=1 - rank)
    assert torch.equal(a.flat[other[0]:other[1]], torch.cat([W[k].reshape(-1) for k in W if key(k) in a.piece_names[0]] +
                                                             [torch.zeros(a.pieces[0][1])])[other[0] - a.pieces[0][0]:other[1] - a.pieces[0][0]])
    a.masters_stale = True
    a.gather_masters(ex)
    for (l, h), t in zip(owned, mine_before):
        assert torch.equal(a.flat[l:h], t)
    torch.save({"flat": a.flat.clone(), "w16": a.w16.clone(), "scalars": a.scalars[:4].clone(), "offsets": dict(a.offsets),
                "repl": (a.repl_off, a.repl_end), "grad_owned": [(l, h, a.grad[l:h].clone()) for l, h in owned]},
               os.path.join(out_dir, "sh%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_exchange_equals_all_reduce_and_full_adam(tmp_path):
    """reduce-scatter -> Adam on the rank's slices -> all-gather (bf16 copy) + gather of the f32 masters, two gloo ranks on CPU,
    against ONE process that takes the reference's whole-batch gradients (golden) through the same Adam on the whole arena: the
    summed gradient slices each rank owns, the loss, the carried mask sum, the full f32 masters and the gathered bf16 copy."""
    world = 2
    mp.spawn(_sharded_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "sh%d.pt" % i)) for i in range(world)]
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    key = lambda k: k.replace(".", "_")
    # both ranks end with identical full masters and identical bf16 copies
    assert torch.equal(r[0]["flat"], r[1]["flat"]) and torch.equal(r[0]["w16"], r[1]["w16"])
    assert torch.equal(r[0]["scalars"], r[1]["scalars"])
    assert abs(float(r[0]["scalars"][0]) - float(Out["loss"])) < 1e-5 and float(r[0]["scalars"][3]) == 17.0 + 18.0
    # single process: golden gradients, Adam on everything
    off = r[0]["offsets"]
    n = r[0]["flat"].numel()
    p, g = torch.zeros(n), torch.zeros(n)
    for k in W:
        o = off[key(k)]
        p[o:o + W[k].numel()] = W[k].reshape(-1)
        g[o:o + W[k].numel()] = G[k].reshape(-1)
    m, v = torch.zeros(n), torch.zeros(n)
    _torch_adam(p, g, m, v, [(0, r[0]["repl"][1])], 5e-3, 1)
    # the summed gradient each rank owns = the whole-batch gradient there (summation order aside)
    gmax = float(g.abs().max())
    for rk in range(world):
        for l, h, gs in r[rk]["grad_owned"]:
            assert (gs - g[l:h]).abs().max().item() <= 2e-5 * max(1.0, gmax), (rk, l, h)
    # Adam's first step is lr * sign(g) wherever |g| >> eps: compare where the gradient is not rounding noise
    sig = g.abs() > 1e-4 * gmax
    assert (r[0]["flat"][:n][sig] - p[sig]).abs().max().item() < 1e-5
    assert torch.equal(r[0]["w16"][:r[0]["repl"][0]].float(), r[0]["flat"][:r[0]["repl"][0]].bfloat16().float())
