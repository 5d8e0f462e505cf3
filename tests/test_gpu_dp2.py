"""Two data-parallel ranks on ONE MI355X (both processes on cuda:0, `gloo` carrying the device tensors): the whole
Trainer step -- global mask-sum denominator, fused HIP step on each rank's image shard, the gradient exchange behind
uic_topdown_grad_ready_wait, Adam -- must leave the same weights as one process training on the whole batch.  Both
exchanges are run: the SHARDED one (round 6 default: reduce-scatter of the gradient pieces, Adam on the rank's slices,
all-gather of the operand-dtype weights consumed by uic_topdown_refresh_weights_gathered) and round 5's all-reduce in four
pieces (opt.allreduce_exchange) -- and they must leave BIT-IDENTICAL weights, in f32 and in bf16.
(RCCL itself needs one GPU per rank; the collectives' call pattern is what is tested.)"""
import argparse
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _opt(cfg, use_bn=0, early_grads=False, allreduce=False, dtype="f32", half=False):
    return argparse.Namespace(early_grads=early_grads, allreduce_exchange=int(allreduce), bf16_gradient_exchange=int(half), allow_many_hw_queues=1, vocab_size=cfg["V"], input_encoding_size=cfg["E"], rnn_size=cfg["H"], num_layers=1,
                              drop_prob_lm=0.0, seq_length=cfg["L"], fc_feat_size=cfg["D"], att_feat_size=cfg["D"],
                              att_hid_size=cfg["A"], use_bn=use_bn, logit_layers=1, caption_model="topdown",
                              compute_dtype=dtype, seed=5, i2t_learning_rate=5e-3, i2t_train_flag=1)


def _train(cfg, W, data, steps, exchange=None, early_grads=False, allreduce=False, dtype="f32", use_bn=0, next_data=False, half=False):
    from unpaired_image_captioning_amd.trainer import Trainer
    tr = Trainer(_opt(cfg, use_bn=use_bn, early_grads=early_grads, allreduce=allreduce, dtype=dtype, half=half), exchange=exchange)
    if use_bn:      # att_embed = [BatchNorm1d, Linear, ...]: the golden Linear moves to index 1, the BatchNorm keeps its initial values
        W = {k.replace("att_embed.0.", "att_embed.1."): v for k, v in W.items()}
    tr.i2t_model.load_state_dict(W, strict=not use_bn)
    tr.build_optimizer()
    grads1 = None
    losses = []
    for i in range(steps):
        # next_data: the NEXT batch's mask sum rides in this step's small all-reduce (a dict on even steps, a callable on odd ones)
        nd = None if not next_data else (data if i % 2 == 0 else (lambda: data))
        losses.append(tr.train(data, next_data=nd))
        if i == 0 and not getattr(tr, "sharded", False):
            torch.cuda.synchronize()
            grads1 = {k: v.detach().cpu().clone() for k, v in tr.arena.grad_views.items()}     # (summed over the ranks by the all-reduce)
    torch.cuda.synchronize()
    tr.grads_after_first_step = grads1
    return tr, losses


def _worker(rank, world, port, out_dir, backend="gloo", uic_comm=False, early_grads=False, allreduce=False, dtype="f32", tag="dp2",
            use_bn=0, next_data=False, half=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank if (backend == "nccl" or uic_comm) else 0)     # RCCL: one GPU per rank; gloo: both ranks share cuda:0
    dist.init_process_group(backend, rank=rank, world_size=world)
    from unpaired_image_captioning_amd.parallel_exchange import GradientExchange, UicCommExchange
    # uic_comm: the collectives go through libuic_hip's own RCCL communicator (uic_comm_*), torch.distributed (gloo) only
    # hands the unique id from rank 0 to the others
    exchange = UicCommExchange.from_torch_distributed() if uic_comm else None
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    lo, hi = GradientExchange().shard_images(cfg["n_img"])
    rows = slice(lo * cfg["S"], hi * cfg["S"])
    data = {k: I[k][rows].numpy() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}
    tr, losses = _train(cfg, W, data, STEPS, exchange, early_grads, allreduce, dtype, use_bn, next_data, half)
    from unpaired_image_captioning_amd import _lib
    if half:
        assert tr.sharded and tr.arena.g16 is not None and tr.arena.g16.dtype == torch.bfloat16
    if early_grads:      # group 1 of the overlapped exchange then also holds the embedding and att_lstm.weight_ih
        assert tr.i2t_model.engine.recurrence & _lib.REC_EARLY_GRADS
    assert tr.exchange.world_size == world
    if allreduce:
        assert not tr.sharded and len(tr.arena_splits) == 3 and 0 < tr.arena_splits[0] < tr.arena_splits[1] < tr.arena_splits[2] < tr.arena.numel
        if early_grads:
            names = list(tr.arena.offsets)
            g1 = [k for k in names if tr.arena_splits[0] <= tr.arena.offsets[k] < tr.arena_splits[1]]
            assert "embed.0.weight" in g1 and "core.att_lstm.weight_ih" in g1, g1
    else:
        a = tr.arena
        assert tr.sharded and a.world == world and a.rank == rank
        assert len(a.pieces) == (3 if early_grads else 4) and all(n % (64 * world) == 0 for _, n in a.pieces)
        assert (a.w16 is not None) == (dtype == "bf16")
        if early_grads:
            assert "embed.0.weight" in a.piece_names[1] and "core.att_lstm.weight_ih" in a.piece_names[1], a.piece_names
        else:
            assert a.piece_names[1] == ["embed.0.weight"] and "core.att_lstm.weight_ih" in a.piece_names[2] and tr.piece_groups == [0, 4, 3, None]
        # the small f32 tensors stay replicated; with a BatchNorm in att_embed its Linear too (the fold needs the f32 master)
        assert "logit.bias" in a.replicated and "core.attention.alpha_net.weight" in a.replicated
        assert ("att_embed.1.weight" in a.replicated) == bool(use_bn)
        if dtype == "bf16":
            # between steps a rank's f32 masters are current only inside its own shard: state_dict refuses until the collective ran
            with pytest.raises(RuntimeError, match="gather_masters"):
                tr.i2t_model.state_dict()
        tr.gather_masters()
    if rank == 0:
        torch.save({"sd": {k: v.cpu() for k, v in tr.i2t_model.state_dict().items()}, "losses": losses, "grads1": tr.grads_after_first_step},
                   os.path.join(out_dir, tag + ".pt"))
    if exchange is not None:
        exchange.close()
    dist.barrier()
    dist.destroy_process_group()


def _check_against_single_process(tmp_path, tag="dp2", early_grads=False):
    res = torch.load(os.path.join(str(tmp_path), tag + ".pt"))
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    data = {k: I[k].numpy() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}
    tr, losses = _train(cfg, W, data, STEPS, early_grads=early_grads)
    assert abs(losses[0] - float(Out["loss"])) < 1e-4
    if res.get("grads1") is not None:
        # the all-reduced gradient arena of the FIRST step against the reference's own gradients (golden) -- before Adam blurs it
        # (hygiene): f32, no dropout, so the two ranks' partial sums differ from one process by summation order only
        floor = 1e-3 * max(float(g.abs().max()) for g in G.values())
        for k, g in G.items():
            err = (res["grads1"][k] - g).abs().max().item() / max(g.abs().max().item(), floor)
            assert err < 2e-5, (k, err)
    for a, b in zip(res["losses"], losses):
        assert abs(a - b) < 1e-4, (res["losses"], losses)
    sd = tr.i2t_model.state_dict()
    for k, v in res["sd"].items():
        ref = sd[k].cpu()
        moved = (ref - W[k]).abs().max().item()
        assert moved > 0, k
        # alpha_net.bias has a mathematically zero gradient (softmax shift invariance): Adam normalises pure rounding
        # noise there, so its update is compared against the learning-rate scale instead of its own movement
        floor = 3 * 5e-3 * 1e-2 if k == "core.attention.alpha_net.bias" else 1e-7
        assert (v - ref).abs().max().item() <= 2e-2 * moved + floor, (k, (v - ref).abs().max().item(), moved)


def _spawn(tmp_path, **kw):
    world = 2
    args = dict(backend="gloo", uic_comm=False, early_grads=False, allreduce=False, dtype="f32", tag="dp2", use_bn=0, next_data=False, half=False)
    args.update(kw)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), args["backend"], args["uic_comm"], args["early_grads"], args["allreduce"],
                            args["dtype"], args["tag"], args["use_bn"], args["next_data"], args["half"]), nprocs=world, join=True)
    return torch.load(os.path.join(str(tmp_path), args["tag"] + ".pt"))


@pytest.mark.parametrize("allreduce", [False, True], ids=["sharded", "allreduce"])
def test_two_ranks_on_one_gpu_match_single_process(tmp_path, allreduce):
    _spawn(tmp_path, allreduce=allreduce)
    _check_against_single_process(tmp_path)


@pytest.mark.parametrize("allreduce", [False, True], ids=["sharded", "allreduce"])
def test_two_ranks_with_the_early_gradient_order_match_single_process(tmp_path, allreduce):
    """opt.early_grads (UIC_REC_EARLY_GRADS): the embedding gradient and att_lstm.weight_ih travel with gradient group 1 of the
    overlapped exchange -- their collective starts at ev_lstm, so they must really be final there."""
    _spawn(tmp_path, early_grads=True, allreduce=allreduce)
    _check_against_single_process(tmp_path, early_grads=True)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_sharded_exchange_leaves_the_same_bits_as_the_all_reduce(tmp_path, dtype):
    """reduce-scatter + Adam on the rank's slices + all-gather of the operand-dtype weights against
    all-reduce + Adam on everything, two ranks, three steps: every weight BIT-IDENTICAL (two ranks: a + b in both collectives;
    Adam is element-wise; bf16(master) is the same number whichever rank rounds it), and so are the losses.  The sharded run also
    carries the next batch's mask sum in the step's small all-reduce (Trainer.train(next_data=...), dict and callable)."""
    sh = _spawn(tmp_path, dtype=dtype, tag="sh", next_data=True)
    ar = _spawn(tmp_path, dtype=dtype, tag="ar", allreduce=True)
    assert sh["losses"] == ar["losses"], (sh["losses"], ar["losses"])
    for k, v in ar["sd"].items():
        assert torch.equal(sh["sd"][k], v), (k, (sh["sd"][k] - v).abs().max().item())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_bf16_gradient_exchange_stays_within_rounding_of_the_exact_one(tmp_path, dtype):
    """opt.bf16_gradient_exchange (off by default): the pieces that become final at the END of the step are reduce-scattered as
    bf16 -- half the bytes on the wire behind the join -- while the logit piece, which hides behind the BPTT loop, stays f32.  Each
    rank rounds its gradient once and the two-rank sum is rounded once more: relative 2^-8 per element, which Adam's m / sqrt(v)
    turns into a small fraction of a step for all but near-zero gradients.  Per tensor, 99.9 % of the weights must stay within 8 % (all of them: 2 % on
    average) of the distance the exact exchange moved the tensor in three steps, the losses agree to 1e-3 relative."""
    h = _spawn(tmp_path, dtype=dtype, tag="h", half=True)
    sh = _spawn(tmp_path, dtype=dtype, tag="sh")
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    assert h["losses"][0] == sh["losses"][0]                       # (the first forward pass saw the same weights)
    for a, b in zip(h["losses"], sh["losses"]):
        assert abs(a - b) <= 1e-3 * abs(b), (h["losses"], sh["losses"])
    differ = 0
    for k, v in sh["sd"].items():
        moved = (v.float() - W[k]).abs().max().item()
        err = (h["sd"][k].float() - v.float()).abs().max().item()
        floor = 3 * 5e-3 * 1e-2 if k == "core.attention.alpha_net.bias" else 1e-7
        # (Adam's m / sqrt(v) is +-1 for ANY gradient in the first steps: an element whose two per-rank gradients nearly cancel can
        # change sign when they are rounded and then moves the other way by a full step -- the maximum is only bounded by the
        # distance itself; 99.9 % of the elements and the mean are bounded tightly)
        d = (h["sd"][k].float() - v.float()).abs().flatten()
        assert err <= 1.0 * moved + floor, (k, err, moved)
        if d.numel() >= 1000:
            assert torch.quantile(d[:1000000], 0.999).item() <= 8e-2 * moved + floor, (k, torch.quantile(d[:1000000], 0.999).item(), moved)
        # (bias-sized tensors whose gradient is a cancellation -- ctx2att.bias, alpha_net -- are Adam-normalised noise in both runs)
        assert d.mean().item() <= (2e-2 if d.numel() >= 1000 else 0.15) * moved + floor, (k, d.mean().item(), moved)
        differ += int(err > 0)
    assert differ > 0                                              # (the option did something)


def test_sharded_exchange_with_batchnorm_in_att_embed(tmp_path):
    """use_bn = 1: att_embed's Linear is folded with the BatchNorm from its f32 master at every refresh, so the sharded exchange
    keeps that tensor replicated (all-reduced with the small tensors) -- same bits as the all-reduce path (bf16 operands).  The
    running statistics are per rank in both (DataParallel replica semantics)."""
    sh = _spawn(tmp_path, dtype="bf16", tag="sh", use_bn=1)
    ar = _spawn(tmp_path, dtype="bf16", tag="ar", allreduce=True, use_bn=1)
    assert sh["losses"] == ar["losses"], (sh["losses"], ar["losses"])
    for k, v in ar["sd"].items():
        assert torch.equal(sh["sd"][k], v), (k, (sh["sd"][k].float() - v.float()).abs().max().item())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: this box has %d" % torch.cuda.device_count())
def test_two_ranks_over_rccl_match_single_process(tmp_path):
    """The same step with the collectives on RCCL (backend "nccl"), one GPU per rank over xGMI: runs wherever two GPUs are
    visible (the 1-GPU boxes skip it)."""
    _spawn(tmp_path, backend="nccl")
    _check_against_single_process(tmp_path)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: this box has %d" % torch.cuda.device_count())
def test_two_ranks_over_uic_comm_match_single_process(tmp_path):
    """The same step with the collectives on libuic_hip's OWN RCCL communicator (uic_comm_init / uic_comm_allreduce with the
    by-value ncclUniqueId hand-over, UicCommExchange.from_torch_distributed): exercises the C-ABI communicator across ranks."""
    _spawn(tmp_path, uic_comm=True)
    _check_against_single_process(tmp_path)


def test_uic_comm_rejects_a_bad_unique_id():
    from unpaired_image_captioning_amd.parallel_exchange import UicCommExchange
    with pytest.raises(ValueError, match="128 bytes"):
        UicCommExchange(0, 1, b"short")


def test_uic_comm_single_rank_rccl():
    """uic_comm_* (include/uic_hip.h): RCCL through libuic_hip's own C-ABI entry points, no torch.distributed.  With one GPU per
    box only a world of 1 can be formed: unique id, communicator, in-place all-reduce (identity for one rank) of an f32 and a
    bf16 buffer on the current stream, destroy; and the exchange object the Trainer takes."""
    import torch
    from unpaired_image_captioning_amd.parallel_exchange import UicCommExchange
    uid = UicCommExchange.new_unique_id()
    assert len(uid) == 128 and any(uid)
    with UicCommExchange(0, 1, uid) as ex:
        g = torch.Generator(device="cuda").manual_seed(5)
        a = torch.randn(1 << 20, device="cuda", generator=g)
        b = torch.randn(4099, device="cuda", generator=g).bfloat16()
        a0, b0 = a.clone(), b.clone()
        ex._sum(a)
        ex._sum(b)
        torch.cuda.synchronize()
        assert torch.equal(a, a0) and torch.equal(b, b0)
        assert ex.world_size == 1 and ex.rank == 0 and ex.allreduce_sum(a) is a
    assert ex._comm is None                                     # closed by the context manager


# ---------------------------------------------------------------------------------------------------------------------------
# The self-critical step and the pivot NMT step across two ranks.  Same harness: two processes on
# cuda:0, gloo carrying the device tensors.

SC_STEPS = 3


def _sc_tokens(cfg, n_rows, seed=11):
    """A fixed `sampled` caption matrix [n_rows, L] with UNEQUAL lengths (rows end at a 0 after 1 .. L tokens): the two ranks' mask
    sums differ, so averaging their per-rank means would weight their rows differently from the whole batch's mean."""
    g = torch.Generator().manual_seed(seed)
    L = cfg["L"]
    tok = torch.randint(1, cfg["V"] + 1, (n_rows, L), generator=g)
    lens = torch.where(torch.arange(n_rows) < n_rows // 2, 1 + torch.arange(n_rows) % 2, L - torch.arange(n_rows) % 2)   # short first half
    for n in range(n_rows):
        tok[n, int(lens[n]):] = 0
    return tok


def _sc_reward(data, sampled, greedy):
    """A deterministic stand-in for CIDEr-D(sampled) - CIDEr-D(greedy): any function of the two token matrices, per row."""
    import numpy as np
    r = (sampled.sum(1) % 7).astype(np.float32) / 7.0 - (greedy.sum(1) % 5).astype(np.float32) / 5.0 + 0.3
    return np.repeat(r[:, None], sampled.shape[1], 1)


def _train_sc(cfg, W, data, forced, steps, exchange=None, allreduce=False):
    from unpaired_image_captioning_amd.trainer import Trainer
    tr = Trainer(_opt(cfg, allreduce=allreduce), exchange=exchange)
    tr.i2t_model.load_state_dict(W)
    tr.build_optimizer()
    tr.forced_samples = forced.cuda()
    losses = [tr.train_self_critical(data, reward_fn=_sc_reward) for _ in range(steps)]
    torch.cuda.synchronize()
    return tr, losses


def _sc_worker(rank, world, port, out_dir, allreduce=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unpaired_image_captioning_amd.parallel_exchange import GradientExchange
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    lo, hi = GradientExchange().shard_images(cfg["n_img"])
    rows = slice(lo * cfg["S"], hi * cfg["S"])
    data = {k: I[k][rows].numpy() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}
    forced = _sc_tokens(cfg, cfg["n_img"] * cfg["S"])[rows]
    tr, losses = _train_sc(cfg, W, data, forced, SC_STEPS, allreduce=allreduce)
    assert tr.exchange.world_size == world and tr.sharded == (not allreduce)
    if rank == 0:
        torch.save({"sd": {k: v.cpu() for k, v in tr.i2t_model.state_dict().items()}, "losses": losses}, os.path.join(out_dir, "sc2.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("allreduce", [False, True], ids=["sharded", "allreduce"])
def test_two_ranks_self_critical_step_matches_single_process(tmp_path, allreduce):
    """Trainer.train_self_critical on two ranks whose captions differ in length: RewardCriterion's denominator is the mask sum of
    the WHOLE batch (P/misc/criterion.py:117-122 on the gathered outputs, P/trainer.py:168-170), so the summed gradients -- and
    the reported loss -- must be those of one process on the whole batch.  (The sampled captions are pinned through
    Trainer.forced_samples and the reward is a fixed function of the tokens: nothing random is left.)"""
    world = 2
    mp.spawn(_sc_worker, args=(world, _free_port(), str(tmp_path), allreduce), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "sc2.pt"))
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    data = {k: I[k].numpy() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}
    forced = _sc_tokens(cfg, cfg["n_img"] * cfg["S"])
    half = forced.shape[0] // 2
    m = lambda t: int((t[:, :-1] > 0).sum()) + t.shape[0]
    assert m(forced[:half]) != m(forced[half:])          # the case the per-rank-mean average gets wrong
    tr, losses = _train_sc(cfg, W, data, forced, SC_STEPS)
    for a, b in zip(res["losses"], losses):
        assert abs(a - b) < 1e-4 * max(1.0, abs(b)), (res["losses"], losses)
    sd = tr.i2t_model.state_dict()
    for k, v in res["sd"].items():
        ref = sd[k].cpu()
        moved = (ref - W[k]).abs().max().item()
        floor = 3 * 5e-3 * 1e-2 if k == "core.attention.alpha_net.bias" else 1e-7
        assert (v - ref).abs().max().item() <= 2e-2 * moved + floor, (k, (v - ref).abs().max().item(), moved)


NMT_CFG = dict(layers=2, H=64, W=64, B=8, S=10, T=9, Vs=120, Vt=130)


def _nmt_trainer(tmp, exchange=None, allreduce=False):
    from test_gpu_nmt import make_opt
    from unpaired_image_captioning_amd.trainer import Trainer
    o = make_opt(NMT_CFG, "f32", dropout=0.0, seed=3)
    o.nmt_train_flag, o.i2t_train_flag, o.checkpoint_path = 1, 0, str(tmp)
    o.nmt_learning_rate, o.nmt_max_grad_norm, o.param_init = 5e-3, 0.5, 0.1      # (a clip that really bites: the norm is ~2)
    o.caption_model = None
    o.allreduce_exchange, o.allow_many_hw_queues = int(allreduce), 1
    tr = Trainer(o, exchange=exchange)
    torch.manual_seed(17)                                 # (param_init draws from torch's global generator: same weights everywhere)
    tr.build_nmt(NMT_CFG["Vs"], NMT_CFG["Vt"])
    return tr


def _nmt_worker(rank, world, port, out_dir, allreduce=False):
    import argparse as ap
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_nmt import synthetic
    from unpaired_image_captioning_amd import _lib
    I = synthetic(NMT_CFG, 9)
    cols = slice(rank, None, world)                       # DataParallel(dim=1) scatters columns; every shard stays length-sorted
    batch = ap.Namespace(src=I["src"][:, cols].contiguous().cuda(), tgt=I["tgt"][:, cols].contiguous().cuda(), lengths=I["lengths"][:, cols].contiguous())
    tr = _nmt_trainer(out_dir, allreduce=allreduce)
    tr.nmt_model.engine.recurrence = _lib.REC_FWD_CHAIN   # (two ranks on one GPU: per-step launches)
    assert tr.exchange.world_size == world and len(tr.optim.nmt_splits) == 2 and tr.optim.nmt_sharded == (not allreduce)
    if not allreduce:
        a = tr.optim.nmt_arena
        assert len(a.pieces) == 3 and a.w16 is None and not a.replicated and all(n % (64 * world) == 0 for _, n in a.pieces)
    names = list(tr.optim.nmt_arena.offsets)
    assert names[0].startswith("generator.") and names[-1].startswith("encoder.")
    losses = [tr.train_nmt(batch) for _ in range(3)]
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"sd": {k: v.cpu() for k, v in tr.nmt_model.state_dict().items()}, "losses": losses}, os.path.join(out_dir, "nmt2.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("allreduce", [False, True], ids=["sharded", "allreduce"])
def test_two_ranks_nmt_step_matches_single_process(tmp_path, allreduce):
    """Trainer.train_nmt on two ranks (column shards of the batch, as DataParallel(nmt_model, dim=1) scatters them,
    P/trainer.py:88): the pieces of the gradient arena travel as they become final (generator, decoder side on the
    communication stream behind uic_nmt_grad_ready_wait, the encoder's share last), the clip takes the norm of the SUMMED
    gradient, and loss and weights after three steps are those of one process on the whole batch."""
    import argparse as ap
    world = 2
    mp.spawn(_nmt_worker, args=(world, _free_port(), str(tmp_path), allreduce), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "nmt2.pt"))
    from test_gpu_nmt import synthetic
    I = synthetic(NMT_CFG, 9)
    batch = ap.Namespace(src=I["src"].cuda(), tgt=I["tgt"].cuda(), lengths=I["lengths"])
    tr = _nmt_trainer(tmp_path)
    W0 = {k: v.detach().cpu().clone() for k, v in tr.nmt_model.state_dict().items()}
    losses = [tr.train_nmt(batch) for _ in range(3)]
    for a, b in zip(res["losses"], losses):
        assert abs(a - b) < 2e-4 * abs(b), (res["losses"], losses)
    sd = tr.nmt_model.state_dict()
    for k, v in res["sd"].items():
        ref = sd[k].cpu()
        moved = (ref - W0[k]).abs().max().item()
        assert moved > 0, k
        assert (v - ref).abs().max().item() <= 2e-2 * moved + 1e-7, (k, (v - ref).abs().max().item(), moved)


def test_bench_two_rank_path_runs_end_to_end_on_one_gpu():
    """bench.py --gpus 2 in its functional mode (UIC_BENCH_SHARE_GPU=1: both ranks on this GPU, gloo, launch-chain recurrence --
    RCCL needs a GPU per rank): the N > 1 code of the benchmark -- rank-local seeds, the sharded exchange with the next batch's mask
    sum riding in the step's all-reduce, the "without exchange" leg, the per-rank gather, the untimed roofline steps with matched
    collectives -- must run and print its one line with the communication object.  (The line says it is not a measurement.)"""
    import json
    import subprocess
    env = dict(os.environ, UIC_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-f32",
                          "--no-cpu-baseline", "--long-run", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "NOT a measurement" in d["data"]
    c = d["communication"]
    assert c["exchange"].startswith("sharded: reduce-scatter of 4 gradient pieces") and len(c["ms_per_step_without_exchange_per_rank"]) == 2
    assert d["final_loss"] == d["final_loss"] and 5.0 < d["final_loss"] < 12.0
