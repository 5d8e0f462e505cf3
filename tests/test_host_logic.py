"""CPU-side checks (no GPU): the C ABI library loads and exports every symbol include/uic_hip.h declares,
the host-side mirror keeps the reference's contracts, and the product path fails loudly without a device."""
import argparse
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def _opt(**kw):
    base = dict(vocab_size=50, input_encoding_size=32, rnn_size=32, num_layers=1, drop_prob_lm=0.5, seq_length=6,
                fc_feat_size=64, att_feat_size=64, att_hid_size=32, use_bn=0, caption_model="topdown", compute_dtype="f32")
    base.update(kw)
    return argparse.Namespace(**base)


def test_library_exports_every_symbol_declared_in_header():
    from unpaired_image_captioning_amd import _lib
    from unpaired_image_captioning_amd.build import build
    build(verbose=False)
    header = open(os.path.join(ROOT, "include", "uic_hip.h")).read()
    declared = set(re.findall(r"\b(uic_[a-z0-9_]+)\s*\(", header))
    declared -= {"uic_topdown_dims", "uic_topdown_weights", "uic_topdown_batch"}
    assert len(declared) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "header declares %s but libuic_hip.so does not export it" % name
    assert set(_lib.EXPORTS) == declared, set(_lib.EXPORTS) ^ declared
    assert _lib.load().uic_version() >= 100


def test_workspace_size_queries_and_argument_errors():
    from unpaired_image_captioning_amd import _lib
    lib = _lib.load()
    d = _lib.Dims(N=640, R=36, D=2048, Dfc=2048, H=512, E=512, A=512, V1=9488, T=17, dtype=1, drop_p=0.5)
    ws_bf16 = lib.uic_topdown_workspace_bytes(C.byref(d))
    d.dtype = 0
    ws_f32 = lib.uic_topdown_workspace_bytes(C.byref(d))
    assert 1 << 30 < ws_bf16 < ws_f32 < 4 << 30 and ws_bf16 % 256 == 0
    assert lib.uic_topdown_derived_bytes(C.byref(d)) > 0
    d.H = 510                                         # not a multiple of 8 -> rejected, with a message
    assert lib.uic_topdown_workspace_bytes(C.byref(d)) == 0
    assert b"multiples of 8" in lib.uic_last_error_string()
    d.H = 512
    d.dtype = 9
    assert lib.uic_topdown_workspace_bytes(C.byref(d)) == 0
    # null pointers are argument errors (negative), never a crash
    assert lib.uic_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) < 0
    # named workspace pieces (tools read time stamps through these): inside the workspace, NULL for unknown names / bad dims
    nd = _lib.NmtDims(B=64, S=30, T=32, H=512, W=512, layers=2, Vs=50004, Vt=50004, dtype=1, drop_p=0.3)
    total = lib.uic_nmt_workspace_bytes(C.byref(nd))
    base = 1 << 40                                    # (an address, never dereferenced: the call only lays the workspace out)
    for name in (b"dec_bwd_dbg", b"dec_fwd_dbg", b"d_cq", b"dscore", b"d_pre"):
        ptr = lib.uic_nmt_workspace_ptr(C.byref(nd), C.c_void_p(base), name)
        assert ptr is not None and base <= ptr < base + total and ptr % 256 == 0, name
    assert lib.uic_nmt_workspace_ptr(C.byref(nd), C.c_void_p(base), b"no such piece") is None
    assert lib.uic_nmt_workspace_ptr(C.byref(nd), None, b"d_cq") is None
    td = _lib.Dims(N=640, R=36, D=2048, Dfc=2048, H=512, E=512, A=512, V1=9488, T=17, dtype=1, drop_p=0.5)
    ptr = lib.uic_topdown_workspace_ptr(C.byref(td), C.c_void_p(base), b"rnn_dbg")
    assert ptr is not None and base <= ptr < base + ws_bf16
    assert lib.uic_topdown_workspace_ptr(C.byref(td), C.c_void_p(base), b"no such piece") is None


def test_weight_struct_matches_reference_state_dict_order():
    from unpaired_image_captioning_amd import _lib, models
    cfg, W, I, Out, G, X = load_golden("topdown_tiny")
    model = models.setup(_opt())
    assert [k for _, k in _lib.WEIGHT_FIELDS] == list(W.keys()) == list(model.state_dict().keys())
    model.load_state_dict(W)
    for k, v in model.state_dict().items():
        assert torch.equal(v, W[k])
    assert [f for f, _ in _lib.Weights._fields_] == [f for f, _ in _lib.WEIGHT_FIELDS] + [f for f, _, _ in _lib.BN_FIELDS] + \
        ["logit_h_w", "logit_h_b"]
    # logit_layers > 1: `logit` becomes a Sequential, hidden blocks at logit.{3l}, the vocabulary layer last (AttModel.py:90-91)
    m3 = models.setup(_opt(logit_layers=3))
    assert [k for _, k, _ in _lib.weight_fields(0, 3)] == list(m3.state_dict().keys())
    assert [k for k in m3.state_dict() if k.startswith("logit.")] == ["logit.0.weight", "logit.0.bias", "logit.3.weight", "logit.3.bias",
                                                                     "logit.6.weight", "logit.6.bias"]
    assert tuple(m3.state_dict()["logit.6.weight"].shape) == (51, 32) and tuple(m3.state_dict()["logit.3.weight"].shape) == (32, 32)


def test_same_seed_gives_reference_initialisation():
    """The parameter-owning module tree is built in the reference's order, so torch.manual_seed(s) reproduces
    the reference model's initial weights (golden `w::` entries were drawn with seed 11)."""
    from unpaired_image_captioning_amd import models
    cfg, W, I, Out, G, X = load_golden("topdown_tiny")
    torch.manual_seed(11)
    model = models.setup(_opt(drop_prob_lm=0.0))
    for k, v in model.state_dict().items():
        assert torch.equal(v, W[k]), k


def test_no_cpu_fallback():
    from unpaired_image_captioning_amd import models
    cfg, W, I, Out, G, X = load_golden("topdown_tiny")
    model = models.setup(_opt(drop_prob_lm=0.0))
    model.load_state_dict(W)
    with pytest.raises(RuntimeError, match="device tensors"):
        model(I["fc_feats"], None, I["att_feats"], I["labels"], I["att_masks"])
    with pytest.raises(RuntimeError, match="device tensors"):
        model(I["fc_feats"], None, I["att_feats"], I["att_masks"], opt={"sample_max": 1}, mode="sample")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No libuic_hip.so -> RuntimeError naming the build command; never a silent CPU / eager path."""
    from unpaired_image_captioning_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libuic_hip.so"))
    with pytest.raises(RuntimeError, match="not built.*no CPU fallback"):
        _lib.load()
    with pytest.raises(RuntimeError, match="not built"):
        from unpaired_image_captioning_amd.topdown_engine import TopDownEngine
        TopDownEngine(dict(V1=51, E=32, H=32, A=32, D=64, Dfc=64))


def test_product_code_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under unpaired_image_captioning_amd/ may import or execute it, and the only
    top-level users are bench.py's cpu_baseline leg and __graft_entry__.smoke()."""
    import ast
    pkg = os.path.join(ROOT, "unpaired_image_captioning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n == "oracle" or n.startswith("oracle.") for n in names), (f, names)
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    users = [fn.name for fn in ast.walk(tree) if isinstance(fn, ast.FunctionDef)
             and any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n) for n in ast.walk(fn))]
    assert users == ["cpu_baseline"], users
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n)]
    assert not top


def test_unsupported_options_raise():
    from unpaired_image_captioning_amd import models
    with pytest.raises(ValueError):
        models.setup(_opt(use_bn=3))
    with pytest.raises(NotImplementedError):
        models.setup(_opt(logit_layers=5))
    with pytest.raises(Exception, match="not supported"):
        models.setup(_opt(caption_model="transformer"))


def test_early_break_step_count_matches_reference_rule():
    from unpaired_image_captioning_amd.models.AttModel import AttModel
    from unpaired_image_captioning_amd.trainer import _steps_from_host_labels
    for name in ("topdown_tiny", "topdown_tiny_earlybreak", "topdown_odd"):
        cfg, W, I, Out, G, X = load_golden(name)
        labels = I["labels"]
        ref_zero = (Out["logprobs"].abs().sum((0, 2)) == 0).nonzero().view(-1)
        expect = int(ref_zero[0]) if ref_zero.numel() else labels.shape[1] - 1
        assert AttModel._steps_to_run(labels) == expect
        assert _steps_from_host_labels(labels.numpy()) == expect


def test_lr_schedule_matches_reference_formula():
    from unpaired_image_captioning_amd.trainer import Trainer
    tr = Trainer(_opt(i2t_learning_rate=4e-4, i2t_learning_rate_decay_start=0, i2t_learning_rate_decay_every=3,
                      i2t_learning_rate_decay_rate=0.8))
    for epoch in range(0, 12):
        tr.update_LearningRate(epoch)
        frac = (epoch - 0) // 3 if epoch > 0 else 0        # P/misc/optimizer.py:116-120
        assert tr.i2t_current_lr == pytest.approx(4e-4 * 0.8 ** frac)


def test_synthetic_batch_layout_cpu():
    from unpaired_image_captioning_amd.synthetic import synthetic_batch
    b = synthetic_batch(4, 5, 36, 64, 100, 16, seed=1, device="cpu", ragged_regions=True)
    assert b["att_feats"].shape == (20, 36, 64) and b["labels"].shape == (20, 18) and b["masks"].shape == (20, 18)
    assert (b["labels"][:, 0] == 0).all() and (b["labels"][:, -1] == 0).all()
    nz = (b["labels"] != 0).sum(1) + 2
    assert torch.equal(b["masks"].sum(1).long(), nz)                    # P/misc/dataloader/dataloader.py:283-286
    assert torch.equal(b["att_feats"][0], b["att_feats"][4]) and not torch.equal(b["att_feats"][0], b["att_feats"][5])
    cnt = b["att_masks"].sum(1)
    assert (cnt[:-1] >= cnt[1:]).all()                                   # sorted by region count, as the loader does


def test_optim_schedules_match_reference_golden():
    """Optim.update_LearningRate('i2t'|'nmt') and update_ScheduledSampling_prob over epochs 0..13 against the values
    the reference's own Optim produced (tests/golden/nmt_optim_clip.npz, P/misc/optimizer.py:108-131)."""
    import argparse
    import os
    import numpy as np
    from conftest import GOLDEN
    from unpaired_image_captioning_amd.misc.optimizer import Optim
    z = np.load(os.path.join(GOLDEN, "nmt_optim_clip.npz"))
    opt = argparse.Namespace(
        i2t_train_flag=1, i2t_learning_rate=4e-4, i2t_learning_rate_decay_start=0, i2t_learning_rate_decay_every=3,
        i2t_learning_rate_decay_rate=0.8, nmt_train_flag=1, nmt_learning_rate=1e-3, nmt_learning_rate_decay_start=8,
        nmt_learning_rate_decay_rate=0.5, scheduled_sampling_start=0, scheduled_sampling_increase_every=5,
        scheduled_sampling_increase_prob=0.05, scheduled_sampling_max_prob=0.25, rnn_size=32)
    o = Optim(opt)
    holder = argparse.Namespace(ss_prob=0.0)
    nmt, i2t, ss = [], [], []
    for epoch in range(14):
        o.update_LearningRate("nmt", epoch)
        o.update_LearningRate("i2t", epoch)
        o.update_ScheduledSampling_prob(opt, epoch, holder)
        nmt.append(o.nmt_current_lr); i2t.append(o.i2t_current_lr); ss.append(holder.ss_prob)
    np.testing.assert_allclose(nmt, z["out::sched_nmt"], rtol=1e-12)
    np.testing.assert_allclose(i2t, z["out::sched_i2t"], rtol=1e-12)
    np.testing.assert_allclose(ss, z["out::sched_ss"], rtol=1e-12)
    assert len(set(i2t)) > 3 and len(set(nmt)) == 2 and max(ss) == 0.1   # the schedules did move


@pytest.mark.parametrize("name,use_bn", [("topdown_tiny_bn1_eval", 1), ("topdown_tiny_bn2_train", 2)])
def test_use_bn_state_dict_matches_reference(name, use_bn):
    """opt.use_bn (reference default 1, P/opts.py:52): att_embed gets BatchNorm1d at index 0 (and 4), the Linear moves
    to index 1, and the checkpoint keys / order / shapes are the reference's (golden `w::` from its own module)."""
    from unpaired_image_captioning_amd import _lib, models
    cfg, W, I, Out, G, X = load_golden(name)
    assert cfg["use_bn"] == use_bn
    model = models.setup(_opt(use_bn=use_bn))
    sd = model.state_dict()
    assert list(sd.keys()) == list(W.keys())
    for k in W:
        assert tuple(sd[k].shape) == tuple(W[k].shape), k
    model.load_state_dict(W)
    assert model.param_names == [k for k, _ in model.named_parameters()]
    assert set(G) == set(model.param_names)
    fields = _lib.weight_fields(use_bn)
    assert [k for _, k, p in fields if p] == model.param_names
    assert all(k in sd for _, k, _ in fields)


def test_decode_sequence_and_if_use_att():
    """P/misc/utils.py:42-66."""
    import torch
    from unpaired_image_captioning_amd.misc import utils
    vocab = {"1": "a", "2": "b", "3": "c"}
    seq = torch.tensor([[1, 2, 0, 3], [3, 3, 3, 3], [0, 1, 1, 1]])
    assert utils.decode_sequence(vocab, seq) == ["a b", "c c c c", ""]
    assert utils.if_use_att("topdown") and not utils.if_use_att("fc")


def test_data_parallel_ranks_start_from_different_dropout_seeds():
    """Dropout / sampling noise is a hash of (seed, site, LOCAL row index): rank r of a data-parallel job must not reuse
    rank 0's stream, or every shard draws the same masks (Trainer._mix_rank_into_seed)."""
    import argparse
    from unpaired_image_captioning_amd.trainer import Trainer

    class FakeExchange(object):
        def __init__(self, rank):
            self.rank, self.world_size = rank, 4

    class M(object):
        _seed_counter = 1234

    seeds = []
    for r in range(4):
        m = M()
        Trainer._mix_rank_into_seed(m, FakeExchange(r))
        seeds.append(m._seed_counter)
    assert seeds[0] == 1234 and len(set(seeds)) == 4 and all(0 <= s_ <= 0x7FFFFFFF for s_ in seeds)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_f32a_gemm_build_has_no_compiler_touch_of_in_flight_registers():
    """tools/audit_f32a_asm.py: the f32-A ping-pong GEMM loads its A units with inline-asm
    global_load_dwordx4 into C++ variables and waits for them by a hand-counted vmcnt four phases later; hipcc counts an asm
    statement's VGPR destination as written when the statement ends.  The audit compiles csrc/gemm_pp.hip for gfx950 (no GPU
    needed), walks the control-flow graph of every f32-A instantiation and fails on any spill, or any instruction outside an asm
    block that reads or writes a destination between its load and the conversion that consumes it."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_f32a_asm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 problems" in r.stdout and r.stdout.count("asm loads") >= 2, r.stdout
