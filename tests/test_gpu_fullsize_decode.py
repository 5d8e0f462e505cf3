"""Full-size parity of the decode and self-critical paths (BASELINE configs[1] widths: R = 36, D = 2048, H = E = A = 512,
vocabulary 9487 + 1, 16 sampled tokens) against the CPU oracle -- the H = 512 / V = 9488 sampling kernels (arg-max,
inverse-CDF segment scan), the training-layout sampling pass and the backward that starts from its kept forward:
  * greedy AttModel._sample (P/models/AttModel.py:198-253) at 128 images: token ids BIT-EXACT in f32; bf16: the device's own
    tokens replayed through the oracle, log-probs within 1e-2;
  * multinomial pass at 128 x 5 rows: the oracle re-draws every token from ITS distribution with the device's own uniform
    numbers (the counter hash of csrc/pointwise.hip restated below) and must pick the same ids; log-probs as above;
  * Trainer-level self-critical step (P/trainer.py:166-171) at 64 x 5 rows: sampled ids, log-probs, loss and every gradient
    tensor against the oracle fed with the device's tokens and dropout masks."""
import numpy as np
import pytest
import torch

from conftest import poison_workspaces
from oracle import topdown as O
from test_gpu_topdown import build_model

pytestmark = pytest.mark.gpu

V, E, H, A, D, L, R = 9487, 512, 512, 512, 2048, 16, 36
CFG = dict(V=V, E=E, H=H, A=A, D=D, L=L)
LOGP_TOL = {"f32": 1e-3, "bf16": 1e-2}


def uniform_draws(seed, N, steps):
    """u[n, t] of sample_step_kernel's inverse-CDF draw (csrc/pointwise.hip): a counter hash of (seed, step, row)."""
    n = np.arange(N, dtype=np.uint64)[:, None]
    t = np.arange(steps, dtype=np.uint64)[None, :]
    M = np.uint64(0xFFFFFFFF)
    x = ((n * np.uint64(0x9E3779B1)) & M) ^ ((np.uint64(seed) + t * np.uint64(0x85EBCA77)) & M)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M
    x ^= x >> np.uint64(16)
    return ((x >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0))


def oracle_threads():
    nt = torch.get_num_threads()
    torch.set_num_threads(min(16, nt))
    return nt


@pytest.fixture(scope="module")
def case():
    """Two weight sets on one batch.  greedy: a peaked word distribution (logit.weight x 25) in which, with image features of
    very different magnitude, the end token wins at once for the weakest images and never for the rest.  sampling: the
    near-uniform distribution of random weights with the end token at ~5 % per step, so sampled captions end at every length."""
    W = O.init_weights(V + 1, E, H, A, D, D, seed=23)
    Wg = {k: v.clone() for k, v in W.items()}
    Wg["logit.weight"] *= 25.0
    Wg["logit.bias"][0] += 0.7
    Ws = {k: v.clone() for k, v in W.items()}
    Ws["logit.bias"][0] += 6.2
    b = O.synthetic_batch(128, 5, R, D, V, L, seed=99, ragged_regions=True)
    return Wg, Ws, b


def alive_mask(seq):
    """positions a caption row is still being decoded at: step 0 and every step after a non-zero token (the step that draws the
    end token included).  Past that the reference records the log-prob of whatever `it` was drawn before masking it out
    (P/models/AttModel.py:229-247), which a forced replay cannot know."""
    return torch.cat([torch.ones(seq.shape[0], 1, dtype=torch.bool), seq[:, :-1] > 0], 1)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_greedy_decode_full_size(case, dtype):
    W, _, b = case
    idx = torch.arange(128) * 5
    sc = 1 + 2.0 * (torch.arange(128) % 8).float()
    fc, att, am = b["fc_feats"][idx] * sc[:, None], b["att_feats"][idx] * sc[:, None, None], b["att_masks"][idx]
    model = build_model(CFG, W, dtype).eval()
    with torch.no_grad():
        seq, lp = model(fc.cuda(), None, att.cuda(), am.cuda(), opt={"sample_max": 1}, mode="sample")
    seq, lp = seq.cpu(), lp.cpu()
    nt = oracle_threads()
    if dtype == "f32":
        seq_o, lp_o = O.sample(W, fc, att, am, L)
    else:       # near-tied words may legitimately swap under bf16 rounding: score the device's own tokens
        seq_o, lp_o = O.sample(W, fc, att, am, L, sample_max=0, forced_tokens=seq)
    torch.set_num_threads(nt)
    lens = (seq_o > 0).sum(1)
    assert int(lens.min()) < 3 and int(lens.max()) == L and len(set(lens.tolist())) >= 2   # captions that end at once and captions that never end
    assert torch.equal(seq, seq_o)                                        # f32: bit-exact greedy ids; bf16: same end-of-caption bookkeeping
    live = alive_mask(seq) if dtype == "bf16" else torch.ones_like(seq, dtype=torch.bool)
    err = (lp - lp_o)[live].abs().max().item()
    assert err < LOGP_TOL[dtype]
    print("greedy full size %s: %d rows, caption lengths %s, max |logp - oracle| %.2e" %
          (dtype, seq.shape[0], sorted(set(lens.tolist())), err))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_multinomial_pass_full_size_redrawn_by_the_oracle(case, dtype):
    _, W, b = case
    fc, att, am = b["fc_feats"], b["att_feats"], b["att_masks"]               # 640 rows (features replicated per caption row)
    model = build_model(CFG, W, dtype).eval()
    with torch.no_grad():
        seq, lp = model(fc.cuda(), None, att.cuda(), am.cuda(), opt={"sample_max": 0, "temperature": 1.0}, mode="sample")
    seed = model._seed_counter
    seq, lp = seq.cpu(), lp.cpu()
    N = seq.shape[0]
    u = torch.from_numpy(uniform_draws(seed, N, L))
    # the oracle walks the device's history (forced tokens) and, at every step, draws from ITS OWN distribution with the device's
    # uniform number: first word whose cumulative mass exceeds u * total (the kernel's rule)
    nt = oracle_threads()
    Hh = O.logit_final_weight(W).shape[1]
    fcp, attp, p_att, masks = O.prepare_feature(W, fc, att, am, None, 0, False)
    state = (torch.zeros(2, N, Hh), torch.zeros(2, N, Hh))
    it = torch.zeros(N, dtype=torch.long)
    alive = torch.ones(N, dtype=torch.bool)
    same, total, worst_lp, near = 0, 0, 0.0, 0
    for t in range(L):
        logp, state, _ = O.logprobs_step(W, it, fcp, attp, p_att, masks, state)
        cum = logp.double().exp().cumsum(1)
        target = u[:, t].double() * cum[:, -1]
        pick = (cum > target[:, None]).double().argmax(1)
        dev = seq[:, t]
        rows = alive.nonzero().flatten()
        raw = torch.where(dev[rows] > 0, dev[rows], torch.zeros_like(dev[rows]))   # a live row that drew 0 ends here
        agree = pick[rows] == raw
        # a draw that lands within f32 rounding of a word boundary may fall on either side of it
        c = cum[rows]
        lo = torch.where(raw > 0, c.gather(1, (raw - 1).clamp(min=0)[:, None])[:, 0], torch.zeros(len(rows), dtype=torch.float64))
        hi = c.gather(1, raw[:, None])[:, 0]
        tg = target[rows]
        # (bf16: the device draws from ITS distribution, whose log-probs are within 1e-2 of the oracle's -- the cumulative mass
        # at a word boundary then differs by up to that fraction of the mass below it)
        slack = (2e-6 if dtype == "f32" else 1.5e-2) * c[:, -1]
        edge = ((tg > lo - slack) & (tg < hi + slack))
        assert bool((agree | edge).all()), (t, int((~(agree | edge)).sum()))
        same += int(agree.sum()); total += len(rows); near += int((~agree & edge).sum())
        worst_lp = max(worst_lp, float((lp[rows, t] - logp[rows, raw]).abs().max()))
        alive = alive & (dev > 0)
        it = dev * alive.long()
        assert bool((seq[~alive, t + 1:] == 0).all()) if t + 1 < L else True       # finished rows stay finished
    torch.set_num_threads(nt)
    assert total > 0.3 * N * L, total                                    # (captions end at ~5 % per step)
    # (bf16: a word holds ~1e-4 of the mass, the same order as the cumulative rounding difference -- a draw often lands on a
    # neighbouring word; what is asserted above is that it always lands within the rounding slack of the oracle's boundary)
    assert same >= (total - max(2, total // 2000) if dtype == "f32" else 0.3 * total), (same, total)
    lens = (seq > 0).sum(1)
    assert len(set(lens.tolist())) > 8, sorted(set(lens.tolist()))         # sampled captions of many lengths
    assert worst_lp < LOGP_TOL[dtype]
    print("multinomial full size %s: %d of %d draws re-drawn identically by the oracle (%d at a rounding boundary), max |logp - oracle| %.2e" %
          (dtype, same, total, near, worst_lp))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_self_critical_step_full_size_vs_oracle(case, dtype):
    """64 images x 5 captions at the real widths: train-mode sampling pass in the training layout (kept forward), reward
    criterion, backward from the kept forward; the oracle replays the device's tokens and dropout masks."""
    from unpaired_image_captioning_amd import _lib as Lb
    from unpaired_image_captioning_amd.misc.criterion import RewardCriterion
    _, W, b = case
    rows = slice(0, 64 * 5)
    fc, att, am = b["fc_feats"][rows], b["att_feats"][rows], b["att_masks"][rows]
    model = build_model(CFG, W, dtype, drop=0.5)
    model.train()
    assert getattr(model, "scst_keep_forward", True)
    seq, lp = model(fc.cuda(), None, att.cuda(), am.cuda(), opt={"sample_max": 0}, mode="sample")
    seed = model._seed_counter
    g = torch.Generator().manual_seed(5)
    reward = torch.randn(seq.shape, generator=g)
    loss = RewardCriterion()(lp, seq, reward.cuda())
    loss.backward()
    lib = Lb.load()
    N = fc.shape[0]

    def mask(n, site):
        out = torch.empty(n, device="cuda")
        Lb.check(lib.uic_dropout_mask(Lb.ptr(out), n, 0.5, seed, site, 0, Lb.stream()))
        return out.cpu()

    drop = dict(fc=mask(N * H, Lb.SITE_FC).view(N, H), att=mask(N * R * H, Lb.SITE_ATT).view(N, R, H),
                embed=mask(L * N * E, Lb.SITE_EMBED).view(L, N, E),
                out=torch.stack([mask(N * H, Lb.SITE_OUT0 + t).view(N, H) for t in range(L)]))
    nt = oracle_threads()
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    seq_o, lp_o = O.sample(Wg, fc, att, am, L, sample_max=0, forced_tokens=seq.cpu(), drop=drop)
    loss_o = O.reward_criterion(lp_o, seq_o, reward)
    loss_o.backward()
    torch.set_num_threads(nt)
    assert torch.equal(seq_o, seq.cpu())
    live = alive_mask(seq.cpu())
    assert len(set((seq.cpu() > 0).sum(1).tolist())) > 8
    assert (lp.detach().cpu() - lp_o.detach())[live].abs().max().item() < LOGP_TOL[dtype]
    assert abs(loss.item() - loss_o.item()) < LOGP_TOL[dtype]
    ref = {k: v.grad for k, v in Wg.items() if v.grad is not None}
    got = {k: p.grad for k, p in model.named_parameters()}
    floor = 1e-3 * max(float(v.norm()) for v in ref.values())
    worst, worst_k = 0.0, ""
    for k, r in ref.items():
        err = ((got[k].float().cpu().double() - r.double()).norm() / max(r.double().norm().item(), floor)).item()
        if err > worst:
            worst, worst_k = err, k
        # measured: f32 4e-6; bf16 3.3e-2 on fc_embed.0.weight (its gradient passes through every decode step's att_lstm), ~1e-2 elsewhere
        assert err < {"f32": 2e-5, "bf16": 5e-2}[dtype], (k, err)
    print("self-critical full size %s: loss %.6f (oracle %.6f), worst per-tensor L2 gradient error %.3e (%s)" %
          (dtype, loss.item(), loss_o.item(), worst, worst_k))


@pytest.mark.parametrize("n_rows", [640, 645, 24, 1, 81, 1283])
def test_persistent_decode_launch_against_the_launch_chain(case, n_rows):
    """bf16: AttModel._sample as ONE persistent launch (rnn_persist.hip's decode mode: embedding, recurrence, logit layer and
    the greedy / multinomial choice inside the launch) against the per-step chain of six launches -- 640 rows (one slab of 8
    groups x 80), 645 (a second slab of 5 rows: groups without rows), 24 (3 rows per group), 1, 81 (11 rows per group, one with 4)
    and 1283 (three slabs).  The two differ only by
    summation order in bf16, so: the chain's tokens replayed through the persistent launch give the same ids and log-probs
    within 1e-2 (eval and train mode / kept forward); greedy decodes mostly identical captions; the placement-independent SAFE
    protocol gives the persistent launch's results bit for bit; and the sampled captions end at every length (the reference's
    finished-row bookkeeping, with its early break, included)."""
    from unpaired_image_captioning_amd import _lib as Lb
    Wg, Ws, b = case
    reps = (n_rows + 639) // 640
    rows = torch.arange(n_rows) % 640
    fc, att, am = b["fc_feats"][rows].cuda(), b["att_feats"][rows].cuda(), b["att_masks"][rows].cuda()
    for W, sample_max, training in ((Wg, 1, False), (Ws, 0, False), (Ws, 0, True)):
        model = build_model(CFG, W, "bf16", drop=0.5)
        eng = poison_workspaces(model.engine)        # (stale values of the previous, identical pass must not help anybody)
        pd = {k: v.detach() for k, v in model.param_dict().items()}
        keep = training

        def run(flags, forced=None):
            eng.recurrence = flags
            try:
                out = eng.sample(pd, fc, att, am, L, sample_max=sample_max, seed=4242, forced=forced, training=training, keep_forward=keep)
            finally:
                eng.recurrence = 0
            if keep:
                eng.release(out[2])
            return out[0].cpu(), out[1].cpu()

        before = Lb.persistent_status()
        seq_c, lp_c = run(Lb.REC_FWD_CHAIN)
        mid = Lb.persistent_status()
        assert (mid[1], mid[2]) == (before[1], before[2])                   # the chain launched no persistent kernel
        seq_p, lp_p = run(0)
        after = Lb.persistent_status()
        assert after[0] == 0 and after[1] - mid[1] == reps and after[2] == mid[2]
        seq_s, lp_s = run(Lb.REC_SAFE)
        assert torch.equal(seq_s, seq_p) and torch.equal(lp_s, lp_p)
        assert Lb.persistent_status()[2] - after[2] == reps
        live = alive_mask(seq_c)
        if sample_max:
            same_rows = (seq_c == seq_p).all(1).float().mean().item()
            assert n_rows < 64 or same_rows > 0.9, same_rows
            both = (seq_c == seq_p).cumprod(1).bool() & live                # positions with an identical history
            assert (lp_c - lp_p)[both].abs().max().item() < LOGP_TOL["bf16"]
        else:
            seq_f, lp_f = run(0, forced=seq_c.cuda())
            assert torch.equal(seq_f, seq_c)
            assert (lp_f - lp_c)[live].abs().max().item() < LOGP_TOL["bf16"]
            same_tok = (seq_c == seq_p).float().mean().item()
            assert n_rows < 64 or same_tok > 0.5, same_tok                                  # a draw at a rounding boundary changes the rest of its row
        lens = (seq_p > 0).sum(1)
        if n_rows >= 640 and W is not Wg:
            assert len(set(lens.tolist())) > 8
        # entries behind a finished row are zero, and log-probs behind the step at which EVERY row had finished too
        fin = ~alive_mask(seq_p)
        assert (seq_p[fin] == 0).all()
        t_dead = int(lens.max().item()) + 1 if (lens < L).all() else L
        if t_dead < L:
            assert (lp_p[:, t_dead:] == 0).all()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_self_critical_step_overlapped_baseline_equals_serial_when_both_passes_share_a_workspace_key(case, dtype):
    """S == 1 (features replicated as the reference ships them): the sampling pass and the greedy baseline of
    Trainer.train_self_critical have the SAME workspace key and run on two streams -- the pool must not hand the baseline's
    workspace to the sampling pass while the baseline still runs.  Three steps with the baseline beside the sampling pass
    (default), with the two passes one after the other (serial_baseline) and -- bf16 -- with each pass as one persistent decode
    launch: every loss and every weight after three Adam steps BIT FOR BIT in the first two (same kernels, same seeds, and no
    floating-point atomics anywhere in the step -- the embedding gradient is a stable counting sort with one owner per table
    row, csrc/pointwise.hip; f32 everywhere the chain).  What the test protects: P/trainer.py:166-171, P/misc/rewards.py:42-47."""
    from unpaired_image_captioning_amd.trainer import Trainer
    from test_gpu_topdown import make_opt
    _, Ws, b = case
    rows = slice(0, 48 * 5)
    data = {k: b[k][rows].numpy() for k in ("fc_feats", "att_feats", "att_masks")}
    data["labels"] = np.zeros((48 * 5, L + 2), dtype=np.int64)
    data["masks"] = np.zeros((48 * 5, L + 2), dtype=np.float32)

    def reward_fn(data, sampled, greedy):
        r = np.where(sampled[:, :1] % 2 == 0, 1.0, -1.0) - np.where(greedy[:, :1] % 2 == 0, 0.5, -0.5)
        return np.repeat(r, sampled.shape[1], 1).astype(np.float32)

    res = {}
    for mode in ("overlap", "serial") + (("persistent",) if dtype == "bf16" else ()):
        opt = make_opt(CFG, dtype, drop=0.5, seed=3)
        opt.i2t_learning_rate = 1e-3
        opt.seq_per_img = 1                              # S == 1: both passes see 240 feature rows
        opt.ship_replicated_features = 1
        tr = Trainer(opt)
        tr.i2t_model.load_state_dict(Ws)
        tr.build_optimizer()
        tr.serial_baseline = mode == "serial"
        tr.persistent_decode = mode == "persistent"
        losses = [tr.train_self_critical(data, reward_fn) for _ in range(3)]
        res[mode] = (losses, {k: v.detach().cpu().clone() for k, v in tr.i2t_model.state_dict().items()})
    l0, w0 = res["serial"]
    l1, w1 = res["overlap"]
    assert l1 == l0, (l1, l0)
    for k in w0:
        assert torch.equal(w1[k], w0[k]), (k, (w1[k].double() - w0[k].double()).abs().max().item())
    if "persistent" in res:      # another summation order in the decode launch: the sampled captions may differ at rounding boundaries
        lp, wp = res["persistent"]
        assert all(np.isfinite(lp)) and abs(lp[0] - l0[0]) < 0.5


@pytest.mark.parametrize("regions,masked", [(7, True), (36, False), (1, True)])
def test_persistent_decode_launch_other_region_counts(case, regions, masked):
    """The same comparison with 7 and 1 regions per image (att_masks ragged inside them) and with 36 unmasked ones: the chain's
    multinomial tokens replayed through the persistent launch, and greedy decoding both ways."""
    from unpaired_image_captioning_amd import _lib as Lb
    Wg, Ws, b = case
    n = 160
    fc, att = b["fc_feats"][:n].cuda(), b["att_feats"][:n, :regions].contiguous().cuda()
    am = b["att_masks"][:n, :regions].contiguous().clone()
    am[:, 0] = 1.0                                    # (every image keeps at least one region)
    am = am.cuda() if masked else None
    for W, sample_max in ((Wg, 1), (Ws, 0)):
        model = build_model(CFG, W, "bf16", drop=0.5)
        eng = model.engine
        pd = {k: v.detach() for k, v in model.param_dict().items()}

        def run(flags, forced=None):
            eng.recurrence = flags
            try:
                out = eng.sample(pd, fc, att, am, L, sample_max=sample_max, seed=99, forced=forced, training=not sample_max)
            finally:
                eng.recurrence = 0
            return out[0].cpu(), out[1].cpu()

        before = Lb.persistent_status()
        seq_c, lp_c = run(Lb.REC_FWD_CHAIN)
        seq_p, lp_p = run(0)
        after = Lb.persistent_status()
        assert after[0] == 0 and after[1] - before[1] == 1
        live = alive_mask(seq_c)
        if sample_max:
            both = (seq_c == seq_p).cumprod(1).bool() & live
            assert (seq_c == seq_p).all(1).float().mean().item() > 0.85
            assert (lp_c - lp_p)[both].abs().max().item() < LOGP_TOL["bf16"]
        else:
            seq_f, lp_f = run(0, forced=seq_c.cuda())
            assert torch.equal(seq_f, seq_c)
            assert (lp_f - lp_c)[live].abs().max().item() < LOGP_TOL["bf16"]


def test_persistent_launches_on_two_streams_wait_for_each_other(case):
    """Two persistent decode passes enqueued on two streams at the same time (each launch holds every CU while it waits for
    its own workgroups): the library orders them (uic_persist_gate), so neither times out and both give what they give alone."""
    from unpaired_image_captioning_amd import _lib as Lb
    Wg, Ws, b = case
    n = 640
    fc, att, am = b["fc_feats"][:n].cuda(), b["att_feats"][:n].cuda(), b["att_masks"][:n].cuda()
    models = [build_model(CFG, Wg, "bf16"), build_model(CFG, Ws, "bf16")]
    pds = [{k: v.detach() for k, v in m.param_dict().items()} for m in models]
    alone = [m.engine.sample(pd, fc, att, am, L, sample_max=1, seed=5) for m, pd in zip(models, pds)]
    torch.cuda.synchronize()
    before = Lb.persistent_status()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for rep in range(4):
        for i in (0, 1):
            streams[i].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[i]):
                outs[i] = models[i].engine.sample(pds[i], fc, att, am, L, sample_max=1, seed=5)
    torch.cuda.synchronize()
    after = Lb.persistent_status()
    assert after[0] == 0 and after[1] - before[1] == 8
    for i in (0, 1):
        assert torch.equal(outs[i][0], alone[i][0]) and torch.equal(outs[i][1], alone[i][1])


def _force_timeout_word(value):
    from unpaired_image_captioning_amd import _lib as Lb
    Lb.status_words()[0] = value
    torch.cuda.synchronize()


def test_timed_out_persistent_launch_poisons_the_captions_and_raises_at_the_call(case):
    """The status word a persistent launch sets when its bounded spin gives up (forced here by writing it): a lone decode pass
    (eval mode, one persistent launch) must hand back POISONED captions -- token -1, log-prob NaN -- and raise at the call
    itself, not at some later training step; the word is cleared by the raise, and the next pass is sound."""
    from unpaired_image_captioning_amd import _lib as Lb
    Wg, _, b = case
    n = 40
    fc, att, am = b["fc_feats"][:n].cuda(), b["att_feats"][:n].cuda(), b["att_masks"][:n].cuda()
    model = build_model(CFG, Wg, "bf16")
    model.eval()
    good = model(fc, None, att, am, opt={"sample_max": 1}, mode="sample")
    _force_timeout_word(0x51)
    try:
        with pytest.raises(RuntimeError, match="timed out"):
            model(fc, None, att, am, opt={"sample_max": 1}, mode="sample")
        assert Lb.persistent_status()[0] == 0                       # cleared by the raise
        _force_timeout_word(0x51)
        model.defer_status_check = True                              # the caller's own sync: look at what came back
        seq, lp = model(fc, None, att, am, opt={"sample_max": 1}, mode="sample")
        assert (seq == -1).all() and torch.isnan(lp).all()
    finally:
        Lb.status_words().zero_()
        model.defer_status_check = False
    again = model(fc, None, att, am, opt={"sample_max": 1}, mode="sample")
    assert torch.equal(again[0], good[0]) and torch.equal(again[1], good[1])


def test_training_step_after_a_timeout_is_skipped_on_the_device_and_raises(case):
    """Trainer.train with the status word set during the step (forced): uic_adam_step_guarded must leave weights, Adam moments
    and the step counter untouched, the call must raise, and the next call must train exactly as if the bad step had never
    been issued."""
    from unpaired_image_captioning_amd import _lib as Lb
    from unpaired_image_captioning_amd.trainer import Trainer
    from test_gpu_topdown import make_opt
    _, Ws, b = case
    rows = slice(0, 16 * 5)
    data = {k: b[k][rows].numpy() for k in ("fc_feats", "att_feats", "labels", "masks", "att_masks")}

    def fresh():
        opt = make_opt(CFG, "bf16", drop=0.5, seed=11)
        opt.i2t_learning_rate = 1e-3
        opt.seq_per_img = 1
        tr = Trainer(opt)
        tr.i2t_model.load_state_dict(Ws)
        tr.build_optimizer()
        return tr

    ref = fresh()
    ref_losses = [ref.train(data) for _ in range(2)]
    tr = fresh()
    assert tr.train(data) == ref_losses[0]
    snap = (tr.arena.flat.clone(), tr.arena.exp_avg.clone(), tr.arena.exp_avg_sq.clone(), tr._step, tr.i2t_model._seed_counter)
    _force_timeout_word(0x33)
    try:
        with pytest.raises(RuntimeError, match="timed out"):
            tr.train(data)
    finally:
        Lb.status_words().zero_()
    assert torch.equal(tr.arena.flat, snap[0]) and torch.equal(tr.arena.exp_avg, snap[1]) and torch.equal(tr.arena.exp_avg_sq, snap[2])
    assert tr._step == snap[3]
    tr.i2t_model._seed_counter = snap[4]                    # (the skipped step drew a dropout seed)
    assert tr.train(data) == ref_losses[1]
    assert torch.equal(tr.arena.flat, ref.arena.flat)
