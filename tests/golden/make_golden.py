#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the REFERENCE's own modules.

Runs only in the build container (needs /root/reference); the reference's files
never travel.  Imports ``models/AttModel.py`` and ``misc/criterion.py`` from
``/root/reference/pivot_based_eccv2018`` through harness-side shims (SURVEY.md
section 8c): stub ``nltk``, a synthetic ``models`` package so that
``models/__init__.py`` (py2-isms, every architecture) never executes, and
``PYTHONDONTWRITEBYTECODE`` so nothing is written into the reference tree.

Each ``.npz`` holds: the reference model's initial ``state_dict`` (``w::<key>``),
the inputs (``in::<name>``) and the reference outputs (``out::<name>``).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys
import types
import importlib.util
import argparse as _argparse

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

P = "/root/reference/pivot_based_eccv2018"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference():
    for name in ("nltk", "nltk.translate", "nltk.translate.bleu_score"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["nltk.translate.bleu_score"].SmoothingFunction = object
    sys.path.insert(0, P)
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(P, "models")]
    sys.modules["models"] = pkg
    import builtins
    import functools
    builtins.reduce = functools.reduce
    torch.Tensor.cuda = lambda self, *a, **k: self          # CaptionModel.py:131,172 call .cuda() unconditionally

    def load(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    import misc.utils  # noqa: F401  (real reference module, nltk stubbed)
    load("models.CaptionModel", os.path.join(P, "models", "CaptionModel.py"))
    att = load("models.AttModel", os.path.join(P, "models", "AttModel.py"))
    crit = load("refcriterion", os.path.join(P, "misc", "criterion.py"))
    global FC_MOD
    FC_MOD = load("models.FCModel_NMT", os.path.join(P, "models", "FCModel_NMT.py"))
    return att, crit


def make_opt(V, E, H, A, D, L, use_bn=0, drop=0.0, logit_layers=1, Dfc=None):
    return _argparse.Namespace(vocab_size=V, input_encoding_size=E, rnn_size=H, num_layers=1,
                               drop_prob_lm=drop, seq_length=L, fc_feat_size=Dfc or D, att_feat_size=D,
                               att_hid_size=A, use_bn=use_bn, logit_layers=logit_layers, caption_model="topdown")


def synth(n_img, S, R, D, V, L, seed, ragged):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle.topdown import synthetic_batch  # data generator only (no model arithmetic)
    return synthetic_batch(n_img, S, R, D, V, L, seed=seed, ragged_regions=ragged)


def run_case(att_mod, crit_mod, name, V, E, H, A, D, L, n_img, S, R, seed, ragged=False,
             use_masks=True, use_bn=0, short_all=False, adam_steps=0, store_grads=True,
             bn_train=False, ss_prob=0.0, ss_seed=0, logit_layers=1, Dfc=None):
    import builtins
    import functools
    builtins.reduce = functools.reduce                                  # AttModel.py:91 and CaptionModel.py:176 are py2
    att_mod.reduce = functools.reduce
    torch.manual_seed(seed)
    opt = make_opt(V, E, H, A, D, L, use_bn=use_bn, logit_layers=logit_layers, Dfc=Dfc)
    model = att_mod.TopDownModel(opt)
    final_logit = model.logit if logit_layers == 1 else model.logit[-1]
    if use_bn:
        # non-trivial BN affine parameters / running stats
        g = torch.Generator().manual_seed(seed + 7)
        for k, v in model.state_dict().items():
            if "att_embed.0" in k or "att_embed.4" in k:
                if k.endswith("weight"):
                    v.copy_(1 + 0.2 * torch.randn(v.shape, generator=g))
                elif k.endswith("bias") or k.endswith("running_mean"):
                    v.copy_(0.1 * torch.randn(v.shape, generator=g))
                elif k.endswith("running_var"):
                    v.copy_(0.5 + torch.rand(v.shape, generator=g))
    crit = crit_mod.LanguageModelCriterion(opt)
    b = synth(n_img, S, R, D, V, L, seed, ragged)
    if short_all:                                   # every caption ends early -> early break
        b["labels"][:, L // 2 + 1:] = 0
        nz = (b["labels"] != 0).sum(1) + 2
        b["masks"] = (torch.arange(L + 2)[None, :] < nz[:, None]).float()
    fc, att, labels, masks = b["fc_feats"], b["att_feats"], b["labels"], b["masks"]
    if Dfc:
        fc = fc[:, :Dfc].contiguous()               # fc_feat_size != att_feat_size (box features widen only the region features)
    att_masks = b["att_masks"] if use_masks else None
    out = {}
    for k, v in model.state_dict().items():
        out["w::" + k] = v.detach().clone().numpy()
    out["in::fc_feats"] = fc.numpy()
    out["in::att_feats"] = att.numpy()
    out["in::labels"] = labels.numpy()
    out["in::masks"] = masks.numpy()
    if att_masks is not None:
        out["in::att_masks"] = att_masks.numpy()
    out["cfg"] = np.array([V, E, H, A, D, L, n_img, S, R, use_bn, int(bn_train)], dtype=np.int64)
    if logit_layers != 1:
        out["logit_layers"] = np.array(logit_layers, dtype=np.int64)
    if Dfc:
        out["fc_feat_size"] = np.array(Dfc, dtype=np.int64)

    model.train(bool(bn_train))      # drop_prob_lm = 0 -> dropout is the identity either way
    attri = torch.zeros(fc.shape[0], 1)
    # _prepare_feature + first three decode steps (reference methods, called directly)
    p_fc, p_att, pp_att, p_masks = model._prepare_feature(fc, att, att_masks)
    out["out::fc_embed"] = p_fc.detach().numpy()
    out["out::att_embed"] = p_att.detach().numpy()
    out["out::p_att"] = pp_att.detach().numpy()
    if use_bn and bn_train:
        # _prepare_feature above already advanced the running stats once: reload them so the
        # full forward below starts from the recorded initial state
        model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in out.items() if k.startswith("w::")})
    state = model.init_hidden(fc.shape[0])
    for t in range(3):
        xt = model.embed(labels[:, t])
        h_prev = state[0][-1]
        x1 = torch.cat([h_prev, p_fc, xt], 1)
        h_att, c_att = model.core.att_lstm(x1, (state[0][0], state[1][0]))
        att_res = model.core.attention(h_att, p_att, pp_att, p_masks)
        logp, state = model.get_logprobs_state(labels[:, t], p_fc, p_att, pp_att, p_masks, state)
        out["out::step%d_h_att" % t] = h_att.detach().numpy()
        out["out::step%d_c_att" % t] = c_att.detach().numpy()
        out["out::step%d_att_res" % t] = att_res.detach().numpy()
        out["out::step%d_h_lang" % t] = state[0][1].detach().numpy()
        out["out::step%d_c_lang" % t] = state[1][1].detach().numpy()
        out["out::step%d_logp" % t] = logp.detach().numpy()
    if use_bn and bn_train:
        model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in out.items() if k.startswith("w::")})

    # full teacher-forced forward through the reference's public call convention
    model.zero_grad()
    if ss_prob > 0:
        # scheduled sampling (AttModel.py:130-143): train mode, torch's CPU generator seeded right before the call, so
        # the oracle can reproduce the reference's uniform_/multinomial draws call for call
        model.train()
        model.ss_prob = ss_prob
        out["ss"] = np.array([ss_prob, ss_seed], dtype=np.float64)
        torch.manual_seed(ss_seed)
    logp = model(fc, attri, att, labels, att_masks)                     # mode='forward'
    loss = crit(logp, labels[:, 1:], masks[:, 1:])
    loss.backward()
    out["out::logprobs"] = logp.detach().numpy()
    out["out::loss"] = np.array(loss.item(), dtype=np.float64)
    if store_grads:
        for k, p in model.named_parameters():
            out["grad::" + k] = p.grad.detach().clone().numpy()
    if use_bn and bn_train:
        for k, v in model.state_dict().items():
            if "running" in k or "num_batches" in k:
                out["bnstat::" + k] = v.detach().clone().numpy()

    # greedy decode (eval mode), one row per image as eval_utils.eval_split does (:256-263)
    model.eval()
    idx = torch.arange(n_img) * S
    with torch.no_grad():
        seq, seq_logp = model(fc[idx], attri[idx], att[idx], att_masks[idx] if att_masks is not None else None,
                              opt={"sample_max": 1, "beam_size": 1}, mode="sample")
    out["out::greedy_seq"] = seq.numpy()
    out["out::greedy_logp"] = seq_logp.numpy()

    # beam search (AttModel._sample_beam + CaptionModel.beam_search) through the public call convention; the EOS bias
    # variant raises logit.bias[0] so that beams really finish early (random weights hardly ever emit token 0)
    import builtins
    import functools
    builtins.reduce = functools.reduce                                  # CaptionModel.py:176 is py2
    for tag, bs, dc, mp, eos_bias in (("b3", 3, 0, 0, 0.0), ("b2c", 2, 1, 0, 0.0), ("b3eos", 3, 0, 0, 3.0), ("b4ppl", 4, 1, 1, 2.5)):
        with torch.no_grad():
            final_logit.bias[0] += eos_bias
            bseq, blp = model(fc[idx], attri[idx], att[idx], att_masks[idx] if att_masks is not None else None,
                              opt={"sample_max": 1, "beam_size": bs, "decoding_constraint": dc, "max_ppl": mp}, mode="sample")
            final_logit.bias[0] -= eos_bias
        out["beam::%s_cfg" % tag] = np.array([bs, dc, mp, eos_bias], dtype=np.float64)
        out["beam::%s_seq" % tag] = bseq.numpy().copy()
        out["beam::%s_logp" % tag] = blp.numpy().copy()

    # diverse beam search (group_size > 1, CaptionModel.py:36-45,100-176): _sample_beam returns done_beams[k][0], the best
    # beam of group 0 -- the group that never sees a diversity penalty
    for tag, bs, gs, dc, mp, eos_bias, lam in (("g2b4", 4, 2, 0, 0, 0.0, 0.5), ("g3b6eos", 6, 3, 1, 1, 2.5, 0.7)):
        with torch.no_grad():
            final_logit.bias[0] += eos_bias
            bseq, blp = model(fc[idx], attri[idx], att[idx], att_masks[idx] if att_masks is not None else None,
                              opt={"sample_max": 1, "beam_size": bs, "group_size": gs, "diversity_lambda": lam,
                                   "decoding_constraint": dc, "max_ppl": mp}, mode="sample")
            final_logit.bias[0] -= eos_bias
        out["beamg::%s_cfg" % tag] = np.array([bs, gs, dc, mp, eos_bias, lam], dtype=np.float64)
        out["beamg::%s_seq" % tag] = bseq.numpy().copy()
        out["beamg::%s_logp" % tag] = blp.numpy().copy()

    # RewardCriterion on a hand-made reward
    g = torch.Generator().manual_seed(seed + 1)
    reward = torch.randn(seq.shape, generator=g)
    rl = crit_mod.RewardCriterion()(seq_logp, seq, reward)
    out["in::reward"] = reward.numpy()
    out["out::reward_loss"] = np.array(rl.item(), dtype=np.float64)

    if adam_steps:
        # deterministic trajectory: Trainer.train semantic = forward, criterion, backward, Adam
        model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in out.items() if k.startswith("w::")})
        model.train(True)
        optim = torch.optim.Adam(model.parameters(), 5e-4, (0.9, 0.999), 1e-8, weight_decay=0)
        losses = []
        for _ in range(adam_steps):
            optim.zero_grad()
            lp = model(fc, attri, att, labels, att_masks)
            ls = crit(lp, labels[:, 1:], masks[:, 1:])
            ls.backward()
            optim.step()
            losses.append(ls.item())
        out["out::adam_losses"] = np.array(losses, dtype=np.float64)
        out["out::adam_final_logit_bias"] = model.logit.bias.detach().numpy()
        out["out::adam_final_h2att_weight"] = model.core.attention.h2att.weight.detach().numpy()

    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) loss=%.6f" % (path, os.path.getsize(path) / 1024, loss.item()))


FC_MOD = None


def run_fc_case(crit_mod, name, V, E, H, D, L, n_img, S, seed, short_all=False, wseed=None):
    """FCModel_NMT (the `fc` caption model, P/models/__init__.py:24-26): forward, loss, grads, greedy decode."""
    torch.manual_seed(seed)
    opt = _argparse.Namespace(vocab_size=V, input_encoding_size=E, rnn_type="LSTM", rnn_size=H, num_layers=1,
                              drop_prob_lm=0.0, seq_length=L, fc_feat_size=D, caption_model="fc")
    model = FC_MOD.FCModel_NMT(opt)
    if wseed is not None:        # big case: weights are a pure function of the seed (oracle.fc.init_weights), not stored
        from oracle.fc import init_weights as fc_init
        model.load_state_dict(fc_init(V + 1, E, H, D, seed=wseed))
    crit = crit_mod.LanguageModelCriterion(opt)
    b = synth(n_img, S, 3, D, V, L, seed, False)
    if short_all:
        b["labels"][:, L // 2 + 1:] = 0
        nz = (b["labels"] != 0).sum(1) + 2
        b["masks"] = (torch.arange(L + 2)[None, :] < nz[:, None]).float()
    fc, labels, masks = b["fc_feats"], b["labels"], b["masks"]
    out = {"cfg": np.array([V, E, H, 0, D, L, n_img, S, 0, 0, 0], dtype=np.int64)}
    if wseed is None:
        for k, v in model.state_dict().items():
            out["w::" + k] = v.detach().clone().numpy()
        out["in::fc_feats"] = fc.numpy()
        out["in::labels"] = labels.numpy()
        out["in::masks"] = masks.numpy()
    else:
        out["seeds"] = np.array([wseed, seed], dtype=np.int64)
    model.train()
    logp = model._forward(fc, None, labels)                      # the reference's own 4-argument signature (:89)
    loss = crit(logp, labels[:, 1:], masks[:, 1:])
    loss.backward()
    out["out::loss"] = np.array(loss.item(), dtype=np.float64)
    if wseed is None:
        out["out::logprobs"] = logp.detach().numpy()
        for k, p in model.named_parameters():
            out["grad::" + k] = p.grad.detach().clone().numpy()
    else:
        out["out::logprobs_sub"] = logp.detach()[:, :, ::37].numpy()
        for k, p in model.named_parameters():
            if p.dim() == 1:
                out["grad::" + k] = p.grad.detach().clone().numpy()
            else:
                out["gradnorm::" + k] = np.array(p.grad.double().norm().item())
    model.eval()
    idx = torch.arange(n_img) * S
    with torch.no_grad():
        seq, seq_logp = model._sample(fc[idx], None, None, {"sample_max": 1, "beam_size": 1})
    out["out::greedy_seq"] = seq.numpy()
    out["out::greedy_logp"] = seq_logp.numpy()
    # beam search (FCModel_NMT._sample_beam + CaptionModel.beam_search); the EOS-bias variants make beams finish early
    if H <= 64:
        for tag, bs, dc, mp, eos_bias in (("b3", 3, 0, 0, 0.0), ("b2c", 2, 1, 0, 0.0), ("b3eos", 3, 0, 0, 3.0), ("b4ppl", 4, 1, 1, 2.5)):
            with torch.no_grad():
                model.logit.bias[0] += eos_bias
                opts = {"sample_max": 1, "beam_size": bs, "decoding_constraint": dc, "max_ppl": mp}
                # public path: FCModel_NMT._sample hands `opt` to _sample_beam as its att_masks argument (:168), so the
                # search really runs with the defaults beam_size 10, no constraint, no max_ppl
                bseq, blp = model._sample(fc[idx], None, None, opts)
                # direct call with the options honoured
                dseq, dlp = model._sample_beam(fc[idx], None, None, opts)
                model.logit.bias[0] -= eos_bias
            out["beam::%s_cfg" % tag] = np.array([bs, dc, mp, eos_bias], dtype=np.float64)
            out["beam::%s_seq" % tag] = bseq.numpy().copy()
            out["beam::%s_logp" % tag] = blp.numpy().copy()
            out["beamd::%s_seq" % tag] = dseq.numpy().copy()
            out["beamd::%s_logp" % tag] = dlp.numpy().copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) loss=%.6f" % (path, os.path.getsize(path) / 1024, loss.item()))


def main():
    att_mod, crit_mod = _load_reference()
    run_fc_case(crit_mod, "fc_tiny", V=50, E=32, H=32, D=64, L=6, n_img=3, S=2, seed=21)
    run_fc_case(crit_mod, "fc_tiny_earlybreak", V=50, E=32, H=32, D=64, L=6, n_img=3, S=2, seed=22, short_all=True)
    run_fc_case(crit_mod, "fc_odd", V=77, E=24, H=40, D=72, L=5, n_img=2, S=3, seed=23)
    # BASELINE config 1 shapes: batch 16 images x 5 captions, seq_len 16, 2048-d fc feats, hidden 512 (weights from a seed,
    # outputs sub-sampled, to keep the fixture small)
    run_fc_case(crit_mod, "fc_cfg1", V=9487, E=512, H=512, D=2048, L=16, n_img=16, S=5, seed=24, wseed=2025)
    tiny = dict(V=50, E=32, H=32, A=32, D=64, L=6, n_img=3, S=2, R=5)
    run_case(att_mod, crit_mod, "topdown_tiny", seed=11, adam_steps=3, **tiny)
    run_case(att_mod, crit_mod, "topdown_tiny_ragged", seed=12, ragged=True, **tiny)
    run_case(att_mod, crit_mod, "topdown_tiny_nomask", seed=13, use_masks=False, **tiny)
    run_case(att_mod, crit_mod, "topdown_tiny_earlybreak", seed=14, short_all=True, **tiny)
    run_case(att_mod, crit_mod, "topdown_tiny_bn1_eval", seed=15, use_bn=1, ragged=True, **tiny)
    run_case(att_mod, crit_mod, "topdown_tiny_bn2_train", seed=16, use_bn=2, ragged=True, bn_train=True, **tiny)
    run_case(att_mod, crit_mod, "topdown_tiny_ss", seed=18, ragged=True, ss_prob=0.5, ss_seed=97, **tiny)
    # logit_layers > 1 (AttModel.py:86-91): eval mode, so the hard-coded Dropout(0.5) of the hidden blocks is the identity
    run_case(att_mod, crit_mod, "topdown_tiny_logit2", seed=19, ragged=True, logit_layers=2, **tiny)
    run_case(att_mod, crit_mod, "topdown_tiny_logit3_bn1", seed=20, ragged=True, logit_layers=3, use_bn=1, **tiny)
    # use_box = 1 (the reference's default, P/opts.py:80): 5 box features widen the region features, att_feat_size is no
    # multiple of 8 any more; with the default use_bn = 1 (train-mode statistics) and without BatchNorm
    box = dict(tiny, D=69)
    run_case(att_mod, crit_mod, "topdown_tiny_box_bn1", seed=31, ragged=True, use_bn=1, bn_train=True, Dfc=64, **box)
    run_case(att_mod, crit_mod, "topdown_tiny_box", seed=32, ragged=True, Dfc=64, **box)
    # non-power-of-two / odd sizes (E != H != A, V1 not a tile multiple)
    run_case(att_mod, crit_mod, "topdown_odd", seed=17, V=77, E=24, H=40, A=48, D=72, L=5,
             n_img=2, S=3, R=7, ragged=True)
    # row-subsampled real-size config (BASELINE config 2 shapes, N = 4): weights are big,
    # so grads are not stored and weights are float16-rounded before running the reference
    run_real(att_mod, crit_mod)


def run_real(att_mod, crit_mod):
    """N=4 rows at R=36, D=2048, H=E=A=512, V1=9488.  To keep the fixture small the weights
    are NOT stored: they are regenerated from a seed by ``oracle.topdown.init_weights`` (a pure
    function of the seed) and loaded into the reference model; only inputs' seed + outputs are stored."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle.topdown import init_weights
    V, E, H, A, D, L, n_img, S, R = 9487, 512, 512, 512, 2048, 16, 2, 2, 36
    opt = make_opt(V, E, H, A, D, L)
    model = att_mod.TopDownModel(opt)
    W = init_weights(V + 1, E, H, A, D, D, seed=2024)
    model.load_state_dict(W)
    crit = crit_mod.LanguageModelCriterion(opt)
    b = synth(n_img, S, R, D, V, L, 99, True)
    fc, att, labels, masks, att_masks = b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"]
    attri = torch.zeros(fc.shape[0], 1)
    model.eval()
    model.zero_grad()
    logp = model(fc, attri, att, labels, att_masks)
    loss = crit(logp, labels[:, 1:], masks[:, 1:])
    loss.backward()
    out = {"cfg": np.array([V, E, H, A, D, L, n_img, S, R, 0, 0], dtype=np.int64),
           "seeds": np.array([2024, 99], dtype=np.int64)}
    # store a strided subsample of the log-probs (full tensor is 2.6 MB) + loss + small grads
    out["out::logprobs_sub"] = logp.detach()[:, :, ::37].numpy()
    out["out::logprobs_rowsum"] = logp.detach().double().sum(2).numpy()
    out["out::loss"] = np.array(loss.item(), dtype=np.float64)
    for k in ("core.attention.alpha_net.weight", "core.attention.h2att.bias", "ctx2att.bias",
              "core.att_lstm.bias_ih", "core.lang_lstm.bias_hh", "fc_embed.0.bias", "logit.bias"):
        out["grad::" + k] = dict(model.named_parameters())[k].grad.detach().numpy()
    out["gradnorm::att_embed.0.weight"] = np.array(model.att_embed[0].weight.grad.double().norm().item())
    out["gradnorm::logit.weight"] = np.array(model.logit.weight.grad.double().norm().item())
    out["gradnorm::embed.0.weight"] = np.array(model.embed[0].weight.grad.double().norm().item())
    idx = torch.arange(n_img) * S
    with torch.no_grad():
        seq, seq_logp = model(fc[idx], attri[idx], att[idx], att_masks[idx],
                              opt={"sample_max": 1, "beam_size": 1}, mode="sample")
    out["out::greedy_seq"] = seq.numpy()
    out["out::greedy_logp"] = seq_logp.numpy()
    path = os.path.join(HERE, "topdown_real_n4.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) loss=%.6f" % (path, os.path.getsize(path) / 1024, loss.item()))


if __name__ == "__main__":
    main()
