#!/usr/bin/env python3
"""Golden vectors for the pivot NMT path from the REFERENCE's own modules:
``models/NMT_Models.py`` (Embeddings, Encoder, Decoder, NMTModel.forward) with the vendored OpenNMT fork's
``StackedRNN.py`` / ``GlobalAttention.py`` / ``Util.py`` / ``Gate.py`` and ``misc/criterion.py::NMTCriterion``.

Build-container only.  Harness-side shims (reference files untouched): a synthetic ``onmt``
package assembled from the py3-clean files of the fork, ``onmt.modules.activations`` and ``evaluation`` stubbed,
``.cuda()`` neutralised (NMT_Models.py:231 calls it unconditionally), ``nltk`` stubbed.

    python tests/golden/make_golden_nmt.py
"""
import argparse
import importlib.util
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

P = "/root/reference/pivot_based_eccv2018"
O = os.path.join(P, "misc", "OpenNMT-py-dalegebit", "onmt")
HERE = os.path.dirname(os.path.abspath(__file__))


def load(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for name in ("nltk", "nltk.translate", "nltk.translate.bleu_score", "evaluation"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["nltk.translate.bleu_score"].SmoothingFunction = object
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    onmt = types.ModuleType("onmt")
    onmt.__path__ = []
    sys.modules["onmt"] = onmt
    onmt.Constants = load("onmt.Constants", os.path.join(O, "Constants.py"))
    mods = types.ModuleType("onmt.modules")
    mods.__path__ = []
    sys.modules["onmt.modules"] = mods
    onmt.modules = mods
    act = types.ModuleType("onmt.modules.activations")
    for n in ("Softmax", "Sparsemax", "ConstrainedSoftmax", "ConstrainedSparsemax"):
        setattr(act, n, type(n, (nn.Module,), {}))
    sys.modules["onmt.modules.activations"] = act
    util = load("onmt.modules.Util", os.path.join(O, "modules", "Util.py"))
    mods.aeq, mods.BottleLinear, mods.Bottle = util.aeq, util.BottleLinear, util.Bottle
    srnn = load("onmt.modules.StackedRNN", os.path.join(O, "modules", "StackedRNN.py"))
    mods.StackedLSTM, mods.StackedGRU = srnn.StackedLSTM, srnn.StackedGRU
    ga = load("onmt.modules.GlobalAttention", os.path.join(O, "modules", "GlobalAttention.py"))
    mods.GlobalAttention = ga.GlobalAttention
    load("onmt.modules.Gate", os.path.join(O, "modules", "Gate.py"))
    sys.path.insert(0, P)
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(P, "models")]
    sys.modules["models"] = pkg
    nmt = load("models.NMT_Models", os.path.join(P, "models", "NMT_Models.py"))
    crit = load("refcriterion", os.path.join(P, "misc", "criterion.py"))
    return nmt, crit


class FakeDict(object):
    def __init__(self, n):
        self.n = n

    def size(self):
        return self.n

    def align(self, other):
        return [0] * self.n


def make_opt(layers, rnn_size, wvec, dropout=0.0):
    return argparse.Namespace(position_encoding=False, word_vec_size=wvec, dropout=dropout, layers=layers, brnn=True,
                              rnn_size=rnn_size, encoder_layer="rnn", decoder_layer="rnn", rnn_type="LSTM", fertility=2.0,
                              predict_fertility=False, supervised_fertility=None, guided_fertility=None, coverage_attn=False,
                              exhaustion_loss=False, input_feed=1, context_gate=None, attention_type="dotprod",
                              attn_transform="softmax", c_attn=0.0, copy_attn=False, gpus=[], batch_size=4,
                              lambda_coverage=1, lambda_fertility=0.4, lambda_exhaust=0.5)


def synth_batch(B, S, T, Vs, Vt, seed):
    """Length-sorted padded src/tgt like onmt_dataset_h5.Batch (P/misc/dataloader/onmt_dataset_h5.py:45-107):
    src [S,B,1] (PAD = 0), lengths [1,B] descending, tgt [T,B] = BOS .. EOS PAD.."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.sort(torch.randint(max(2, S // 2), S + 1, (B,), generator=g), descending=True)[0]
    lens[0] = S
    src = torch.zeros(S, B, 1, dtype=torch.long)
    for b in range(B):
        src[:lens[b], b, 0] = torch.randint(4, Vs, (int(lens[b]),), generator=g)
    tl = torch.randint(max(3, T // 2), T + 1, (B,), generator=g)
    tl[0] = T
    tgt = torch.zeros(T, B, dtype=torch.long)
    for b in range(B):
        n = int(tl[b])
        tgt[0, b] = 2
        tgt[1:n - 1, b] = torch.randint(4, Vt, (n - 2,), generator=g)
        tgt[n - 1, b] = 3
    return src, lens.view(1, -1), tgt


def run_case(nmt, crit_mod, name, layers, H, B, S, T, Vs, Vt, seed):
    torch.manual_seed(seed)
    opt = make_opt(layers, H, H)
    sd, td = FakeDict(Vs), FakeDict(Vt)
    enc = nmt.Encoder(opt, sd)
    dec = nmt.Decoder(opt, td)
    model = nmt.NMTModel(opt, enc, dec, sd, td)
    generator = nn.Sequential(nn.Linear(H, Vt), nn.LogSoftmax(dim=1))       # P/trainer.py:85
    model.generator = generator
    loss_fn = crit_mod.NMTCriterion(Vt, opt)
    src, lengths, tgt = synth_batch(B, S, T, Vs, Vt, seed)
    model.train()
    outputs, attns, dec_state, ub = model(src, tgt, lengths)
    scores = generator(outputs.view(-1, outputs.size(2)))
    loss = loss_fn(scores, tgt[1:].view(-1))
    loss.backward()
    enc_hidden, context, _ = enc(src, lengths)
    out = {"cfg": np.array([layers, H, B, S, T, Vs, Vt], dtype=np.int64)}
    for k, v in model.state_dict().items():
        out["w::" + k] = v.detach().clone().numpy()
    out["in::src"] = src.numpy()
    out["in::lengths"] = lengths.numpy()
    out["in::tgt"] = tgt.numpy()
    out["out::context"] = context.detach().numpy()
    out["out::enc_h"] = model._fix_enc_hidden(enc_hidden[0]).detach().numpy()
    out["out::enc_c"] = model._fix_enc_hidden(enc_hidden[1]).detach().numpy()
    out["out::outputs"] = outputs.detach().numpy()
    out["out::attn"] = attns["std"].detach().numpy()
    out["out::scores"] = scores.detach().numpy()
    out["out::loss"] = np.array(loss.item(), dtype=np.float64)
    pred = scores.max(1)[1]
    nonpad = tgt[1:].view(-1).ne(0)
    out["out::num_correct"] = np.array(int((pred.eq(tgt[1:].view(-1)) & nonpad).sum()))
    out["out::num_words"] = np.array(int(nonpad.sum()))
    for k, p in model.named_parameters():
        out["grad::" + k] = p.grad.detach().clone().numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) loss=%.5f keys=%d" % (path, os.path.getsize(path) / 1024, loss.item(), len(model.state_dict())))


def recipe_weights(state_dict, seed, scale):
    """Weights as a pure function of (state_dict order and shapes, seed, scale): U(-scale, scale) from ONE torch CPU generator,
    tensor after tensor.  The GPU-side test rebuilds them with the same three lines (tests/test_gpu_nmt.py::recipe_weights),
    so a fixture at the real widths (50 004-word vocabularies: 300 MB of weights) only has to store results."""
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.rand(v.shape, generator=g) * 2 - 1) * scale for k, v in state_dict.items()}


def run_real_width_case(nmt, crit_mod, name, layers, H, B, S, T, Vs, Vt, seed, scale=0.08):
    """BASELINE configs[2] at its real widths (2 layers, 512, vocabularies of 50 004) on a few sentences: outputs, attention,
    loss, counters, the norm of every gradient tensor, decoder.attn.linear_in.weight's gradient in full and the generator /
    embedding gradient rows of the words that occur."""
    torch.manual_seed(seed)
    opt = make_opt(layers, H, H)
    sd, td = FakeDict(Vs), FakeDict(Vt)
    model = nmt.NMTModel(opt, nmt.Encoder(opt, sd), nmt.Decoder(opt, td), sd, td)
    generator = nn.Sequential(nn.Linear(H, Vt), nn.LogSoftmax(dim=1))
    model.generator = generator
    model.load_state_dict(recipe_weights(model.state_dict(), seed, scale))
    loss_fn = crit_mod.NMTCriterion(Vt, opt)
    src, lengths, tgt = synth_batch(B, S, T, Vs, Vt, seed)
    model.train()
    outputs, attns, _, _ = model(src, tgt, lengths)
    scores = generator(outputs.view(-1, outputs.size(2)))
    loss = loss_fn(scores, tgt[1:].view(-1))
    loss.backward()
    out = {"cfg": np.array([layers, H, B, S, T, Vs, Vt], dtype=np.int64), "recipe": np.array([seed, scale], dtype=np.float64),
           "keys": np.array(list(model.state_dict().keys())),
           "in::src": src.numpy(), "in::lengths": lengths.numpy(), "in::tgt": tgt.numpy(),
           "out::outputs": outputs.detach().numpy(), "out::attn": attns["std"].detach().numpy(),
           "out::loss": np.array(loss.item(), dtype=np.float64)}
    pred = scores.max(1)[1]
    nonpad = tgt[1:].view(-1).ne(0)
    out["out::num_correct"] = np.array(int((pred.eq(tgt[1:].view(-1)) & nonpad).sum()))
    out["out::num_words"] = np.array(int(nonpad.sum()))
    tgt_rows = torch.unique(tgt[1:].reshape(-1))
    src_rows = torch.unique(src.reshape(-1))
    dec_rows = torch.unique(tgt[:-1].reshape(-1))
    out["rows::generator"], out["rows::enc_lut"], out["rows::dec_lut"] = tgt_rows.numpy(), src_rows.numpy(), dec_rows.numpy()
    for k, p_ in model.named_parameters():
        gr = p_.grad.detach()
        out["gnorm::" + k] = np.array(float(gr.double().norm()), dtype=np.float64)
        if k == "decoder.attn.linear_in.weight":
            out["grad::" + k] = gr.numpy()
        elif k == "generator.0.weight":
            out["gradrows::" + k] = gr[tgt_rows].numpy()
        elif k == "encoder.embeddings.word_lut.weight":
            out["gradrows::" + k] = gr[src_rows].numpy()
        elif k == "decoder.embeddings.word_lut.weight":
            out["gradrows::" + k] = gr[dec_rows].numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) loss=%.5f words=%d" % (path, os.path.getsize(path) / 1024, loss.item(), int(nonpad.sum())))


def optim_opt(**kw):
    o = argparse.Namespace(
        i2t_train_flag=0, i2t_eval_flag=0, i2t_optim="adam", i2t_learning_rate=4e-4, i2t_learning_rate_decay_start=0,
        i2t_learning_rate_decay_every=3, i2t_learning_rate_decay_rate=0.8, i2t_optim_alpha=0.9, i2t_optim_beta=0.999,
        i2t_optim_epsilon=1e-8, i2t_momentum=0, i2t_max_grad_norm=0, i2t_grad_clip=0.1, i2t_decay_method="", i2t_weight_decay=0,
        nmt_train_flag=1, nmt_eval_flag=0, nmt_optim="adam", nmt_learning_rate=1e-3, nmt_learning_rate_decay_start=8,
        nmt_learning_rate_decay_every=3, nmt_learning_rate_decay_rate=0.5, nmt_optim_alpha=0.9, nmt_optim_beta=0.999,
        nmt_optim_epsilon=1e-8, nmt_momentum=0, nmt_max_grad_norm=5, nmt_grad_clip=0.1, nmt_decay_method="", nmt_weight_decay=0,
        nmt_warmup_steps=4000, rnn_size=32, start_from=None, scheduled_sampling_start=0, scheduled_sampling_increase_every=5,
        scheduled_sampling_increase_prob=0.05, scheduled_sampling_max_prob=0.25)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def run_optim_case(nmt, crit_mod, name, steps, seed, **optkw):
    """Trainer.train's NMT half (P/trainer.py:175-193) driven by the reference's own Optim (P/misc/optimizer.py:59-131):
    zero_grad, forward, NMTCriterion, backward, step (noam LR, clip_grad_norm 5, Adam) on ONE fixed batch."""
    import torch.nn.utils as U
    U.clip_grad_norm = U.clip_grad_norm_                     # removed spelling, imported by misc/optimizer.py:5
    optim_mod = load("refoptimizer", os.path.join(P, "misc", "optimizer.py"))
    layers, H, B, S, T, Vs, Vt = 2, 32, 4, 9, 10, 40, 45
    torch.manual_seed(seed)
    mopt = make_opt(layers, H, H)
    sd, td = FakeDict(Vs), FakeDict(Vt)
    model = nmt.NMTModel(mopt, nmt.Encoder(mopt, sd), nmt.Decoder(mopt, td), sd, td)
    generator = nn.Sequential(nn.Linear(H, Vt), nn.LogSoftmax(dim=1))
    model.generator = generator
    loss_fn = crit_mod.NMTCriterion(Vt, mopt)
    src, lengths, tgt = synth_batch(B, S, T, Vs, Vt, seed)
    oo = optim_opt(**optkw)
    optim = optim_mod.Optim(oo)
    optim.set_parameters(None, model)
    out = {"cfg": np.array([layers, H, B, S, T, Vs, Vt], dtype=np.int64),
           "optcfg": np.array([oo.nmt_learning_rate, oo.nmt_max_grad_norm, oo.nmt_warmup_steps, 1.0 if oo.nmt_decay_method == "noam" else 0.0,
                               oo.nmt_optim_alpha, oo.nmt_optim_beta, oo.nmt_optim_epsilon], dtype=np.float64)}
    for k, v in model.state_dict().items():
        out["w::" + k] = v.detach().clone().numpy()
    out["in::src"], out["in::lengths"], out["in::tgt"] = src.numpy(), lengths.numpy(), tgt.numpy()
    losses, norms, lrs = [], [], []
    model.train()
    for it in range(steps):
        optim.zero_grad()
        outputs, attns, _, _ = model(src, tgt, lengths)
        loss = loss_fn(generator(outputs.view(-1, outputs.size(2))), tgt[1:].view(-1))
        loss.backward()
        norms.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters()))))
        optim.step()
        losses.append(loss.item())
        lrs.append(optim.nmt_optimizer.param_groups[0]["lr"])
    out["out::losses"] = np.array(losses, dtype=np.float64)
    out["out::grad_norms"] = np.array(norms, dtype=np.float64)
    out["out::lrs"] = np.array(lrs, dtype=np.float64)
    for k, v in model.state_dict().items():
        out["final::" + k] = v.detach().clone().numpy()
    # the epoch schedules (P/misc/optimizer.py:108-131)
    sched_nmt, sched_i2t, sched_ss = [], [], []
    holder = argparse.Namespace(ss_prob=0.0)
    o2 = optim_mod.Optim(optim_opt(i2t_train_flag=1))
    o2.set_parameters(nn.Linear(2, 2), model)
    for epoch in range(14):
        o2.update_LearningRate("nmt", epoch)
        o2.update_LearningRate("i2t", epoch)
        o2.update_ScheduledSampling_prob(o2.opt, epoch, holder)
        sched_nmt.append(o2.nmt_current_lr); sched_i2t.append(o2.i2t_current_lr); sched_ss.append(holder.ss_prob)
    out["out::sched_nmt"] = np.array(sched_nmt); out["out::sched_i2t"] = np.array(sched_i2t); out["out::sched_ss"] = np.array(sched_ss)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) losses=%s norms=%s lrs=%s" % (path, os.path.getsize(path) / 1024, losses, norms, lrs))


def old_torch_semantics():
    """The 2018 translator relies on torch 0.3/0.4 behaviour that later versions changed.  Emulated in the harness (the
    reference files stay untouched): torch.cuda.*Tensor constructors on a CPU-only box, integer `/` as floor division
    (O/Beam.py:77 `prevK = bestScoresId / numWords`), and masked_fill_ with a same-numel mask of a different shape
    (O/modules/GlobalAttention.py:139 fills a [batch*beam, S] tensor with a [beam, batch, S] mask)."""
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.LongTensor = torch.LongTensor
    _td = torch.Tensor.__truediv__

    def _old_div(a, b):
        if not a.is_floating_point() and (isinstance(b, int) or (torch.is_tensor(b) and not b.is_floating_point())):
            return torch.div(a, b, rounding_mode="floor")
        return _td(a, b)
    torch.Tensor.__truediv__ = _old_div
    _mf = torch.Tensor.masked_fill_

    def _old_masked_fill_(self, mask, value):
        if mask.shape != self.shape and mask.numel() == self.numel():
            mask = mask.reshape(self.shape)
        return _mf(self, mask.bool(), value)
    torch.Tensor.masked_fill_ = _old_masked_fill_


def run_translate_case(nmt, name, layers, H, B, S, Vs, Vt, seed, eos_bias, min_len=2):
    """NMTModel.translateBatch (P/models/NMT_Models.py:322-395) with the fork's Beam (O/Beam.py): beam 15, at most 100
    steps, source padding masked in the attention, encoder run WITHOUT lengths (PAD positions go through the LSTM)."""
    import onmt
    onmt.Beam = load("onmt.Beam", os.path.join(O, "Beam.py")).Beam
    torch.manual_seed(seed)
    opt = make_opt(layers, H, H)
    sd, td = FakeDict(Vs), FakeDict(Vt)
    model = nmt.NMTModel(opt, nmt.Encoder(opt, sd), nmt.Decoder(opt, td), sd, td)
    model.generator = nn.Sequential(nn.Linear(H, Vt), nn.LogSoftmax(dim=1))
    with torch.no_grad():
        # stronger weights than the default init, so that the output distribution really depends on the decoder state (with
        # the default it is almost the same at every step: EOS either wins at step 0 or never), then an EOS (= 3) bias tuned
        # per case so that the sentences finish at different steps
        for k, v in model.named_parameters():
            if "rnn" in k or "attn" in k:
                v.mul_(4.0)
            elif "word_lut" in k:
                v.mul_(3.0)
        model.generator[0].weight.mul_(10.0)
        model.generator[0].bias[3] += eos_bias
    src, lengths, _ = synth_batch(B, S, 4, Vs, Vt, seed)
    if min_len < S:                                       # more padding than synth_batch makes
        for b in range(1, B):
            n = max(min_len, int(lengths[0, b]) - b)
            src[n:, b, 0] = 0
    batch = argparse.Namespace(src=src, batchSize=B)
    model.eval()
    with torch.no_grad():
        allHyp, allScores, allAttn, gold = model.translateBatch(batch)
    n_iter = len(allHyp[0][0])
    out = {"cfg": np.array([layers, H, B, S, n_iter, Vs, Vt], dtype=np.int64)}
    for k, v in model.state_dict().items():
        out["w::" + k] = v.detach().clone().numpy()
    out["in::src"] = src.numpy()
    out["out::hyp"] = np.array([[int(t) for t in allHyp[b][0]] for b in range(B)], dtype=np.int64)
    out["out::scores"] = np.array([float(allScores[b][0]) for b in range(B)], dtype=np.float64)
    attn = np.zeros((B, n_iter, S), dtype=np.float32)
    for b in range(B):
        a = allAttn[b][0].numpy()
        attn[b, :, :a.shape[1]] = a                       # (columns of PAD positions were dropped by the reference)
    out["out::attn"] = attn
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) n_iter=%d first EOS at %s scores=%s" % (
        path, os.path.getsize(path) / 1024, n_iter, [([int(t) for t in allHyp[b][0]] + [3]).index(3) for b in range(B)],
        [round(float(allScores[b][0]), 3) for b in range(B)]))


if __name__ == "__main__":
    nmt, crit_mod = load_reference()
    run_optim_case(nmt, crit_mod, "nmt_optim_clip", steps=4, seed=41, nmt_learning_rate=1e-2)
    run_optim_case(nmt, crit_mod, "nmt_optim_noam", steps=4, seed=42, nmt_learning_rate=0.2, nmt_decay_method="noam", nmt_warmup_steps=3,
                   nmt_max_grad_norm=0)
    run_case(nmt, crit_mod, "nmt_tiny", layers=2, H=32, B=4, S=9, T=10, Vs=40, Vt=45, seed=31)
    run_case(nmt, crit_mod, "nmt_tiny_1layer", layers=1, H=32, B=3, S=6, T=7, Vs=30, Vt=37, seed=32)
    run_case(nmt, crit_mod, "nmt_odd", layers=2, H=48, B=5, S=11, T=8, Vs=53, Vt=61, seed=33)
    run_real_width_case(nmt, crit_mod, "nmt_real_b4", layers=2, H=512, B=4, S=12, T=11, Vs=50004, Vt=50004, seed=34)
    old_torch_semantics()                                  # (after the training-path cases: they need none of it)
    run_translate_case(nmt, "nmt_translate_tiny", layers=2, H=32, B=3, S=7, Vs=30, Vt=35, seed=51, eos_bias=4.0)      # 7 steps
    run_translate_case(nmt, "nmt_translate_odd", layers=2, H=48, B=5, S=9, Vs=41, Vt=52, seed=52, eos_bias=8.0)       # 6 steps
    run_translate_case(nmt, "nmt_translate_1layer", layers=1, H=32, B=2, S=5, Vs=30, Vt=33, seed=53, eos_bias=2.0)    # 4 steps
    run_translate_case(nmt, "nmt_translate_long", layers=2, H=32, B=3, S=7, Vs=30, Vt=35, seed=51, eos_bias=3.0)      # all 100 steps
