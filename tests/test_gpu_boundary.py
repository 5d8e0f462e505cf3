"""The step-by-step decoding surface of the reference and its option contract:

  * AttModel._prepare_feature / get_logprobs_state / init_hidden (P/models/AttModel.py:94-117,158-165), the calls
    CaptionModel.beam_search (:172) and eval_ensemble.py drive a model with -- here a host-side beam loop (the oracle's
    restatement of CaptionModel.beam_search) runs on top of them and must reproduce the reference's beam-search goldens
    token for token;
  * model.done_beams[k]: the reference's full sorted list of finished beams (:174-176, AttModel.py:191-193);
  * the argparse Namespace of the reference's own opts.parse_opt() (tests/golden/opts_*.json, made by make_golden_opts.py):
    models.setup, Trainer and Optim are built from it as train.py builds them, and a training step runs."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import topdown as O
from test_gpu_topdown import BEAM_TAGS, BN_FIXTURES, FIXTURES, absmax, build_model, final_logit, rel

pytestmark = pytest.mark.gpu


def _with_bn_stats(W, X):
    W = dict(W)
    for k, v in X.items():
        if k.startswith("bnstat::"):
            W[k.split("::", 1)[1]] = torch.as_tensor(v)
    return W


@pytest.mark.parametrize("name", FIXTURES + BN_FIXTURES)
def test_prepare_feature_and_get_logprobs_state_drive_a_host_beam_search(name):
    cfg, W, I, Out, G, X = load_golden(name)
    model = build_model(cfg, _with_bn_stats(W, X), "f32").eval()
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    fc, att = I["fc_feats"][idx].cuda(), I["att_feats"][idx].cuda()
    am = I["att_masks"][idx].cuda() if "att_masks" in I else None
    p_fc, p_att, pp_att, p_am = model._prepare_feature(fc, att, am)
    Rc = p_att.shape[1]
    assert p_fc.dtype == torch.float32 and p_att.shape == (cfg["n_img"], Rc, cfg["H"]) and pp_att.shape == (cfg["n_img"], Rc, cfg["A"])
    if am is not None:
        assert Rc == int(am.sum(1).max()) and torch.equal(p_am, am[:, :Rc])          # clip_att
    L = cfg["L"]
    for tag in BEAM_TAGS:
        bs, dc, mp, eos_bias = [float(x) for x in X["beam::%s_cfg" % tag]]
        B = int(bs)
        with torch.no_grad():
            final_logit(model).bias[0] += eos_bias
        seqs, lps = [], []
        for k in range(cfg["n_img"]):                 # AttModel._sample_beam's per-image loop (:181-194)
            state = model.init_hidden(B)
            t_fc = p_fc[k:k + 1].expand(B, -1).contiguous()
            t_att = p_att[k:k + 1].expand(B, -1, -1).contiguous()
            t_patt = pp_att[k:k + 1].expand(B, -1, -1).contiguous()
            t_am = p_am[k:k + 1].expand(B, -1).contiguous() if p_am is not None else None

            def step_fn(it, st):
                lp, st2 = model.get_logprobs_state(it.cuda(), t_fc, t_att, t_patt, t_am, tuple(s_.cuda() for s_ in st))
                return lp.cpu(), tuple(s_.cpu() for s_ in st2)
            logprobs, state = step_fn(torch.zeros(B, dtype=torch.long), tuple(s_.cpu() for s_ in state))
            done = O.beam_search_core(step_fn, logprobs, state, L, B, int(dc), int(mp))
            seqs.append(done[0]["seq"])
            lps.append(done[0]["logps"])
        with torch.no_grad():
            final_logit(model).bias[0] -= eos_bias
        assert torch.equal(torch.stack(seqs), torch.as_tensor(X["beam::%s_seq" % tag])), tag
        assert absmax(torch.stack(lps), torch.as_tensor(X["beam::%s_logp" % tag])) < 1e-3, tag


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_get_logprobs_state_equals_the_teacher_forced_forward(dtype):
    """Stepping get_logprobs_state over the label tokens reproduces _forward's log-probs (eval mode)."""
    cfg, W, I, Out, G, X = load_golden("topdown_tiny_ragged")
    model = build_model(cfg, W, dtype).eval()
    fc, att, labels, am = (I[k].cuda() for k in ("fc_feats", "att_feats", "labels", "att_masks"))
    with torch.no_grad():
        ref = model(fc, None, att, labels, am)
    p_fc, p_att, pp_att, p_am = model._prepare_feature(fc, att, am)
    state = model.init_hidden(fc.shape[0])
    t_run = model._steps_to_run(labels)
    for t in range(t_run):
        lp, state = model.get_logprobs_state(labels[:, t], p_fc, p_att, pp_att, p_am, state)
        assert absmax(lp, ref[:, t]) < (2e-5 if dtype == "f32" else 2e-2), t
    assert state[0].shape == (2, fc.shape[0], cfg["H"]) and state[1].shape == (2, fc.shape[0], cfg["H"])


@pytest.mark.parametrize("name", FIXTURES[:3])
def test_done_beams_is_the_full_sorted_list(name):
    cfg, W, I, Out, G, X = load_golden(name)
    model = build_model(cfg, W, "f32").eval()
    idx = torch.arange(cfg["n_img"]) * cfg["S"]
    fc, att = I["fc_feats"][idx].cuda(), I["att_feats"][idx].cuda()
    am = I["att_masks"][idx].cuda() if "att_masks" in I else None
    Wm = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for B, eos_bias, mp in ((3, 0.0, 0), (4, 2.0, 1), (2, 4.0, 0)):
        with torch.no_grad():
            final_logit(model).bias[0] += eos_bias
            Wm[[k for k in Wm if k.startswith("logit") and k.endswith("bias")][-1]][0] += eos_bias
        seq, lp = model(fc, None, att, am, opt={"sample_max": 1, "beam_size": B, "max_ppl": mp}, mode="sample")
        _, _, beams = O.sample_beam(Wm, fc.cpu(), att.cpu(), am.cpu() if am is not None else None, cfg["L"], B, 0, mp,
                                    use_bn=cfg["use_bn"], return_beams=True)
        with torch.no_grad():
            final_logit(model).bias[0] -= eos_bias
            Wm[[k for k in Wm if k.startswith("logit") and k.endswith("bias")][-1]][0] -= eos_bias
        assert len(model.done_beams) == cfg["n_img"]
        for k in range(cfg["n_img"]):
            got, ref = model.done_beams[k], beams[k]
            assert len(got) == len(ref) <= B and len(got) >= 1
            assert torch.equal(got[0]["seq"], seq[k])                                  # the first beam is what _sample_beam returns
            for g_, r_ in zip(got, ref):
                assert torch.equal(g_["seq"].cpu(), r_["seq"])
                assert absmax(g_["logps"], r_["logps"]) < 1e-3 and abs(g_["p"] - r_["p"]) < 1e-3
                assert abs(g_["unaug_p"] - float(r_["logps"].sum())) < 1e-3
            assert all(got[i]["p"] >= got[i + 1]["p"] for i in range(len(got) - 1))


# ---------------------------------------------------------------- the reference's own option Namespace
def _ref_opt(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return argparse.Namespace(**json.load(f))


def test_reference_default_options_name_a_model_outside_the_hot_path():
    """parse_opt() with no flags selects caption_model = 'transformer' (out of scope): setup says so."""
    from unpaired_image_captioning_amd import models
    opt = _ref_opt("opts_default")
    assert opt.caption_model == "transformer" and opt.rnn_size == 1300 and opt.use_bn == 1 and opt.use_box == 1
    opt.vocab_size, opt.seq_length = 50, 8
    with pytest.raises(Exception, match="not supported by the MI355X hot path"):
        models.setup(opt)


def test_models_trainer_and_optim_from_the_reference_option_namespace():
    """opts_topdown512.json = the reference's parse_opt() for `--caption_model topdown --rnn_size 512`, every other flag at its
    default (use_bn 1, use_box 1 -> att_feat_size 2053, logit_layers 1, drop_prob_lm 0.5, seq_per_img 5, batch_size 4, Adam
    4e-4, nmt_max_grad_norm 5, seed -1 ...).  Completed the way train.py does (:22-26), it must build the captioner, the
    Trainer and the Optim and run training steps whose loss falls."""
    from unpaired_image_captioning_amd import models
    from unpaired_image_captioning_amd.misc.optimizer import Optim
    from unpaired_image_captioning_amd.trainer import Trainer
    opt = _ref_opt("opts_topdown512")
    assert (opt.caption_model, opt.rnn_size, opt.use_bn, opt.use_box, opt.logit_layers, opt.nmt_max_grad_norm) == ("topdown", 512, 1, 1, 1, 5)
    if opt.use_box:
        opt.att_feat_size = opt.att_feat_size + 5            # P/train.py:22
    opt.vocab_size, opt.seq_length = 120, 10                 # P/train.py:25-26 (from the loader)
    opt.start_from = None                                    # (the default names a checkpoint directory of the authors')
    model = models.setup(opt)
    sd = model.state_dict()
    assert sd["att_embed.0.running_mean"].shape == (2053,) and sd["att_embed.1.weight"].shape == (512, 2053)      # use_bn = 1
    assert sd["core.att_lstm.weight_ih"].shape == (2048, 512 + 2 * 512) and sd["logit.weight"].shape == (121, 512)
    tr = Trainer(opt)
    tr.build_optimizer()
    optim = Optim(opt)
    assert optim.i2t_lr == 4e-4 and optim.nmt_max_grad_norm == 5 and optim.i2t_train_flag == 1 and optim.nmt_train_flag == 0
    n_img, S, R = opt.batch_size, opt.seq_per_img, 7
    g = torch.Generator().manual_seed(3)
    att = torch.rand(n_img, R, opt.att_feat_size, generator=g)
    data = {"fc_feats": np.repeat(att.mean(1)[:, :opt.fc_feat_size].numpy(), S, 0),
            "att_feats": np.repeat(att.numpy(), S, 0),
            "att_masks": np.ones((n_img * S, R), dtype=np.float32),
            "labels": np.zeros((n_img * S, opt.seq_length + 2), dtype=np.int64),
            "masks": np.zeros((n_img * S, opt.seq_length + 2), dtype=np.float32)}
    toks = torch.randint(1, opt.vocab_size + 1, (n_img * S, 6), generator=g).numpy()
    data["labels"][:, 1:7] = toks
    data["masks"][:, :8] = 1.0
    losses = []
    for _ in range(8):
        tr.train(data)
        losses.append(float(tr.i2t_train_loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
