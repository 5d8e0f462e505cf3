"""The captioner half of the reference's training driver (P/train.py:16-140 + the per-epoch preamble and the
best-checkpoint bookkeeping of P/trainer.py:98-104,150-160,200-212) on this package's DataLoader / Trainer / eval_split:
same `opt` fields, same epoch accounting (`bounds['wrapped']`), learning-rate and scheduled-sampling schedules, switch to
self-critical training at `self_critical_after`, validation every `save_checkpoint_every` iterations, and the same files
in `opt.checkpoint_path`: `model_i2t[-best].pth` (state_dict), `infos[-best].pkl` (`iter, epoch, iterators, split_ix,
best_val_score, opt, vocab`), `histories[-best].pkl`.  TensorBoard logging, the COCO side loader and the NMT half of the
loop are the reference's control plane and stay there.

    opt = <the reference's argparse Namespace>;  train_loop.main(opt, max_iterations=1000)
"""
import os
import pickle
import time

import torch

from . import eval_utils
from .misc import utils
from .misc.dataloader.dataloader import DataLoader
from .trainer import Trainer


def init(opt, rank=0, world_size=1):
    """P/train.py:16-43."""
    infos, histories = {}, {}
    if getattr(opt, 'seed', 0) > 0:
        torch.manual_seed(opt.seed)
    opt.use_att = utils.if_use_att(opt.caption_model)
    if getattr(opt, 'use_box', 0):
        opt.att_feat_size = opt.att_feat_size + 5
    loader = DataLoader(opt, rank=rank, world_size=world_size)
    opt.vocab_size = loader.vocab_size
    opt.seq_length = loader.seq_length
    start_from = getattr(opt, 'start_from', None)
    if start_from:
        with open(os.path.join(start_from, 'infos-best.pkl'), 'rb') as f:
            infos = pickle.load(f)
        saved = vars(infos['opt'])
        for checkme in ("rnn_type", "rnn_size", "num_layers"):
            assert saved.get(checkme) == vars(opt).get(checkme), \
                "Command line argument and saved model disagree on '%s' " % checkme
        hist = os.path.join(start_from, 'histories-best.pkl')
        if os.path.isfile(hist):
            with open(hist, 'rb') as f:
                histories = pickle.load(f)
    return opt, loader, infos, histories


def main(opt, max_iterations=None, exchange=None, rank=0, world_size=1, log=print):
    opt, loader, infos, histories = init(opt, rank, world_size)
    iteration = infos.get('iter', 0)
    epoch = infos.get('epoch', 0)
    loader.iterators = infos.get('iterators', loader.iterators)
    loader.split_ix = infos.get('split_ix', loader.split_ix)
    val_result_history = histories.get('val_result_history', {})
    loss_history = histories.get('loss_history', {})
    lr_history = histories.get('lr_history', {})
    ss_prob_history = histories.get('ss_prob_history', {})

    trainer = Trainer(opt, exchange)
    start_from = getattr(opt, 'start_from', None)
    if start_from and os.path.isfile(os.path.join(start_from, 'model_i2t-best.pth')):
        trainer.i2t_model.load_state_dict(torch.load(os.path.join(start_from, 'model_i2t-best.pth')))
    trainer.build_optimizer()
    best_val_score = infos.get('best_val_score', None)
    update_lr_flag = True
    fetch = lambda: loader.get_batch('train')
    data = fetch()
    while True:
        start = time.time()
        if update_lr_flag:                                             # P/trainer.py:151-160
            trainer.update_LearningRate(epoch)
            ss_start = getattr(opt, 'scheduled_sampling_start', -1)
            if epoch > ss_start and ss_start >= 0:                     # P/misc/optimizer.py:108-112
                frac = (epoch - ss_start) // opt.scheduled_sampling_increase_every
                trainer.i2t_model.ss_prob = min(opt.scheduled_sampling_increase_prob * frac, opt.scheduled_sampling_max_prob)
            sc_after = getattr(opt, 'self_critical_after', -1)
            trainer.sc_flag = sc_after != -1 and epoch >= sc_after
            update_lr_flag = False
        # the next batch is fetched (files, H2D, assembly kernel) after this step is enqueued, on the copy stream
        if trainer.sc_flag:
            trainer.train_self_critical(data, next_data=fetch)
        else:
            trainer.train(data, next_data=fetch)
        wrapped = data['bounds']['wrapped']
        data = trainer.next_data
        iteration += 1
        if wrapped:
            epoch += 1
            update_lr_flag = True

        if iteration % getattr(opt, 'losses_log_every', 25) == 0:
            loss_history[iteration] = trainer.i2t_train_loss if not trainer.sc_flag else trainer.i2t_avg_reward
            lr_history[iteration] = trainer.i2t_current_lr
            ss_prob_history[iteration] = trainer.i2t_model.ss_prob
            log("{}|{}/{}|I2T-{:.3f}%|TrainLoss:{:.2f}|TB:{:.3f}|".format(
                getattr(opt, 'id', ''), iteration, epoch,
                100 * float(loader.iterators['train']) / float(len(loader.split_ix['train'])), trainer.i2t_train_loss,
                time.time() - start))

        last = max_iterations is not None and iteration >= max_iterations
        if iteration % opt.save_checkpoint_every == 0 or last:
            eval_kwargs = {'split': 'val', 'verbose': False, 'verbose_beam': 0}
            eval_kwargs.update(vars(opt))
            eval_kwargs['dataset'] = opt.input_json
            train_iter = loader.iterators['train']                     # (validation batches come from another split: untouched)
            val_loss, predictions, lang_stats, _, _ = eval_utils.eval_split(opt, loader, trainer.i2t_model, None, eval_kwargs)
            assert loader.iterators['train'] == train_iter
            val_result_history[iteration] = {'loss': val_loss, 'lang_stats': lang_stats, 'predictions': predictions}
            current_score = lang_stats['CIDEr'] if getattr(opt, 'language_eval', 0) == 1 else -val_loss   # P/trainer.py:207
            best = best_val_score is None or current_score > best_val_score
            if best:
                best_val_score = current_score
            tag = '-best' if best else ''
            infos.update(iter=iteration, epoch=epoch, iterators=loader.iterators, split_ix=loader.split_ix,
                         best_val_score=best_val_score, opt=opt, vocab=loader.get_vocab())
            histories.update(val_result_history=val_result_history, loss_history=loss_history, lr_history=lr_history,
                             ss_prob_history=ss_prob_history)
            if rank == 0:
                trainer.save_models(tag)
                with open(os.path.join(opt.checkpoint_path, 'infos' + tag + '.pkl'), 'wb') as f:
                    pickle.dump(infos, f)
                with open(os.path.join(opt.checkpoint_path, 'histories' + tag + '.pkl'), 'wb') as f:
                    pickle.dump(histories, f)
            log("validation loss %.4f%s" % (val_loss, " (best)" if best else ""))
        if last or (getattr(opt, 'max_epochs', -1) != -1 and epoch >= opt.max_epochs):
            break
    return trainer, infos, histories
