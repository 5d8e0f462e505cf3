"""`eval_split` of the reference (P/eval_utils.py:194-313) for the captioner on the MI355X path: validation loss with the
reference's criterion and the generated caption of every image of a split (greedy / sampling / beam search through
`eval_kwargs`, as `model(..., opt=eval_kwargs, mode='sample')` takes them), read through this package's DataLoader --
whose batches already hold every image's features ONCE, on the device, so the reference's "only leave one feature for
each image" indexing (:243-247) and its host -> device copies disappear.

`language_eval = 1` needs the coco-caption scorers (Java METEOR / PTB tokenizer, P/eval_utils.py:27-80), which are outside
this package: pass `eval_kwargs['language_eval_fn'](predictions, split) -> dict` to plug a scorer in, otherwise it raises.
Returns `(loss_sum / loss_evals, predictions, lang_stats, 0.0, 0.0)` like the reference's i2t-only branch (:311-312).
"""
import numpy as np
import torch

from .misc import criterion
from .misc import utils


def eval_split(opt, loader, i2t_model, nmt_model=None, eval_kwargs={}):
    verbose = eval_kwargs.get('verbose', True)
    verbose_beam = eval_kwargs.get('verbose_beam', 1)
    verbose_loss = eval_kwargs.get('verbose_loss', 1)
    num_images = eval_kwargs.get('num_images', eval_kwargs.get('val_images_use', -1))
    split = eval_kwargs.get('split', 'val')
    lang_eval = eval_kwargs.get('language_eval', 0)
    beam_size = eval_kwargs.get('beam_size', 1)
    if getattr(opt, 'nmt_eval_flag', 0) or nmt_model is not None:
        raise NotImplementedError("eval_split here evaluates the captioner; the NMT validation statistics come from NMT_loss(eval=True)")
    if lang_eval == 1 and 'language_eval_fn' not in eval_kwargs:
        raise NotImplementedError("language_eval = 1 needs the coco-caption scorers: pass eval_kwargs['language_eval_fn']")

    i2t_crit = criterion.LanguageModelCriterion(opt)
    was_training = i2t_model.training
    i2t_model.eval()
    loader.reset_iterator(split)

    n = 0
    loss = 0
    loss_sum = 0
    loss_evals = 1e-8
    predictions = []
    device = next(i2t_model.parameters()).device
    while True:
        data = loader.get_batch(split)
        n = n + loader.batch_size
        fc_feats, att_feats, att_masks = data['fc_feats'], data['att_feats'], data['att_masks']      # one row per image, on the device
        with torch.no_grad():
            if data.get('labels', None) is not None and verbose_loss:
                labels = torch.from_numpy(np.ascontiguousarray(data['labels'])).to(device)
                masks = torch.from_numpy(np.ascontiguousarray(data['masks'])).to(device)
                outputs = i2t_model(fc_feats, None, att_feats, labels, att_masks)                   # :233-236
                loss = i2t_crit(outputs, labels[:, 1:], masks[:, 1:]).item()
                loss_sum = loss_sum + loss
                loss_evals = loss_evals + 1
            seq = i2t_model(fc_feats, None, att_feats, att_masks, opt=eval_kwargs, mode='sample')[0]   # :251
        if beam_size > 1 and verbose_beam:
            for i in range(loader.batch_size):
                print('\n'.join([utils.decode_sequence(loader.get_vocab(), _['seq'].unsqueeze(0))[0] for _ in i2t_model.done_beams[i]]))
                print('--' * 10)
        sents = utils.decode_sequence(loader.get_vocab(), seq)
        for k, sent in enumerate(sents):
            if verbose:
                print('image %s: ' % (data['infos'][k]['id']), sent)
            entry = {'image_id': data['infos'][k]['id'], 'caption': sent}
            if eval_kwargs.get('dump_path', 0) == 1:
                entry['file_name'] = data['infos'][k]['file_path']
            predictions.append(entry)

        ix0 = data['bounds']['it_pos_now']                             # :277-289: drop what ran past the split / the budget
        ix1 = data['bounds']['it_max']
        if num_images != -1:
            ix1 = min(ix1, num_images)
        for i in range(n - ix1):
            predictions.pop()
        if verbose:
            print('evaluating validation preformance... %d/%d (%f)' % (ix0 - 1, ix1, loss))
        if data['bounds']['wrapped']:
            break
        if num_images >= 0 and n >= num_images:
            break

    lang_stats = eval_kwargs['language_eval_fn'](predictions, split) if lang_eval == 1 else None
    if was_training:
        i2t_model.train()
    return loss_sum / loss_evals, predictions, lang_stats, 0.0, 0.0
