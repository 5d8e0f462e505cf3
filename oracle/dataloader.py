"""CPU restatement of the reference's batch assembly -- TEST INFRASTRUCTURE ONLY (imported by tests/, smoke() and the
cpu_baseline leg of bench.py; never by the product path).

Follows P/misc/dataloader/dataloader.py (P = /root/reference/pivot_based_eccv2018):
  region_features  <- DataLoader.__getitem__   :302-331  (L2 norm, box features, sort by box area)
  get_captions     <- DataLoader.get_captions  :181-198  (uses the `random` module exactly like the reference)
  merge_batch      <- DataLoader.get_batch     :209-299  (sort by region count, S-fold replication, padding, masks)
  next_index       <- BlobFetcher._get_next_minibatch_inds :373-387

Pinned: tests/golden/dataloader_*.npz hold what the reference's own DataLoader methods returned for the same raw
arrays (tests/golden/make_golden_dataloader.py); tests/test_oracle_dataloader.py checks this file against them bit for
bit.  numpy arithmetic is float32 throughout, as in the reference (f32 feature / box files, python-int image sizes).
"""
import random

import numpy as np


def region_features(att_feat, box_feat=None, height=None, width=None, norm_att_feat=0, norm_box_feat=0):
    """att_feat [R, D] f32 (the `feat` array of one image's .npz), box_feat [R, 4] f32 or None -> [R, D(+5)] f32."""
    if norm_att_feat:
        att_feat = att_feat / np.linalg.norm(att_feat, 2, 1, keepdims=True)                       # :310-311
    if box_feat is not None:
        x1, y1, x2, y2 = np.hsplit(box_feat, 4)                                                   # :320
        h, w = height, width
        box_feat = np.hstack((x1 / w, y1 / h, x2 / w, y2 / h, (x2 - x1) * (y2 - y1) / (w * h)))    # :322
        if norm_box_feat:
            box_feat = box_feat / np.linalg.norm(box_feat, 2, 1, keepdims=True)                   # :323-324
        att_feat = np.hstack([att_feat, box_feat])                                                # :325
        att_feat = np.stack(sorted(att_feat, key=lambda x: x[-1], reverse=True))                  # :327
    return att_feat


def get_captions(labels, label_start_ix, label_end_ix, ix, seq_per_img, seq_length):
    """:181-198 -- draws from the global `random` stream in the reference's order."""
    ix1 = label_start_ix[ix] - 1
    ix2 = label_end_ix[ix] - 1
    ncap = ix2 - ix1 + 1
    assert ncap > 0, "an image does not have any label"
    if ncap < seq_per_img:
        seq = np.zeros([seq_per_img, seq_length], dtype='int')
        for q in range(seq_per_img):
            ixl = random.randint(ix1, ix2)
            seq[q, :] = labels[ixl, :seq_length]
    else:
        ixl = random.randint(ix1, ix2 - seq_per_img + 1)
        seq = labels[ixl: ixl + seq_per_img, :seq_length]
    return seq


def next_index(split_ix, iterators, split, shuffle):
    """:373-387 -> (image index, wrapped)."""
    max_index = len(split_ix[split])
    wrapped = False
    ri = iterators[split]
    ix = split_ix[split][ri]
    ri_next = ri + 1
    if ri_next >= max_index:
        ri_next = 0
        if shuffle:
            random.shuffle(split_ix[split])
        wrapped = True
    iterators[split] = ri_next
    return ix, wrapped


def merge_batch(fc_batch, att_batch, label_rows, gts, infos, seq_per_img, seq_length):
    """The second half of get_batch (:263-299).  fc_batch / att_batch: per-image arrays in fetch order; label_rows
    [n_img * S, seq_length] the get_captions() results stacked in fetch order."""
    batch_size = len(fc_batch)
    S = seq_per_img
    label_batch = np.zeros([batch_size * S, seq_length + 2], dtype='int')
    mask_batch = np.zeros([batch_size * S, seq_length + 2], dtype='float32')
    label_batch[:, 1:seq_length + 1] = label_rows
    order = sorted(range(batch_size), key=lambda i: len(att_batch[i]), reverse=True)             # :264-265 (stable)
    fc_batch = [fc_batch[i] for i in order]
    att_batch = [att_batch[i] for i in order]
    label_split = [np.vsplit(label_batch, batch_size)[i] for i in order]
    gts = [gts[i] for i in order]
    infos = [infos[i] for i in order]
    data = {}
    data['fc_feats'] = np.stack([f for f in fc_batch for _ in range(S)])                         # :267
    max_att_len = max(a.shape[0] for a in att_batch)
    data['att_feats'] = np.zeros([batch_size * S, max_att_len, att_batch[0].shape[1]], dtype='float32')
    data['att_masks'] = np.zeros(data['att_feats'].shape[:2], dtype='float32')
    for i, a in enumerate(att_batch):
        data['att_feats'][i * S:(i + 1) * S, :a.shape[0]] = a                                    # :273-274
        data['att_masks'][i * S:(i + 1) * S, :a.shape[0]] = 1                                    # :276-277
    data['labels'] = np.vstack(label_split)
    nonzeros = np.array([(row != 0).sum() + 2 for row in data['labels']])                        # :281
    for ix, row in enumerate(mask_batch):
        row[:nonzeros[ix]] = 1
    data['masks'] = mask_batch
    data['gts'] = gts
    data['infos'] = infos
    return data
