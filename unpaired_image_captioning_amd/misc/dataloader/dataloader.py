"""DataLoader of the reference (P/misc/dataloader/dataloader.py, P = pivot_based_eccv2018) for the MI355X path: same
constructor (`DataLoader(opt, train=True)`), same on-disk formats, same iteration state (`iterators`, `split_ix`, shuffle
at wrap-around, caption sampling from the global `random` stream), same batch dict -- but the per-image numpy work of
`__getitem__` (:302-331: L2 norm, box features, sort by box area) and the padding of `get_batch` (:270-283) run ON THE
DEVICE in one kernel (csrc/loader.hip, `uic_att_batch_assemble`, bit-identical to numpy), and the features are shipped
ONCE PER IMAGE: `att_feats [n_img, Rmax, att_feat_size]`, `fc_feats [n_img, fc_feat_size]`, `att_masks [n_img, Rmax]` are
device tensors with one row per image, `labels` / `masks` [n_img * seq_per_img, L + 2] numpy arrays as in the reference.
The captioner replicates on the device (uic_topdown_dims.seq_per_img); `reference_layout(data)` gives the reference's
replicated numpy arrays where a caller wants them.

What the host does: the library's thread team (csrc/loader_io.hip: uic_loader_scan / uic_loader_read) reads the files --
.npy, .npz stored or deflated -- STRAIGHT INTO a pinned staging buffer at the offsets the kernel wants, the next batch's
on a read-ahead thread while this one trains; then one asynchronous H2D copy per array.  Nothing here falls back to a CPU computation: without the
HIP library the constructor raises.

On-disk formats:
  opt.input_json         {'ix_to_word': {'1': ..}, 'images': [{'id', 'file_path', 'split', 'height', 'width'}]}
  opt.input_att_dir      <id>.npz with `feat` [R_i, D] f32 (scripts/make_bu_data.py:55)
  opt.input_fc_dir       <id>.npz with `feat` [Dfc] f32                                  (:331)
  opt.input_box_dir      <id>.npy [R_i, 4] f32 (x1, y1, x2, y2)
  opt.input_label_h5     HDF5 with `labels` uint32 [M, L], `label_start_ix` / `label_end_ix` (1-indexed), read through
                         h5py when it is installed, else through the HDF5 C library (ctypes); an .npz with the same array
                         names is accepted as well.
  opt.input_nmt_h5       (nmt_train_flag / nmt_eval_flag) the NMT corpus, batched by onmt_dataset_h5 in this package;
                         `data['nmt']` and `bounds['wrapper_nmt']` as :200-207,292-296.
Not served here: use_box_cls_prob (`attri_feats`, unused by the TopDown / FC captioners: the entry is None), the .pt corpus
(input_nmt_choice = 0) and the onmt.Dict vocabularies (only printed by the reference's loader).
"""
import ctypes as C
import json
import os
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ... import _lib
from ..._lib import check, ptr, stream
from .label_store import NMT_NAMES, open_label_store
from .onmt_dataset_h5 import onmt_dataset_h5


def padded_width(D):
    """Row stride the captioner's library wants for att_feat_size D (TopDownEngine: a multiple of 8 as is, else the next
    multiple of 128): the assembly kernel zero-fills up to it, so no second padding pass is needed."""
    return D if D % 8 == 0 else (D + 127) // 128 * 128


def cpu_budget():
    """CPUs this process can keep busy: the affinity mask, cut down to the cgroup's CPU quota when it has one (cgroup v2
    cpu.max "quota period", v1 cpu.cfs_quota_us / cpu.cfs_period_us)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 8
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0 and per > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


class DataLoader(object):

    def reset_iterator(self, split):                                   # :26-30
        self.iterators[split] = 0

    def get_vocab_size(self):
        return self.vocab_size

    def get_vocab(self):
        return self.ix_to_word

    def get_seq_length(self):
        return self.seq_length

    def __init__(self, opt, train=True, device="cuda", read_threads=None, read_ahead=True, rank=0, world_size=1):
        self.lib = _lib.load()                                         # raises when the HIP library is missing
        # one process per GPU: every rank walks the SAME global sequence (same `random` seed: same shuffles, same
        # caption draws) over world_size * batch_size images per step and keeps images [rank * B, (rank + 1) * B) of it
        self.rank, self.world_size = int(rank), int(world_size)
        assert 0 <= self.rank < self.world_size
        self.device = torch.device(device)
        self.opt = opt
        self.batch_size = self.opt.batch_size
        self.nmt_train_flag = getattr(opt, 'nmt_train_flag', 0)
        self.nmt_eval_flag = getattr(opt, 'nmt_eval_flag', 0)
        self.fc_feat_size = opt.fc_feat_size
        self.att_feat_size = opt.att_feat_size
        self.seq_per_img = opt.seq_per_img
        self.type = train
        self.use_att = getattr(opt, 'use_att', True)
        self.use_box = getattr(opt, 'use_box', 0)
        self.use_box_cls_prob = 0                                      # see the module docstring
        self.norm_att_feat = getattr(opt, 'norm_att_feat', 0)
        self.norm_box_feat = getattr(opt, 'norm_box_feat', 0)

        with open(self.opt.input_json) as f:
            self.info = json.load(f)
        self.ix_to_word = self.info['ix_to_word']
        self.vocab_size = len(self.ix_to_word)

        store = open_label_store(self.opt.input_label_h5)
        self.labels = np.ascontiguousarray(store['labels'])            # [M, L] (uint32 on disk)
        self.seq_length = self.labels.shape[1]
        self.label_start_ix = np.asarray(store['label_start_ix']).astype(np.int64)       # 1-indexed
        self.label_end_ix = np.asarray(store['label_end_ix']).astype(np.int64)
        self.num_images = self.label_start_ix.shape[0]

        self.input_fc_dir = self.opt.input_fc_dir
        self.input_att_dir = self.opt.input_att_dir
        self.input_box_dir = getattr(self.opt, 'input_box_dir', None)

        self.split_ix = {'train': [], 'val': [], 'test': []}           # :100-110
        for ix, img in enumerate(self.info['images']):
            if img['split'] in self.split_ix:
                self.split_ix[img['split']].append(ix)
            elif getattr(opt, 'train_only', 0) == 0:                   # restval
                self.split_ix['train'].append(ix)
        self.iterators = {'train': 0, 'val': 0, 'test': 0}
        self.nmt_batchIdx = 0
        self.nmt_trainData = self.nmt_validData = None
        if self.nmt_train_flag or self.nmt_eval_flag and self.type:      # :78 (the reference's own precedence)
            corpus = open_label_store(self.opt.input_nmt_h5, NMT_NAMES)
            on_device = self.device.type == "cuda"
            self.nmt_trainData = onmt_dataset_h5(corpus, 'train', opt.batch_size, on_device)
            self.nmt_validData = onmt_dataset_h5(corpus, 'valid', opt.batch_size, on_device, volatile=True)
            self.batchOrder = torch.randperm(len(self.nmt_trainData))   # :138 (drawn, never used, by the reference too)

        # make_bu_data.py:56 writes the fc vectors as <id>.npy, the reference's loader reads <id>.npz['feat'] (:331, the
        # .npy line is commented out at :330): take whichever the data set has
        first = str(self.info['images'][0]['id'])
        self._fc_ext = '.npz' if os.path.exists(os.path.join(self.input_fc_dir, first + '.npz')) else '.npy'
        # the library's reader team (a persistent pool since round 6): 32 workers move stored members at the page cache's rate;
        # DEFLATED members (np.savez_compressed, what make_bu_data.py writes) cost ~0.55 ms of CPU each with the library's own decoder
        # (csrc/inflate_fast.h, two streams per thread; zlib: ~0.95 ms) and get up to 128
        # -- both capped by the CPUs this process may actually USE (affinity mask and cgroup quota): a team larger than a container's
        # CPU quota gets the whole process frozen until the next scheduling period once the quota is spent (measured on a 16-CPU
        # pod of a 256-thread host: 64-128 inflating threads = 3 ms batches with 20-170 ms stalls every few batches, 16 threads =
        # a steady 7.5 ms -- the same average, tools/loader_bench.py --compressed --read-threads N)
        cpus = cpu_budget()
        self.read_threads = int(read_threads or max(1, min(32, cpus)))
        # (two CPUs are left to the training loop's own host threads: with all 16 of a 16-CPU quota inflating, batches took 4.6 ms
        # but every few of them 10-15 ms; 14 threads: a steady 5.5 ms)
        self.read_threads_deflate = int(read_threads or max(1, min(128, cpus - 2)))
        self.read_ahead = read_ahead
        self._pool = ThreadPoolExecutor(max_workers=1)                 # the read-ahead thread (the team is inside the library)
        self._ahead_job = None
        self._pin = {}
        self._pin_turn = 0

    # ------------------------------------------------------------------ iteration state (BlobFetcher, :373-387)
    def _next_index(self, split):
        max_index = len(self.split_ix[split])
        wrapped = False
        ri = self.iterators[split]
        ix = self.split_ix[split][ri]
        ri_next = ri + 1
        if ri_next >= max_index:
            ri_next = 0
            if split == 'train':
                random.shuffle(self.split_ix[split])
            wrapped = True
        self.iterators[split] = ri_next
        return ix, wrapped

    def get_captions(self, ix, seq_per_img):                           # :181-198
        ix1 = int(self.label_start_ix[ix]) - 1
        ix2 = int(self.label_end_ix[ix]) - 1
        ncap = ix2 - ix1 + 1
        assert ncap > 0, 'an image does not have any label. this can be handled but right now isn\'t'
        if ncap < seq_per_img:
            seq = np.zeros([seq_per_img, self.seq_length], dtype='int')
            for q in range(seq_per_img):
                ixl = random.randint(ix1, ix2)
                seq[q, :] = self.labels[ixl, :self.seq_length]
        else:
            ixl = random.randint(ix1, ix2 - seq_per_img + 1)
            seq = self.labels[ixl: ixl + seq_per_img, :self.seq_length]
        return seq

    def get_nmt_batch(self, split):                                    # :200-207
        wrapped_nmt = False
        self.nmt_batch = self.nmt_trainData[self.nmt_batchIdx]
        if self.nmt_batchIdx + 1 < self.nmt_trainData.numBatches:
            self.nmt_batchIdx = self.nmt_batchIdx + 1
        else:
            wrapped_nmt = True
            self.nmt_batchIdx = 0
        return self.nmt_batch, wrapped_nmt

    # ------------------------------------------------------------------ files -> pinned staging (host only)
    def _paths(self, indices):
        ids = [str(self.info['images'][ix]['id']) for ix in indices]
        fc = [os.path.join(self.input_fc_dir, i + self._fc_ext).encode() for i in ids]
        att = [os.path.join(self.input_att_dir, i + '.npz').encode() for i in ids] if self.use_att else []
        box = [os.path.join(self.input_box_dir, i + '.npy').encode() for i in ids] if self.use_att and self.use_box else []
        return fc, att, box

    def _scan(self, paths, member):
        arr = (C.c_char_p * len(paths))(*paths)
        info = np.empty((len(paths), 6), dtype=np.int64)
        check(self.lib.uic_loader_scan(arr, len(paths), member, info.ctypes.data, self.read_threads), "loader_scan")
        return arr, info

    def _stage(self, indices, turn):
        """Files of the images `indices` (fetch order) -> pinned staging buffers, laid out as uic_att_batch_assemble wants
        them; returns what _ship needs.  Host work only (library calls release the GIL): runs on the read-ahead thread."""
        n_img = len(indices)
        fc_p, att_p, box_p = self._paths(indices)
        # ONE scan for every file of the batch (a thread team is started per library call; `member` only matters for zips)
        _, info_all = self._scan(fc_p + att_p + box_p, b"feat")
        fc_info, att_info, box_info = (info_all[:n_img], info_all[n_img:n_img + len(att_p)], info_all[n_img + len(att_p):])
        Dfc = int(fc_info[0, 1] * fc_info[0, 2])
        if not (fc_info[:, 1] * fc_info[:, 2] == Dfc).all():
            raise ValueError("fc feature files of different sizes in one batch")
        st = {"n_img": n_img, "turn": turn}
        if self.use_att:
            if not (att_info[:, 0] == 2).all() or not (att_info[:, 2] == att_info[0, 2]).all():
                raise ValueError("att feature files must hold [regions, %d] arrays" % att_info[0, 2])
            counts = [int(c) for c in att_info[:, 1]]
            D = int(att_info[0, 2])
        else:
            counts, D = [1] * n_img, 0
        order = sorted(range(n_img), key=lambda i: counts[i], reverse=True)               # :264-265, stable
        slot_of = np.empty(n_img, dtype=np.int32)
        slot_of[order] = np.arange(n_img, dtype=np.int32)
        start = np.zeros(n_img + 1, dtype=np.int32)
        start[1:] = np.cumsum(counts)
        total = int(start[-1])
        st.update(counts=counts, order=order, slot_of=slot_of, start=start, D=D, Dfc=Dfc)

        st["fc"] = fc_h, _ = self._staging("fc", (n_img, Dfc), torch.float32, turn)
        paths, infos = list(fc_p), [fc_info]
        dsts = [fc_h.data_ptr() + int(slot_of[i]) * Dfc * 4 for i in range(n_img)]       # fc rows land in batch order
        if self.use_att:
            st["feat"] = feat_h, _ = self._staging("feat", (total, D), torch.float32, turn)
            paths += att_p
            infos.append(att_info)
            dsts += [feat_h.data_ptr() + int(start[i]) * D * 4 for i in range(n_img)]
            st["meta"] = meta_h, _ = self._staging("meta", (2 * n_img + 1,), torch.int32, turn)
            meta_h.numpy()[:n_img + 1] = start
            meta_h.numpy()[n_img + 1:] = slot_of
            if self.use_box:
                if not (box_info[:, 1] == att_info[:, 1]).all() or not (box_info[:, 2] == 4).all():
                    raise ValueError("box files must hold [regions, 4] arrays with the regions of the att features")
                st["box"] = box_h, _ = self._staging("box", (total * 4 + n_img * 3,), torch.float32, turn)
                paths += box_p
                infos.append(box_info)
                dsts += [box_h.data_ptr() + int(start[i]) * 16 for i in range(n_img)]
                hw = box_h.numpy()[total * 4:].reshape(n_img, 3)
                for i, ix in enumerate(indices):
                    img = self.info['images'][ix]
                    h, w = img['height'], img['width']
                    hw[i] = (np.float32(h), np.float32(w), np.float32(w * h))
        arr = (C.c_char_p * len(paths))(*paths)
        info = np.ascontiguousarray(np.concatenate(infos, 0))
        dst = (C.c_void_p * len(dsts))(*dsts)
        deflated = bool((info[:, 4] == 8).any())
        check(self.lib.uic_loader_read(arr, len(paths), info.ctypes.data, dst, self.read_threads_deflate if deflated else self.read_threads),
              "loader_read")
        return st

    def _staging(self, key, shape, dtype, turn):
        """Pinned host buffer (two per key, alternating: the copy out of the previous batch's may still be in flight)."""
        n = int(np.prod(shape))
        slot = self._pin.setdefault((key, turn), [None, None])
        if slot[0] is None or slot[0].numel() < n or slot[0].dtype != dtype:
            slot[0] = torch.empty(max(n, 1), dtype=dtype, pin_memory=self.device.type == "cuda")
        elif slot[1] is not None:
            slot[1].synchronize()
        return slot[0][:n].view(shape), slot

    def _to_device(self, host_and_slot):
        host_view, slot = host_and_slot
        dev = host_view.to(self.device, non_blocking=True)
        if self.device.type == "cuda":
            slot[1] = torch.cuda.Event()
            slot[1].record()
        return dev

    def _staged(self, indices):
        """The staged files of this batch: from the read-ahead thread when it guessed the indices, else read now."""
        ahead, self._ahead_job = self._ahead_job, None
        if ahead is not None:
            want, fut = ahead
            st = fut.result()
            if want == list(indices):
                return st
        self._pin_turn ^= 1
        return self._stage(list(indices), self._pin_turn)

    def _read_ahead(self, split, count):
        """Start staging the files the next get_batch(split) will ask for (known unless the epoch wraps before)."""
        ri = self.iterators[split]
        glob = self.split_ix[split][ri: ri + count * self.world_size]
        nxt = list(glob[self.rank * count: (self.rank + 1) * count]) if len(glob) == count * self.world_size else []
        if len(nxt) == count:
            self._pin_turn ^= 1
            self._ahead_job = (nxt, self._pool.submit(self._stage, nxt, self._pin_turn))

    # ------------------------------------------------------------------ the batch
    def get_batch(self, split, batch_size=None, seq_per_img=None):
        """:209-299.  Features once per image, on the device (module docstring); everything else as the reference."""
        batch_size = batch_size or self.batch_size
        S = seq_per_img or self.seq_per_img
        L = self.seq_length
        label_batch = np.zeros([batch_size * S, L + 2], dtype='int')
        mask_batch = np.zeros([batch_size * S, L + 2], dtype='float32')
        wrapped = False
        indices, infos, gts = [], [], []
        for g in range(batch_size * self.world_size):
            ix, w = self._next_index(split)                            # may reshuffle: BEFORE the caption draw, as :236,247
            seq = self.get_captions(ix, S)
            wrapped = wrapped or w
            i = g - self.rank * batch_size
            if i < 0 or i >= batch_size:
                continue                                               # another rank's image (its draws were consumed above)
            indices.append(ix)
            label_batch[i * S:(i + 1) * S, 1:L + 1] = seq
            gts.append(self.labels[self.label_start_ix[ix] - 1: self.label_end_ix[ix]])
            img = self.info['images'][ix]
            infos.append({'ix': ix, 'id': img['id'], 'file_path': img['file_path']})
        st = self._staged(indices)
        order = st["order"]
        label_batch = np.vstack([label_batch[i * S:(i + 1) * S] for i in order])
        gts = [gts[i] for i in order]
        infos = [infos[i] for i in order]

        data = {}
        data['fc_feats'], data['att_feats'], data['att_masks'] = self._ship(st)
        if self.read_ahead:                 # (after a wrap the split is already reshuffled: the peek below sees the new order)
            self._read_ahead(split, batch_size)
        data['attri_feats'] = None
        data['labels'] = label_batch
        nonzeros = (label_batch != 0).sum(1) + 2                       # :287-290
        mask_batch[np.arange(L + 2)[None, :] < nonzeros[:, None]] = 1
        data['masks'] = mask_batch
        data['gts'] = gts
        data['nmt'], wrapped_nmt = self.get_nmt_batch('train') if self.nmt_train_flag and self.type else (None, False)
        data['bounds'] = {'it_pos_now': self.iterators[split], 'it_max': len(self.split_ix[split]), 'wrapped': wrapped,
                          'wrapper_nmt': wrapped_nmt}
        data['infos'] = infos
        data['seq_per_img'] = S
        return data

    def _ship(self, st):
        """Staged batch -> device: one asynchronous copy per array, then the assembly kernel."""
        n_img = st["n_img"]
        fc_d = self._to_device(st["fc"])
        if not self.use_att:
            return fc_d, torch.zeros(n_img, 1, 1, device=self.device), torch.ones(n_img, 1, device=self.device)
        D, total = st["D"], int(st["start"][-1])
        Rmax = max(st["counts"])
        feat_d = self._to_device(st["feat"])
        meta_d = self._to_device(st["meta"])
        start_d, slot_d = meta_d[:n_img + 1], meta_d[n_img + 1:]
        box_d = hw_d = None
        if self.use_box:
            bd = self._to_device(st["box"])
            box_d, hw_d = bd[:total * 4], bd[total * 4:]
        Dout = D + (5 if self.use_box else 0)
        ld = padded_width(Dout)
        att = torch.empty(n_img, Rmax, ld, dtype=torch.float32, device=self.device)
        masks = torch.empty(n_img, Rmax, dtype=torch.float32, device=self.device)
        check(self.lib.uic_att_batch_assemble(ptr(feat_d), ptr(box_d), ptr(start_d), ptr(hw_d), ptr(slot_d), n_img, D,
                                              int(bool(self.norm_att_feat)), int(bool(self.norm_box_feat)), Rmax, ld,
                                              ptr(att), ptr(masks), stream()), "att_batch_assemble")
        view = att[:, :, :Dout] if ld != Dout else att
        view._uic_zero_padded_ld = ld                                  # TopDownEngine._pad_att: already padded, no copy
        return fc_d, view, masks

    def __len__(self):
        return len(self.info['images'])


def reference_layout(data):
    """The batch dict with the reference's host arrays: features replicated seq_per_img times (:267-277), float32 numpy."""
    S = int(data['seq_per_img'])
    out = dict(data)
    for k in ('fc_feats', 'att_feats', 'att_masks'):
        out[k] = np.repeat(data[k].detach().cpu().numpy(), S, axis=0)
    return out
