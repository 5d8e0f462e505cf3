"""The NMT corpus batcher of the reference (P/misc/dataloader/onmt_dataset_h5.py): batch `index` of the `train` / `valid`
split of the HDF5 corpus file (`{train,valid}_{src,tgt}_label[_length]`, scripts/prepro_aic_nmt.py:435-448) as the
time-major tensors the NMT model takes -- `src [S, n, 1]`, `tgt [T, n]` (int64, PAD = 0), `lengths [1, n]`, sentences
sorted by decreasing source length.  Integer index work only: it is done with array slicing on the host into pinned
memory and shipped with one asynchronous copy per tensor (the reference fills a float tensor row by row, :39-43).
"""
import math

import numpy as np
import torch

PAD = 0                                                                 # onmt.Constants.PAD


class Batch(object):
    """:120-141."""

    def __init__(self, src, tgt, lengths, indices, batchSize, alignment=None):
        self.src = src
        self.tgt = tgt
        self.lengths = lengths
        self.indices = indices
        self.batchSize = batchSize
        self.alignment = alignment

    def words(self):
        return self.src[:, :, 0]

    def features(self, j):
        return self.src[:, :, j + 1]

    def truncate(self, start, end):
        return Batch(self.src, self.tgt[start:end], self.lengths, self.indices, self.batchSize,
                     self.alignment[start:end] if self.alignment is not None else None)


class onmt_dataset_h5(object):

    def __init__(self, nmt_Data, split, batchSize, cuda, volatile=False, data_type="text", srcFeatures=None,
                 tgtFeatures=None, alignment=None):
        pre = 'train' if split == 'train' else 'valid'                  # :24-29
        self.src = nmt_Data[pre + '_src_label']
        self.src_len = np.asarray(nmt_Data[pre + '_src_label_length'])
        self.tgt = nmt_Data[pre + '_tgt_label']
        self.tgt_len = np.asarray(nmt_Data[pre + '_tgt_label_length'])
        assert len(self.src) == len(self.tgt)
        if alignment is not None or srcFeatures or tgtFeatures:
            raise NotImplementedError("copy alignments / word features are not part of the reference's h5 corpus path (:26,30)")
        self._type = data_type
        self.cuda = cuda
        self.alignment = None
        self.batchSize = batchSize
        self.numBatches = math.ceil(len(self.src) / batchSize)
        self.volatile = volatile

    def _batchify(self, data, lengths):
        """[n, max(lengths)] int64, row i = data[i, :lengths[i]] then PAD (:37-43)."""
        data = np.asarray(data)
        lengths = lengths.astype(np.int64)
        width = int(lengths.max())
        keep = np.arange(width)[None, :] < lengths[:, None]
        return np.where(keep, data[:, :width].astype(np.int32).astype(np.int64), PAD)

    def __getitem__(self, index):
        assert index < self.numBatches, "%d > %d" % (index, self.numBatches)
        s, e = index * self.batchSize, (index + 1) * self.batchSize
        src_lengths = self.src_len[s:e]
        src = self._batchify(self.src[s:e], src_lengths)
        tgt = self._batchify(self.tgt[s:e], self.tgt_len[s:e])
        n = src.shape[0]
        # within-batch sort by decreasing source length -- the reference's own call, so ties fall the same way (:75)
        lengths, perm = torch.sort(torch.from_numpy(src_lengths.astype('int32')), 0, descending=True)
        order = perm.numpy()
        src_t = np.ascontiguousarray(src[order].T)[:, :, None]          # [S, n, 1]
        tgt_t = np.ascontiguousarray(tgt[order].T)                      # [T, n]
        tgt_dev = self._ship(tgt_t)
        if self.cuda:
            # how many target positions are words is known here where the batch is assembled: the NMT step's generator and
            # criterion skip the padding (uic_nmt_dims.tgt_live_count; NMTModel.forward reads the attribute)
            tgt_dev.uic_live = (None, int(np.count_nonzero(tgt_t[1:])))
        return Batch(self._ship(src_t), tgt_dev, lengths.view(1, -1), [int(p) for p in perm], n)

    def _ship(self, a):
        t = torch.from_numpy(a)
        if not self.cuda:
            return t
        return t.pin_memory().cuda(non_blocking=True)

    def __len__(self):
        return self.numBatches
