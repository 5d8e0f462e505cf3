"""The helpers of P/misc/utils.py the hot path's callers use: `if_use_att`, `decode_sequence`."""
import numpy as np
import torch


def if_use_att(caption_model):
    """P/misc/utils.py:42-46."""
    return caption_model not in ['show_tell', 'all_img', 'fc']


def decode_sequence(ix_to_word, seq):
    """P/misc/utils.py:49-66: token ids [N, D] (0 = END) -> list of N strings.  One device-to-host copy for the whole
    batch instead of one `.item()` synchronisation per token."""
    ids = seq.detach().cpu().numpy() if torch.is_tensor(seq) else np.asarray(seq)
    out = []
    for row in ids:
        words = []
        for ix in row:
            if ix <= 0:
                break
            words.append(ix_to_word[str(int(ix))])
        out.append(' '.join(words))
    return out
