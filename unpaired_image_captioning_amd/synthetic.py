"""Synthetic batches with the layout of DataLoader.get_batch (P/misc/dataloader/dataloader.py:209-299),
generated directly on the device (BASELINE.md section 3 / SURVEY.md section 8d): non-negative
L2-normalised region features, fc = region mean, each image replicated seq_per_img times,
tokens ~ U{1..V}, caption lengths ~ U{L/2..L}, masks of ones on the first len + 2 positions."""
import torch


def synthetic_batch(n_img, seq_per_img, R, D, V, L, seed=1234, device="cuda", ragged_regions=False):
    g = torch.Generator(device=device).manual_seed(seed)
    att = torch.randn(n_img, R, D, generator=g, device=device).abs_()
    att = att / att.norm(dim=2, keepdim=True)
    if ragged_regions:
        cnt = torch.randint(max(1, R // 4), R + 1, (n_img,), generator=g, device=device)
        cnt[0] = R
        cnt, _ = torch.sort(cnt, descending=True)            # the loader sorts by region count (:267-268)
    else:
        cnt = torch.full((n_img,), R, dtype=torch.long, device=device)
    att_masks = (torch.arange(R, device=device)[None, :] < cnt[:, None]).float()
    att = att * att_masks.unsqueeze(2)
    fc = att.sum(1) / cnt[:, None].float()
    N = n_img * seq_per_img
    rep = torch.arange(n_img, device=device).repeat_interleave(seq_per_img)
    lens = torch.randint(max(1, L // 2), L + 1, (N,), generator=g, device=device)
    toks = torch.randint(1, V + 1, (N, L), generator=g, device=device)
    labels = torch.zeros(N, L + 2, dtype=torch.long, device=device)
    pos = torch.arange(L, device=device)[None, :]
    labels[:, 1:L + 1] = toks * (pos < lens[:, None]).long()
    masks = (torch.arange(L + 2, device=device)[None, :] < (lens[:, None] + 2)).float()
    return dict(fc_feats=fc[rep].contiguous(), att_feats=att[rep].contiguous(),
                att_masks=att_masks[rep].contiguous(), labels=labels, masks=masks)
