"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/).

CPU restatement of the CNN sentence discriminator of csrc/discriminator.hip -- PARITY UNPINNED: the reference tree holds no
discriminator code (SURVEY finding 2; BASELINE configs[3] names one), so this file restates the SAME spec the HIP path
implements (the usual text-CNN critic: embedding -> ReLU -> 1-D convolutions of several widths -> ReLU -> max over time ->
highway -> dropout -> linear -> sigmoid) with plain torch CPU ops and autograd; it checks the kernels against the spec, not
against the reference.
"""
import torch
import torch.nn.functional as F_


def init_weights(V1, E, F, widths, seed=0, scale=0.1):
    g = torch.Generator().manual_seed(seed)
    Ft = F * len(widths)
    W = {"embed.weight": torch.randn(V1, E, generator=g) * 0.5}
    for w in widths:
        W["conv%d.weight" % w] = torch.randn(F, w, E, generator=g) * scale
        W["conv%d.bias" % w] = torch.randn(F, generator=g) * scale
    W["highway.weight"] = torch.randn(2 * Ft, Ft, generator=g) * scale
    W["highway.bias"] = torch.randn(2 * Ft, generator=g) * scale
    W["out.weight"] = torch.randn(Ft, generator=g) * scale
    W["out.bias"] = torch.randn(1, generator=g) * scale
    return W


def _q(t, on):
    """Round to bf16 and back (autograd passes the gradient through the two casts): where the HIP bf16 path stores an operand."""
    return t.to(torch.bfloat16).float() if on else t


def forward(W, tokens, widths, drop_mask=None, bf16=False):
    """tokens [N, L] int64 -> logits [N].  drop_mask: the multiplicative dropout mask [N, Ft] (0 or 1/(1-p)) or None.
    bf16=True rounds what the bf16 path holds in bf16 (embedded rows, conv / highway weights, conv outputs), so that the
    max-over-time winners -- which a rounding-sized difference can flip, re-routing a whole gradient -- are the same."""
    x = _q(torch.relu(W["embed.weight"][tokens]), bf16)           # [N, L, E]
    N, L, E = x.shape
    pooled = []
    for w in widths:
        xp = F_.pad(x, (0, 0, 0, w - 1))                          # right zero padding in time
        y = F_.conv1d(xp.transpose(1, 2), _q(W["conv%d.weight" % w], bf16).permute(0, 2, 1), W["conv%d.bias" % w])   # [N, F, L]
        pooled.append(_q(torch.relu(y), bf16).max(dim=2)[0])
    p = torch.cat(pooled, 1)                                      # [N, Ft]
    Ft = p.shape[1]
    gh = p @ _q(W["highway.weight"], bf16).t() + W["highway.bias"]
    g, h = torch.sigmoid(gh[:, :Ft]), torch.relu(gh[:, Ft:])
    z = g * h + (1 - g) * p
    if drop_mask is not None:
        z = z * drop_mask
    return z @ W["out.weight"] + W["out.bias"]


def loss_and_grads(W, tokens, labels, widths, drop_mask=None, bf16=False):
    """BCEWithLogitsLoss (mean) and the gradient of every tensor of W."""
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    logits = forward(Wg, tokens, widths, drop_mask, bf16)
    loss = F_.binary_cross_entropy_with_logits(logits, labels)
    loss.backward()
    return loss.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Wg.items()}, logits.detach()
