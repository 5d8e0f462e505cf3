"""CPU: the discriminator oracle (parity unpinned: there is no reference discriminator) is self-consistent -- its convolution
is the sliding dot product the HIP path's multi-segment GEMM computes, written out with explicit loops."""
import torch

from oracle import discriminator as O


def test_oracle_convolution_equals_explicit_loops():
    V1, E, F, widths, N, L = 20, 8, 4, (1, 2, 3), 3, 6
    W = O.init_weights(V1, E, F, widths, seed=1)
    g = torch.Generator().manual_seed(2)
    tok = torch.randint(0, V1, (N, L), generator=g)
    x = torch.relu(W["embed.weight"][tok])
    pooled = []
    for w in widths:
        y = torch.zeros(N, L, F)
        for t in range(L):
            for j in range(w):
                if t + j < L:
                    y[:, t] += x[:, t + j] @ W["conv%d.weight" % w][:, j, :].t()
            y[:, t] += W["conv%d.bias" % w]
        pooled.append(torch.relu(y).max(dim=1)[0])
    p = torch.cat(pooled, 1)
    Ft = p.shape[1]
    gh = p @ W["highway.weight"].t() + W["highway.bias"]
    z = torch.sigmoid(gh[:, :Ft]) * torch.relu(gh[:, Ft:]) + (1 - torch.sigmoid(gh[:, :Ft])) * p
    ref = z @ W["out.weight"] + W["out.bias"]
    assert torch.allclose(O.forward(W, tok, widths), ref, atol=1e-5)
    loss, grads, _ = O.loss_and_grads(W, tok, torch.tensor([1.0, 0.0, 1.0]), widths)
    assert torch.isfinite(loss) and set(grads) == set(W)
