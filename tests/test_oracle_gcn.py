"""CPU: the GCN oracle (parity unpinned: there is no reference GCN) equals the layer written out with explicit loops."""
import torch

from oracle import gcn as O


def test_oracle_layer_equals_explicit_loops():
    N, R, D, H = 2, 4, 6, 5
    W = O.init_weights(D, H, 1, seed=3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, R, D, generator=g)
    adj = O.normalised_adjacency(N, R, seed=4)
    assert torch.allclose(adj.sum(2), torch.ones(N, R))
    ref = torch.zeros(N, R, H)
    for n in range(N):
        for i in range(R):
            acc = torch.zeros(H)
            for j in range(R):
                acc += adj[n, i, j] * (W["gcn.0.weight"] @ x[n, j])
            ref[n, i] = torch.relu(acc + W["gcn.0.bias"])
    assert torch.allclose(O.forward(W, x, adj, 1), ref, atol=1e-5)
    out, grads, dx = O.forward_backward(W, x, adj, 1, torch.ones(N, R, H))
    assert set(grads) == set(W) and dx.shape == x.shape
