"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/).

CPU restatement of the scene-graph GCN encoder of csrc/gcn.hip -- PARITY UNPINNED: the reference tree holds no GCN code
(BASELINE configs[4] names one), so this file restates the SAME spec the HIP path implements,
    X_{l+1} = relu(A_hat (X_l W_l^T) + b_l),
with plain torch CPU ops and autograd; it checks the kernels against the spec, not against the reference.
"""
import torch


def init_weights(D, H, layers, seed=0):
    g = torch.Generator().manual_seed(seed)
    W = {}
    for l in range(layers):
        fan = D if l == 0 else H
        W["gcn.%d.weight" % l] = (torch.rand(H, fan, generator=g) * 2 - 1) / fan ** 0.5
        W["gcn.%d.bias" % l] = (torch.rand(H, generator=g) * 2 - 1) / fan ** 0.5
    return W


def normalised_adjacency(n_img, R, seed=0, density=0.25):
    """A random relation graph per image: symmetric, self loops, rows scaled to sum 1 (A_hat = D^-1 (A + I))."""
    g = torch.Generator().manual_seed(seed)
    a = (torch.rand(n_img, R, R, generator=g) < density).float()
    a = ((a + a.transpose(1, 2)) > 0).float() + torch.eye(R)[None]
    a = (a > 0).float()
    return a / a.sum(2, keepdim=True)


def _q(t, on):
    return t.to(torch.bfloat16).float() if on else t


def forward(W, x, adj, layers, bf16=False):
    """x [N, R, D], adj [N, R, R] -> [N, R, H].  bf16=True rounds what the bf16 path stores in bf16 (operands of the GEMMs,
    the layer outputs)."""
    h = _q(x, bf16)
    for l in range(layers):
        y = _q(h @ _q(W["gcn.%d.weight" % l], bf16).t(), bf16)
        h = _q(torch.relu(adj @ y + W["gcn.%d.bias" % l]), bf16)
    return h


def forward_backward(W, x, adj, layers, dout, bf16=False):
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    xg = x.clone().requires_grad_(True)
    out = forward(Wg, xg, adj, layers, bf16)
    out.backward(dout)
    return out.detach(), {k: v.grad for k, v in Wg.items()}, xg.grad
