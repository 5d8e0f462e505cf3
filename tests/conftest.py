import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Return (cfg dict, weights, inputs, outputs, grads, extra) of one golden fixture as torch tensors."""
    import numpy as np
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    keys = ["V", "E", "H", "A", "D", "L", "n_img", "S", "R", "use_bn", "bn_train"]
    cfg = dict(zip(keys, [int(x) for x in z["cfg"]]))
    W, I, O, G, X = {}, {}, {}, {}, {}
    for k in z.files:
        if k == "cfg":
            continue
        if "::" not in k:
            X[k] = z[k]
            continue
        kind, key = k.split("::", 1)
        t = torch.from_numpy(z[k])
        {"w": W, "in": I, "out": O, "grad": G}.get(kind, X)[key if kind in ("w", "in", "out", "grad") else k] = t
    cfg["logit_layers"] = int(X.pop("logit_layers", 1))
    cfg["Dfc"] = int(X.pop("fc_feat_size", cfg["D"]))
    return cfg, W, I, O, G, X


@pytest.fixture(scope="session")
def golden_loader():
    return load_golden


def poison_workspaces(eng):
    """Every workspace the engine hands out from now on is first filled with 0xFF bytes (NaN in f32 and bf16, -1 in the integer
    buffers), on the stream that is about to use it.  A pooled or re-allocated buffer otherwise still holds the values of the
    previous, often identical, run -- which would hide a kernel that reads something it (or a neighbour) should have written
    first, or data that never became visible to the reader (the SAFE exchange protocol's write-through stores)."""
    orig = eng.checkout

    def checkout(d, device):
        ws = orig(d, device)
        ws.buf.fill_(255)
        return ws

    eng.checkout = checkout
    return eng


@pytest.fixture(autouse=True)
def _poisoned_engine_workspaces(request, monkeypatch):
    """GPU tests: every workspace any TopDownEngine hands out is poisoned first (see poison_workspaces) -- results must never
    depend on what a pooled or recycled buffer still held."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from unpaired_image_captioning_amd.topdown_engine import TopDownEngine
    orig = TopDownEngine.checkout

    def checkout(self, d, device):
        ws = orig(self, d, device)
        ws.buf.fill_(255)
        return ws

    monkeypatch.setattr(TopDownEngine, "checkout", checkout)
    yield
