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

