import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Returns (arrays, weights) — weights are the 'w::'-prefixed entries as torch tensors."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrs, w = {}, {}
    for k in z.files:
        if k.startswith("w::"):
            w[k[3:]] = torch.from_numpy(z[k])
        else:
            arrs[k] = z[k]
    return arrs, w


def scaled_decoder(w, scale):
    """The state dict of oracle/gen_fixtures.py's second generate() pair: every 2-D `model.layers.*` matrix x scale."""
    return {k: (v * float(scale) if (k.startswith("model.layers.") and v.ndim == 2) else v) for k, v in w.items()}


def t(a):
    return torch.from_numpy(np.asarray(a))


def rel_err(a, b):
    a = a.float(); b = b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.fixture(scope="session")
def golden():
    return load_golden
