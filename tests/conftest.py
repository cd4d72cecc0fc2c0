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


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden(dict):
    def t(self, key):
        return torch.from_numpy(np.asarray(self[key]))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
                cache[name] = Golden({k: z[k] for k in z.files})
        return cache[name]

    return load


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = torch.as_tensor(a).detach().cpu().to(torch.float64)
    b = torch.as_tensor(b).detach().cpu().to(torch.float64)
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax((err - tol).reshape(-1)))
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} out of tol; worst idx {i}: "
            f"got {a.reshape(-1)[i].item():.9g} want {b.reshape(-1)[i].item():.9g} "
            f"(err {err.reshape(-1)[i].item():.3g}, rtol {rtol}, atol {atol})")
