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


@pytest.fixture(params=[0, 1], ids=["f32", "bf16x3"])
def conv_math(request):
    """Runs a GPU test under both convolution arithmetics: exact fp32 (library default) and the split-bf16 mode that
    bench.py times (twin kernels, twin-only tensors and DIGA_TWIN_CONV3 at their defaults = on)."""
    from diga_amd import _lib
    prev = _lib.get_conv_math()
    _lib.set_conv_math(request.param)
    yield request.param
    _lib.set_conv_math(prev)


def winograd_tile(n, cin, h, w, cout, k, stride, pad, dil):
    """Output-tile edge of the Winograd path an exact-fp32 DigaConv2d of this geometry takes (0: a direct kernel); WINO_TOL gives the
    bound its results are held to against float64, in units of the tensor's scale."""
    from diga_amd.model import conv as dc
    cp = dc._pad_to(cin)
    if k != 3 or not dc._winograd_ok(n, h, w, cp, cout, 3, 3, (stride, stride), (-pad, -pad), (dil, dil), h, w):
        return 0
    return dc._wino_plan(h, w, dil)[0]


# (outputs and input gradients, weight gradients) per Winograd tile edge; 0 = direct kernels.  F(2x2,3x3) is as accurate as the direct
# fmaf chain; the larger tiles' transforms multiply by up to 8 / 32 and 1/15 / 1/90: bounds = about 3x the worst value measured on the
# test shapes (tests print them)
# measured (test_winograd_f32_vs_float64, worst of nine shapes): 4x4 tiles y 5.8e-6 / dx 6.0e-6 / dw 8.0e-6; 6x6 tiles 2.7e-5 / 2.8e-5 / 1.6e-5
WINO_TOL = {0: (1e-5, 1e-5), 2: (1e-5, 1e-5), 4: (2e-5, 3e-5), 6: (9e-5, 9e-5)}


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


def assert_mostly_close(a, b, rtol, atol, frac, l2, what=""):
    """For gradients of networks with ReLUs compared against a FIXED capture: a 1e-5-level arithmetic difference (the
    split-bf16 conv mode) flips the few ReLUs whose pre-activation is that close to zero, which moves the gradients in
    their receptive field by O(1) -- a property of the function, present between any two fp32 implementations at a
    smaller rate.  At most `frac` of the elements may exceed the elementwise bound and the whole tensor must agree to
    `l2` in relative L2 norm."""
    a = torch.as_tensor(a).detach().cpu().to(torch.float64)
    b = torch.as_tensor(b).detach().cpu().to(torch.float64)
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    bad = float(((a - b).abs() > atol + rtol * b.abs()).double().mean())
    err = float((a - b).norm() / b.norm().clamp_min(1e-300))
    frac = max(frac, 8.0 / a.numel())            # small tensors (bias gradients): up to 8 elements
    assert bad <= frac and err <= l2, f"{what}: {bad:.2e} of the elements out of tol (allowed {frac}), L2 error {err:.2e} (allowed {l2})"
