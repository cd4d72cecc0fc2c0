"""Colour-augmentation view (diga_color_aug_view) against oracle/coloraug.py -- the restatement of kornia 0.5.8's
ColorJitter / RandomGrayscale / RandomGaussianBlur / RandomSharpness and of the reference's Normalize + beta blend
(G5/train_DiGA_gta2city_warm_up.py:105-111,233; util/utils.py:141-156).  kornia itself is absent: parity unpinned
against it, pinned against the restatement.  Float chain (HSV round trips, 3x3 filters): 5e-6 absolute on O(1) values."""
import itertools

import numpy as np
import pytest
import torch

from oracle import coloraug as oc
from oracle import synth

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _table(P):
    b = len(P["jitter"])
    tab = np.zeros((b, 12), dtype=np.float32)
    tab[:, 0], tab[:, 1], tab[:, 2], tab[:, 3] = P["jitter"], P["gray"], P["blur"], P["sharp"]
    tab[:, 4:8] = P["factors"]
    tab[:, 8] = P["sharp_factor"]
    return tab, np.array(P["order"], dtype=np.int32)


def test_parameter_generator_matches_oracle_restatement():
    from diga_amd.util import augment as A
    for seed in (0, 1, 12345, 2 ** 32 - 1):
        tab, order = A.draw_params(seed, 7)
        t2, o2 = _table(oc.params_from_seed(seed, 7))
        assert np.array_equal(tab, t2) and np.array_equal(order, o2)
    tab, _ = A.draw_params(99, 4000)
    # frequencies of the four decisions and ranges of the factors
    assert abs(tab[:, 0].mean() - 0.5) < 0.03 and abs(tab[:, 1].mean() - 0.3) < 0.03
    assert abs(tab[:, 2].mean() - 0.8) < 0.03 and abs(tab[:, 3].mean() - 0.3) < 0.03
    assert tab[:, 4].min() >= 0.6 and tab[:, 4].max() <= 1.4 and tab[:, 6].min() >= 0.8 and tab[:, 6].max() <= 1.2
    assert np.abs(tab[:, 7]).max() <= 0.1 and tab[:, 8].min() >= 0 and tab[:, 8].max() <= 0.5
    orders = {tuple(A.draw_params(s, 1)[1]) for s in range(400)}
    assert len(orders) == 24                                   # every permutation occurs


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(37, 70), (16, 64), (3, 3), (65, 129)])
def test_color_aug_view_vs_oracle(H, W):
    from diga_amd.util import augment as A
    g = synth.gen(H * 1000 + W)
    perms = list(itertools.permutations(range(4)))
    for trial in range(6):
        B = 8
        # [0,1]-ish images, normalised images (negative values: kornia's clamps cut them), and constant-colour patches
        x = torch.rand((B, 3, H, W), generator=g)
        if trial % 2:
            x = (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)
        x[0, :, : H // 2] = x[0, :, :1, :1]                   # grey / saturated flats: HSV corner cases (delta = 0, ties)
        x[1, 0] = x[1, 1]
        P = {"jitter": [b % 2 == 0 for b in range(B)], "gray": [b in (1, 2, 6) for b in range(B)],
             "blur": [b not in (0, 3) for b in range(B)], "sharp": [b in (0, 2, 4, 5, 7) for b in range(B)],
             "factors": np.stack([0.6 + 0.8 * torch.rand(B, generator=g).numpy(), 0.6 + 0.8 * torch.rand(B, generator=g).numpy(),
                                  0.8 + 0.4 * torch.rand(B, generator=g).numpy(), -0.1 + 0.2 * torch.rand(B, generator=g).numpy()],
                                 1).astype(np.float32),
             "sharp_factor": np.array([0.0, 0.3, 0.5, 0.2, 1.0, 0.45, 0.1, 1.5], dtype=np.float32),
             "order": list(perms[(trial * 5 + H) % 24])}
        want = oc.color_aug_view(x, P, 0.4, MEAN, STD)
        tab, order = _table(P)
        got = A.color_aug_view(x.cuda(), 0.4, MEAN, STD, tab, order).cpu()
        err = (got - want).abs()
        # hue is discontinuous where two channels tie for the maximum: compare away from exact ties, count the rest
        assert float(err.max()) < 5e-5 or float((err > 5e-5).float().mean()) < 1e-4, f"trial {trial}: max err {float(err.max()):.2e}"
        assert float(err.median()) < 1e-6


@pytest.mark.gpu
def test_color_aug_fullsize_identities():
    """C2-size batch: with every augmentation switched off the view is beta*Normalize(x) + (1-beta)*x; grayscale makes
    the three channels of extra_aug equal; the module draws new parameters on every call and is reproducible."""
    from diga_amd.util import augment as A
    g = synth.gen(5)
    x = torch.randn((8, 3, 768, 768), generator=g).cuda()
    tab = np.zeros((8, 12), dtype=np.float32)
    tab[:, 4:7] = 1.0
    out = A.color_aug_view(x, 0.4, MEAN, STD, tab, [0, 1, 2, 3])
    m, s = torch.tensor(MEAN, device="cuda").view(1, 3, 1, 1), torch.tensor(STD, device="cuda").view(1, 3, 1, 1)
    ref = 0.4 * ((x - m) / s) + 0.6 * x
    assert float((out - ref).abs().max()) < 1e-5
    tab[:, 1] = 1.0
    y = A.color_aug_view(x, 1.0, (0, 0, 0), (1, 1, 1), tab, [0, 1, 2, 3])
    assert torch.equal(y[:, 0], y[:, 1]) and torch.equal(y[:, 1], y[:, 2])
    a1, a2 = A.ExtraAug(seed=3), A.ExtraAug(seed=3)
    y1, y2, y3 = a1(x[:2]), a2(x[:2]), a1(x[:2])
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    v = a2.view(x[:2], 0.4, MEAN, STD)
    assert v.shape == x[:2].shape and bool(torch.isfinite(v).all())
