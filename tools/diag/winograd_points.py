#!/usr/bin/env python3
"""fp32 error of Winograd F(m x m, 3x3) for candidate interpolation points (CPU, numpy): a 512-channel layer on post-ReLU inputs, every
transform and the channel contraction rounded to float32, against a float64 direct convolution.  This is the measurement behind the
point sets in tools/gen_winograd_xforms.py.

    python tools/diag/winograd_points.py"""
import os
import sys
from fractions import Fraction as Fr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gen_winograd_xforms import matrices  # noqa: E402


def mats(m, pts):
    AT, G, BT = matrices(m, pts)
    f = lambda M: np.array(M.tolist(), dtype=np.float64)  # noqa: E731
    return f(AT), f(G), f(BT)


def conv_direct(x, w):
    C, H, W = x.shape
    y = np.zeros((w.shape[0], H - 2, W - 2))
    for r in range(3):
        for s in range(3):
            y += np.einsum("kc,chw->khw", w[:, :, r, s].astype(np.float64), x[:, r:r + H - 2, s:s + W - 2].astype(np.float64))
    return y


def wino(x, w, AT, G, BT, m):
    f = np.float32
    AT, G, BT = AT.astype(f), G.astype(f), BT.astype(f)
    a = m + 2
    U = np.einsum("ir,kcrs,js->kcij", G, w, G).astype(f)
    ty, tx = (x.shape[1] - 2) // m, (x.shape[2] - 2) // m
    y = np.zeros((w.shape[0], ty * m, tx * m), dtype=f)
    for i in range(ty):
        for j in range(tx):
            d = x[:, i * m:i * m + a, j * m:j * m + a]
            V = np.einsum("ciq,jq->cij", np.einsum("ip,cpq->ciq", BT, d).astype(f), BT).astype(f)
            M = np.einsum("kcij,cij->kij", U, V, dtype=f)
            y[:, i * m:i * m + m, j * m:j * m + m] = np.einsum("kpj,qj->kpq", np.einsum("pi,kij->kpj", AT, M).astype(f), AT).astype(f)
    return y


def main():
    rng = np.random.default_rng(0)
    C, K = 512, 32
    x = np.maximum(rng.standard_normal((C, 14, 14)), 0).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) * (2 / (C * 9)) ** .5).astype(np.float32)
    ref = conv_direct(x, w)
    sc = np.abs(ref).max()
    H = Fr(1, 2)
    sets = {"F2  0,1,-1": (2, [0, 1, -1]), "F4  0,+-1,+-2 (Lavin & Gray)": (4, [0, 1, -1, 2, -2]), "F4  0,1,-1,2,-1/2 (taken)": (4, [0, 1, -1, 2, -H]),
            "F4  0,1,-1,1/2,-2": (4, [0, 1, -1, H, -2]), "F4  0,+-1,+-1/2": (4, [0, 1, -1, H, -H]),
            "F6  0,+-1,+-2,+-1/2 (taken)": (6, [0, 1, -1, 2, -2, H, -H]), "F6  0,+-1,+-1/2,+-3/2": (6, [0, 1, -1, H, -H, Fr(3, 2), Fr(-3, 2)]),
            "F6  0,+-1,+-1/2,+-3/4": (6, [0, 1, -1, H, -H, Fr(3, 4), Fr(-3, 4)]), "F6  0,+-1,+-1/2,2,-3": (6, [0, 1, -1, H, -H, 2, -3])}
    for name, (m, pts) in sets.items():
        AT, G, BT = mats(m, pts)
        y = wino(x, w, AT, G, BT, m)
        r = ref[:, :y.shape[1], :y.shape[2]]
        print(f"{name:34s} max {np.abs(y - r).max() / sc:.2e}  rms {np.sqrt(((y - r) ** 2).mean()) / sc:.2e}   |B^T| {np.abs(BT).max():.3g} |G| "
              f"{np.abs(G).max():.3g} |A^T| {np.abs(AT).max():.3g}")


if __name__ == "__main__":
    main()
