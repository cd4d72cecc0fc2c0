#!/usr/bin/env python3
"""Generates diga_amd/csrc/winograd_xforms.h: the 1-D transforms of Winograd F(m,3) for m = 4 and m = 6 as straight-line
device code (exact rational Toom-Cook construction with sympy, coefficients printed as fp32 literals).

    python tools/gen_winograd_xforms.py [out.h]    # rewrites the header (committed; the build does not need sympy)

For interpolation points p_0 .. p_{n-1} (n = m + 1) and the point at infinity, with f_j = prod_{k != j} (p_j - p_k):
    A^T [m x (m+2)]      A^T[i][j] = p_j^i,  last column e_{m-1}
    G   [(m+2) x 3]      G[j][k]   = p_j^k / f_j,  last row e_2
    B^T [(m+2) x (m+2)]  row j = coefficients of prod_{k != j} (x - p_k),  last row = coefficients of prod_k (x - p_k)
so that  y = A^T [ (G g) (.) (B^T d) ]  is the m-output correlation of d (m + 2 values) with g (3 taps).

Points: chosen by measured fp32 error of a 512-channel layer against float64 (tools/diag/winograd_points.py):
    F(4,3): 0, 1, -1, 2, -1/2    max 4.9e-6 of the output scale (Lavin & Gray's 0, +-1, +-2: 1.2e-5)
    F(6,3): 0, +-1, +-2, +-1/2   max 2.7e-5 (the usual set; nothing tried was better)
Rows that belong to a pair of points +-p share their even / odd parts (r+ = E + O, r- = E - O)."""
import os
import sys
from fractions import Fraction as Fr

import numpy as np
import sympy

POINTS = {4: [0, 1, -1, 2, Fr(-1, 2)], 6: [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)]}


def matrices(m, pts, r=3):
    a = m + r - 1
    n = a - 1
    assert len(pts) == n
    x = sympy.symbols("x")
    p = [sympy.Rational(Fr(q).numerator, Fr(q).denominator) for q in pts]
    f = [sympy.prod([p[i] - p[j] for j in range(n) if j != i]) for i in range(n)]
    AT = sympy.zeros(m, a)
    for i in range(m):
        for j in range(n):
            AT[i, j] = p[j] ** i
    AT[m - 1, n] = 1
    G = sympy.zeros(a, r)
    for j in range(n):
        for k in range(r):
            G[j, k] = p[j] ** k / f[j]
    G[n, r - 1] = 1
    BT = sympy.zeros(a, a)
    M = sympy.prod([x - q for q in p])
    for j in range(n):
        for k, c in enumerate(sympy.Poly(sympy.cancel(M / (x - p[j])), x).all_coeffs()[::-1]):
            BT[j, k] = c
    for k, c in enumerate(sympy.Poly(M, x).all_coeffs()[::-1]):
        BT[n, k] = c
    return AT, G, BT


def lit(c):
    v = np.float32(float(c))
    s = repr(float(v))
    if "e" not in s and "." not in s:
        s += ".0"
    return f"{np.format_float_scientific(v, unique=True, trim='0')}f" if "e" in s else f"{s}f"


def term_chain(coeffs, names, out):
    """Statements computing out = sum coeffs[k] * names[k] (zeros skipped, +-1 as add / sub, the rest as fma)."""
    items = [(c, nm) for c, nm in zip(coeffs, names) if c != 0]
    if not items:
        return [f"{out} = vzero<V>();"]
    # start from a +1 term when there is one (saves a multiply)
    items.sort(key=lambda t: 0 if t[0] == 1 else 1)
    c0, n0 = items[0]
    lines = [f"{out} = {n0};" if c0 == 1 else f"{out} = f4scale({lit(c0)}, {n0});"]
    for c, nm in items[1:]:
        if c == 1:
            lines.append(f"{out} = f4add({out}, {nm});")
        elif c == -1:
            lines.append(f"{out} = f4sub({out}, {nm});")
        else:
            lines.append(f"{out} = f4fma({lit(c)}, {nm}, {out});")
    return lines


def emit_matvec(name, Mx, doc):
    """template <typename V> void name(const V* x, V* r): r = Mx x, pairing rows with r_j[k] = +-(-1)^k r_i[k]."""
    rows, cols = Mx.shape
    R = [[Mx[i, k] for k in range(cols)] for i in range(rows)]
    used = set()
    body = []
    tmp = 0
    for i in range(rows):
        if i in used:
            continue
        partner, sign = None, 0
        for j in range(i + 1, rows):
            if j in used:
                continue
            if all(R[j][k] == R[i][k] * (-1) ** k for k in range(cols)) and any(R[i][k] != 0 for k in range(1, cols, 2)):
                partner, sign = j, 1
                break
            if all(R[j][k] == -R[i][k] * (-1) ** k for k in range(cols)) and any(R[i][k] != 0 for k in range(0, cols, 2)):
                partner, sign = j, -1
                break
        if partner is None:
            body += term_chain(R[i], [f"x[{k}]" for k in range(cols)], f"r[{i}]")
            used.add(i)
            continue
        ev = [R[i][k] if k % 2 == 0 else 0 for k in range(cols)]
        od = [R[i][k] if k % 2 == 1 else 0 for k in range(cols)]
        e, o = f"e{tmp}", f"o{tmp}"
        tmp += 1
        body.append(f"V {e}, {o};")
        body += term_chain(ev, [f"x[{k}]" for k in range(cols)], e)
        body += term_chain(od, [f"x[{k}]" for k in range(cols)], o)
        body.append(f"r[{i}] = f4add({e}, {o});")
        body.append(f"r[{partner}] = f4sub({e}, {o});" if sign == 1 else f"r[{partner}] = f4sub({o}, {e});")
        used |= {i, partner}
    src = [f"// {doc}", "template <typename V>", f"__device__ __forceinline__ void {name}(const V* x, V* r) {{"]
    src += ["    " + ln for ln in body]
    src.append("}")
    return "\n".join(src)


def fmt_matrix(Mx):
    return "\n".join("//     [" + "  ".join(f"{str(Mx[i, k]):>7s}" for k in range(Mx.shape[1])) + " ]" for i in range(Mx.shape[0]))


def main():
    out = ["// GENERATED by tools/gen_winograd_xforms.py -- do not edit.  1-D transforms of Winograd F(m,3), m = 4 and 6, as straight-line",
           "// code on vector types V (float, float2, float4: f4add / f4sub / f4fma / f4scale / vzero<V> of winograd.hip).  r and x never alias.",
           "#pragma once", ""]
    for m, pts in POINTS.items():
        AT, G, BT = matrices(m, pts)
        # self-check in exact arithmetic
        d = sympy.Matrix([sympy.Rational(k * k + 3 * k + 1, 7) for k in range(m + 2)])
        g = sympy.Matrix([sympy.Rational(2, 3), sympy.Rational(-5, 4), sympy.Rational(1, 9)])
        y = AT * sympy.Matrix([(G * g)[j] * (BT * d)[j] for j in range(m + 2)])
        ref = sympy.Matrix([sum(d[i + k] * g[k] for k in range(3)) for i in range(m)])
        assert y == ref, (m, y, ref)
        out.append(f"// ---------------------------------------------------------------- F({m},3), points {', '.join(str(q) for q in pts)}, inf")
        out.append("//   B^T =\n" + fmt_matrix(BT))
        out.append("//   G =\n" + fmt_matrix(G))
        out.append("//   A^T =\n" + fmt_matrix(AT))
        out.append(emit_matvec(f"wino{m}_bt", BT, f"r[{m + 2}] = B^T x[{m + 2}]   (input transform, one column / row)"))
        out.append(emit_matvec(f"wino{m}_g", G, f"r[{m + 2}] = G x[3]   (weight transform)"))
        out.append(emit_matvec(f"wino{m}_at", AT, f"r[{m}] = A^T x[{m + 2}]   (output transform)"))
        out.append(emit_matvec(f"wino{m}_a", AT.T, f"r[{m + 2}] = A x[{m}]   (output-gradient transform of the weight gradient)"))
        out.append(emit_matvec(f"wino{m}_gt", G.T, f"r[3] = G^T x[{m + 2}]   (back to the 3 taps of the weight gradient)"))
        out.append("")
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diga_amd",
                                                              "csrc", "winograd_xforms.h")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
