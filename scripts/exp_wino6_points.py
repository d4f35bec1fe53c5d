#!/usr/bin/env python3
"""Derivation (sympy, exact) and fp32 accuracy of Cook-Toom transforms for several point sets.  First part: F(4x4,3x3), whose matrices in
csrc/wino6.hip are the output for the points (0, 1, -1, 2, -1/2, inf).  Second part (general_test): F(m x m, r x r) for any m, r --
F(4x4,4x4) over (0, 1, -1, 2, -2, 1/2, inf) = w7_*, F(6x6,3x3) over (0, 1, -1, 2, -2, 1/2, -1/2, inf) = w8_*, and the alternatives that
were measured against them.  CPU only."""
import numpy as np, sympy as sp, itertools
from fractions import Fraction
def mats(points, m=4, r=3):
    n = m + r - 1
    pts = [sp.Rational(p) for p in points]   # n-1 finite points; last is infinity
    def ev(deg):  # evaluation matrix (n x (deg+1)): rows finite points, last row inf (leading coeff)
        M = sp.zeros(n, deg + 1)
        for i, p in enumerate(pts):
            for j in range(deg + 1): M[i, j] = p ** j
        M[n - 1, deg] = 1
        return M
    Ea, Eg, V = ev(m - 1), ev(r - 1), ev(n - 1)
    AT = Ea.T; G = Eg; BT = (V.inv()).T
    return AT, G, BT
def scale_rows(G, BT):
    # move row scales so that BT rows have integer/dyadic entries where possible: scale BT row i by lcm of denominators
    n = BT.shape[0]
    G2, B2 = G.copy(), BT.copy()
    for i in range(n):
        den = sp.ilcm(*[sp.fraction(x)[1] for x in BT.row(i)])
        num = sp.igcd(*[sp.fraction(x * den)[0] for x in BT.row(i) if x != 0])
        s = sp.Rational(den, num)
        B2[i, :] = BT.row(i) * s
        G2[i, :] = G.row(i) / s
    return G2, B2
def test(points, seed=0, C=256, T=64):
    AT, G, BT = mats(points)
    G, BT = scale_rows(G, BT)
    ATn, Gn, BTn = (np.array(M.tolist(), dtype=np.float64) for M in (AT, G, BT))
    print("points", points); print("BT=\n", BT); print("G=\n", G); print("AT=\n", AT)
    rng = np.random.default_rng(seed)
    # T tiles of 6x6 x C channels, K=8 output channels
    K = 8
    d = rng.standard_normal((T, C, 6, 6)); g = rng.standard_normal((K, C, 3, 3)) * 0.02
    # direct fp64
    y64 = np.zeros((T, K, 4, 4))
    for a in range(3):
        for b in range(3):
            y64 += np.einsum('tcij,kc->tkij', d[:, :, a:a+4, b:b+4], g[:, :, a, b])
    def run(dt):
        A_, G_, B_ = ATn.astype(dt), Gn.astype(dt), BTn.astype(dt)
        U = np.einsum('ia,kcab,jb->kcij', G_, g.astype(dt), G_).astype(dt)
        V = np.einsum('ia,tcab,jb->tcij', B_, d.astype(dt), B_).astype(dt)
        M = np.einsum('tcij,kcij->tkij', V, U).astype(dt)   # (numpy accumulates in dt)
        return np.einsum('pi,tkij,qj->tkpq', A_, M, A_).astype(dt), V, U
    y32, V32, U32 = run(np.float32)
    yd32 = np.zeros((T, K, 4, 4), np.float32)
    for a in range(3):
        for b in range(3):
            yd32 += np.einsum('tcij,kc->tkij', d[:, :, a:a+4, b:b+4].astype(np.float32), g[:, :, a, b].astype(np.float32))
    print("fwd: wino32 err/max %.2e   direct32 err/max %.2e" % (np.abs(y32 - y64).max() / np.abs(y64).max(), np.abs(yd32 - y64).max() / np.abs(y64).max()))
    # weight gradient: dg[k,c,a,b] = sum_t sum_ij dY[t,k,i,j] d[t,c,i+a,j+b]
    dY = rng.standard_normal((T, K, 4, 4))
    dg64 = np.zeros((K, C, 3, 3))
    for a in range(3):
        for b in range(3):
            dg64[:, :, a, b] = np.einsum('tkij,tcij->kc', dY, d[:, :, a:a+4, b:b+4])
    def wg(dt):
        A_, G_, B_ = ATn.astype(dt), Gn.astype(dt), BTn.astype(dt)
        V = np.einsum('ia,tcab,jb->tcij', B_, d.astype(dt), B_).astype(dt)
        Yt = np.einsum('pi,tkpq,qj->tkij', A_, dY.astype(dt), A_).astype(dt)      # A dY A^T  (A = AT^T: 6x4)
        dU = np.einsum('tkij,tcij->kcij', Yt, V).astype(dt)
        return np.einsum('ia,kcij,jb->kcab', G_, dU, G_).astype(dt)                # G^T dU G
    dg32 = wg(np.float32)
    print("wgrad: wino32 err/max %.2e  (wino64 %.2e)" % (np.abs(dg32 - dg64).max() / np.abs(dg64).max(), np.abs(wg(np.float64) - dg64).max() / np.abs(dg64).max()))
for pts in ([0, 1, -1, 2, -2], [0, 1, -1, sp.Rational(1,2), -sp.Rational(1,2)], [0, 1, -1, 2, -sp.Rational(1,2)], [0,1,-1,sp.Rational(1,2),-2]):
    test(pts)


# ---------------------------------------------------------------------------------------------------------------- general F(m, r)
def general_test(points, m, r, C=256, T=48, K=8, seed=0, show=False):
    n = m + r - 1
    AT, G, BT = mats(points, m, r)
    G, BT = scale_rows(G, BT)
    if show:
        print("BT ="); sp.pprint(BT); print("G ="); sp.pprint(G); print("AT ="); sp.pprint(AT)
    ATn, Gn, BTn = (np.array(M.tolist(), dtype=np.float64) for M in (AT, G, BT))
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((T, C, n, n)); g = rng.standard_normal((K, C, r, r)) * 0.02
    y64 = np.zeros((T, K, m, m))
    for a in range(r):
        for b in range(r):
            y64 += np.einsum('tcij,kc->tkij', d[:, :, a:a+m, b:b+m], g[:, :, a, b])
    dt = np.float32
    A_, G_, B_ = ATn.astype(dt), Gn.astype(dt), BTn.astype(dt)
    U = np.einsum('ia,kcab,jb->kcij', G_, g.astype(dt), G_).astype(dt)
    V = np.einsum('ia,tcab,jb->tcij', B_, d.astype(dt), B_).astype(dt)
    M = np.einsum('tcij,kcij->tkij', V, U).astype(dt)
    y32 = np.einsum('pi,tkij,qj->tkpq', A_, M, A_).astype(dt)
    e = np.abs(y32 - y64).max() / np.abs(y64).max()
    dY = rng.standard_normal((T, K, m, m))
    dg64 = np.zeros((K, C, r, r))
    for a in range(r):
        for b in range(r):
            dg64[:, :, a, b] = np.einsum('tkij,tcij->kc', dY, d[:, :, a:a+m, b:b+m])
    Yt = np.einsum('pi,tkpq,qj->tkij', A_, dY.astype(dt), A_).astype(dt)
    dU = np.einsum('tkij,tcij->kcij', Yt, V).astype(dt)
    dg32 = np.einsum('ia,kcij,jb->kcab', G_, dU, G_).astype(dt)
    ew = np.abs(dg32 - dg64).max() / np.abs(dg64).max()
    print(f"F({m}x{m},{r}x{r}) points {points}: forward err/max {e:.2e}   weight gradient err/max {ew:.2e}   |BT|max {np.abs(BTn).max():.1f}")
h, q, t = sp.Rational(1, 2), sp.Rational(1, 4), sp.Rational(1, 3)
general_test([0, 1, -1, 2, -h], 4, 3)
general_test([0, 1, -1, 2, -2, h], 4, 4, show=True)                 # csrc/wino6.hip w7_*
general_test([0, 1, -1, 2, -2, h, -h], 6, 3, show=True)             # csrc/wino6.hip w8_*
for pts in ([0, 1, -1, h, -h, 2, -q], [0, 1, -1, sp.Rational(3, 2), -sp.Rational(3, 2), sp.Rational(2, 3), -sp.Rational(2, 3)],
            [0, 1, -1, h, -h, sp.Rational(3, 2), -sp.Rational(3, 2)], [0, 1, -1, 2, -2, t, -t]):
    general_test(pts, 6, 3)
