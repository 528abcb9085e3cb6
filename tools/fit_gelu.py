"""Fit of the single-range GELU polynomial used by gelu_erf_1r (gmf_amd/csrc/enc_common.hpp): p(t) ~ log2(erfc(t))/t on
[0, 4.3], weighted minimax on the absolute erfc error (Lawson iterations), with an fp32 Horner accuracy check.  CPU only."""
import numpy as np
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as C
TMAX = 4.3
t = np.linspace(1e-9, TMAX, 200001)
f = np.log2(erfc(t)) / t                  # r2(t)/t,  r2 = log2(erfc)
# absolute error target on E = erfc: dE = E * ln2 * t * dp  -> weight = E*t
w = erfc(t) * t + 1e-12
for deg in (8, 9, 10, 11):
    u = 2 * t / TMAX - 1
    ww = w.copy()
    for it in range(60):                  # Lawson iterations toward weighted minimax
        c = C.chebfit(u, f, deg, w=ww)
        err = np.abs((C.chebval(u, c) - f) * w)
        ww = ww * (0.5 + err / err.max())
    dE = np.abs((C.chebval(u, c) - f)) * erfc(t) * t * np.log(2)
    # monomial coefficients in t
    p = C.cheb2poly(c)
    # convert from u to t: u = a t + b
    a, b = 2 / TMAX, -1.0
    P = np.polynomial.Polynomial(p)(np.polynomial.Polynomial([b, a]))
    coef = P.coef
    # fp32 Horner evaluation check over x
    x = np.concatenate([np.linspace(-12, 12, 2000001), np.random.default_rng(0).normal(0, 2, 1000000)]).astype(np.float32)
    tt = np.minimum(np.abs(x) * np.float32(0.70710678118654752440), np.float32(TMAX)).astype(np.float32)
    cf = coef.astype(np.float32)
    acc = np.full_like(tt, cf[-1])
    for k in range(len(cf) - 2, -1, -1):
        acc = (acc * tt + cf[k]).astype(np.float32)          # (no fma: slightly pessimistic)
    E = np.exp2((acc * tt).astype(np.float32)).astype(np.float32)
    g = (np.maximum(x, 0) - (np.float32(0.5) * np.abs(x)) * E).astype(np.float32)
    xd = x.astype(np.float64)
    gref = 0.5 * xd * (1 + erf(xd / np.sqrt(2)))
    # the reference's own fp32 formula: x * 0.5 * (1 + erf(x * sqrt(1/2)))
    gfp32 = (x * np.float32(0.5) * (np.float32(1) + erf((x * np.float32(0.7071067811865476)).astype(np.float32).astype(np.float64)).astype(np.float32))).astype(np.float32)
    print(f"deg {deg}: max dE (double poly) {dE.max():.2e}; fp32 gelu max abs err {np.abs(g - gref).max():.2e}  (exact-formula fp32: {np.abs(gfp32 - gref).max():.2e});"
          f" max rel-to-(|x|+1) {np.abs((g - gref) / (np.abs(xd) + 1)).max():.2e} vs {np.abs((gfp32 - gref) / (np.abs(xd) + 1)).max():.2e}")
    if deg in (8, 9):
        print("  coef:", ", ".join(f"{v:.9e}f" for v in cf))
