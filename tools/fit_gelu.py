#!/usr/bin/env python3
"""Coefficients of csrc/sc_common.h gelu_bf2 / gelu_bf: gelu(x) ~ x * sigmoid(x p(t)), t = min(x^2, clamp), p a polynomial of n terms.
Fit: iteratively re-weighted least squares of p against logit(Phi(x)) / x (weights: the sensitivity of gelu to p, x^2 Phi (1 - Phi)),
then a Nelder-Mead minimax refinement of |x Phi(x) - approx| over [-12, 12]; finally the error of the fp32 evaluation as the kernels do
it (coefficients times -log2(e), exp2, rcp).  Needs scipy; CPU only.   python tools/fit_gelu.py [terms=5] [clamp=36]"""
import sys
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf, log_ndtr, ndtr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
clamp = float(sys.argv[2]) if len(sys.argv) > 2 else 36.0
x = np.linspace(1e-3, 9, 20000)
Phi = ndtr(x)
g = (log_ndtr(x) - log_ndtr(-x)) / x
t = x * x
xx = np.linspace(-12, 12, 96001)
gel = xx * ndtr(xx)


def model(c, x):
    return x / (1 + np.exp(-(x * np.polyval(c[::-1], np.minimum(x * x, clamp)))))


m = t <= clamp
sens = (x * x * Phi * (1 - Phi))[m]
w = sens.copy()
A = np.vander(t[m], n, increasing=True)
for _ in range(60):
    c, *_ = np.linalg.lstsq(A * w[:, None], g[m] * w, rcond=None)
    err = np.abs((A @ c - g[m]) * sens)
    w = w * (1 + 4 * err / err.max())
    w /= w.max()
r = minimize(lambda c: np.max(np.abs(model(c, xx) - gel)), c, method="Nelder-Mead", options=dict(xatol=1e-14, fatol=1e-16, maxiter=40000, maxfev=40000))
if r.fun < np.max(np.abs(model(c, xx) - gel)):
    c = r.x
print("c =", [float("%.10g" % v) for v in c], " max |error| %.3e" % np.max(np.abs(model(c, xx) - gel)))
K = np.float32(-c * 1.4426950408889634)
print("K = -c log2(e) as fp32:", [float(k) for k in K])
xf = np.linspace(-12, 12, 2000001).astype(np.float32)
tf = np.minimum(xf * xf, np.float32(clamp))
p = np.full_like(xf, K[-1])
for k in K[-2::-1]:
    p = (p * tf + k).astype(np.float32)
wv = (p * xf).astype(np.float32)
gv = (xf * (np.float32(1) / (np.exp2(wv.astype(np.float64)).astype(np.float32) + np.float32(1)))).astype(np.float32)
ex = xf.astype(np.float64) * 0.5 * (1 + erf(xf.astype(np.float64) / np.sqrt(2)))
e = np.abs(gv - ex)
print("fp32 evaluation: max |error| %.3e at x = %.3f" % (e.max(), xf[e.argmax()]))
rel = e / np.maximum(np.abs(ex), 1e-30)
for lo, hi in ((-1, 12), (-2, -1), (-3, -2), (-4, -3)):
    mm = (xf >= lo) & (xf < hi) & (np.abs(ex) > 1e-7)
    print("  x in [%g, %g): max relative error %.2e, max absolute %.2e" % (lo, hi, rel[mm].max(), e[mm].max()))
