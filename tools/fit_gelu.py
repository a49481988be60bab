#!/usr/bin/env python3
"""Coefficients of csrc/sc_common.h gelu_bf2 / gelu_bf: gelu(x) ~ x * sigmoid(x (c1 + c3 t + c5 t^2)), t = min(x^2, 64) - minimax fit of
|x Phi(x) - approx| over [-12, 12] (Nelder-Mead from the classical logistic approximation of the normal CDF), then the error of the
fp32 evaluation as the kernels do it (coefficients times -log2(e), exp2, rcp).  Needs scipy; CPU only."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x = np.linspace(-12, 12, 48001)
gelu = x * 0.5 * (1 + erf(x / np.sqrt(2)))


def approx(c, x):
    t = np.minimum(x * x, 64.0)
    return x / (1 + np.exp(-(x * (c[0] + t * (c[1] + t * c[2])))))


best = None
np.random.seed(0)
for _ in range(12):
    c0 = np.array([1.5976, 0.07056 * 1.05, -7e-4]) * (1 + 0.003 * np.random.randn(3))
    r = minimize(lambda c: np.max(np.abs(approx(c, x) - gelu)), c0, method="Nelder-Mead", options=dict(xatol=1e-12, fatol=1e-14, maxiter=40000))
    best = r if best is None or r.fun < best.fun else best
c = best.x
K = np.float32(-c * 1.4426950408889634)
print("c =", c.tolist(), " max |error| %.3e" % best.fun)
print("K = -c log2(e) as fp32:", [float(k) for k in K])
xf = np.linspace(-12, 12, 2000001).astype(np.float32)
t = np.minimum(xf * xf, np.float32(64))
p = (K[2] * t + K[1]).astype(np.float32)
p = (p * t + K[0]).astype(np.float32)
w = (p * xf).astype(np.float32)
g = (xf * (np.float32(1) / (np.exp2(w.astype(np.float64)).astype(np.float32) + np.float32(1)))).astype(np.float32)
ex = xf.astype(np.float64) * 0.5 * (1 + erf(xf.astype(np.float64) / np.sqrt(2)))
err = np.abs(g - ex)
print("fp32 evaluation: max |error| %.3e at x = %.3f" % (err.max(), xf[err.argmax()]))
rel = err / np.maximum(np.abs(ex), 1e-30)
for lo, hi in ((-1, 12), (-2, -1), (-3, -2), (-4, -3)):
    m = (xf >= lo) & (xf < hi) & (np.abs(ex) > 1e-7)
    print("  x in [%g, %g): max relative error %.2e, max absolute %.2e" % (lo, hi, rel[m].max(), err[m].max()))
