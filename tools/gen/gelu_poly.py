"""Coefficients of gelu_cdf_fast2 (csrc/common.h): Phi(u) - 1/2 = u Q(u^2) on |u| <= C, Q of degree DEG by weighted least squares iterated to a
minimax fit (Lawson), rescaled so that C Q(C^2) = 1/2 exactly (the clamped argument saturates at Phi = 0 / 1), then checked in fp32 Horner form.
python tools/gen/gelu_poly.py  ->  the constants + the measured error bounds (|Phi error| <= 1.5e-5, |GELU error| <= 5.6e-5 at |u| ~ 4)."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf

CL, DEG = 4.2, 8


def Phi(v):
    return 0.5 * (1 + erf(v / np.sqrt(2)))


n = 8000
x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 + 0.5
v = x * CL
v = v[v > 1e-9]
s = v * v
g = (Phi(v) - 0.5) / v
V = C.chebvander(2 * s / (CL * CL) - 1, DEG)
w = np.ones_like(v)
for _ in range(200):
    coef, *_ = np.linalg.lstsq(V * (w * v)[:, None], g * w * v, rcond=None)
    err = (V @ coef - g) * v
    w = w * (0.2 + np.abs(err) / np.abs(err).max())
    w /= w.max()
pw = C.Chebyshev(coef, domain=[0, CL * CL]).convert(kind=P.Polynomial, domain=[-1, 1], window=[-1, 1]).coef
pw = pw * (0.5 / (CL * np.polyval(pw[::-1], CL * CL)))
pw32 = pw.astype(np.float32)
vt = np.linspace(-8, 8, 1600001)
vc = np.clip(vt.astype(np.float32), np.float32(-CL), np.float32(CL))
s32 = vc * vc
acc = np.full_like(s32, pw32[-1])
for k in range(len(pw32) - 2, -1, -1):
    acc = acc * s32 + pw32[k]
cdf = (vc * acc + np.float32(0.5)).astype(np.float64)
print("coefficients, lowest power of u^2 first:", ", ".join("%.9ef" % c for c in pw32))
print("fp32 Horner: max |Phi error| %.3g, cdf in [%.3g, %.9f], max |GELU error| %.3g" % (np.abs(cdf - Phi(vt)).max(), cdf.min(), cdf.max(), np.abs(vt * cdf - vt * Phi(vt)).max()))
