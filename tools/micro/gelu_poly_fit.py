"""Derives the coefficients of csrc/gemm_bf16.hip::gelu2 -- GELU(x) = x (1/2 + t Q(z)), t = clamp(x, -U, U), z = 2 t^2 / U^2 - 1 --
and prints the error of the float32 evaluation against erf in float64.

    python tools/micro/gelu_poly_fit.py [U=5] [degree=11]
"""
import sys

import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf


def fit(U=5.0, m=11):
    k = np.arange(8000)
    ss = (np.cos(np.pi * (k + 0.5) / 8000) + 1) * U * U / 2          # Chebyshev nodes in s = t^2 on [0, U^2]
    ts = np.sqrt(ss)
    q = 0.5 * erf(ts / np.sqrt(2)) / ts                             # (Phi(t) - 1/2) / t
    cf = C.chebfit(ss * 2 / (U * U) - 1, q, m, w=ts ** 2)            # weight t^2: the error of GELU is |x| t dQ
    return C.cheb2poly(cf).astype(np.float32)                       # powers of z


def gelu_f32(x, coef, U=5.0):
    """float32 evaluation as the kernel does it (Horner in z; numpy rounds every product and sum, the kernel fuses them)"""
    f = np.float32
    x = x.astype(f)
    t = np.clip(x, f(-U), f(U))
    z = (t * t * f(2 / (U * U)) - f(1)).astype(f)
    q = np.full_like(x, coef[-1])
    for c in coef[-2::-1]:
        q = (q * z + c).astype(f)
    return (x * (t * q + f(0.5)).astype(f)).astype(f)


if __name__ == "__main__":
    U = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    coef = fit(U, m)
    print("z = %.9ef * t^2 - 1" % (2 / (U * U)))
    print("C = {" + ", ".join("%.9ef" % c for c in coef) + "}")
    x = np.linspace(-12, 12, 2400001)
    err = np.abs(gelu_f32(x, coef, U).astype(np.float64) - x * 0.5 * (1 + erf(x / np.sqrt(2))))
    inner = np.abs(x) <= 8
    print("max |error| on [-8, 8]: %.2e at x = %.3f;  on [-12, 12]: %.2e" % (err[inner].max(), x[inner][err[inner].argmax()], err.max()))
