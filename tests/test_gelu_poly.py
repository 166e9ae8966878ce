"""The GELU of the GEMM write-outs and of the fused Swin MLP (csrc/gelu_poly.h::gelu2) is a polynomial, not erf: this pins its coefficients to the
generating script and its float32 error to the bound the kernel's comment states.  (CPU; the kernel itself is compared
with torch's erf GELU in tests/test_gpu_kernels.py::test_gemm_activation_epilogues.)"""
import os
import re
import sys

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "micro"))
import gelu_poly_fit  # noqa: E402


def _kernel_coefficients():
    src = open(os.path.join(ROOT, "vsc22-submission_amd", "csrc", "gelu_poly.h")).read()
    assert "GELU_DEG = 8;" in src and "GELU_U = 4.5f" in src
    body = re.search(r"constexpr float GELU_C\[GELU_DEG \+ 1\] = \{([^}]*)\}", src).group(1)
    return np.array([float(v.strip().rstrip("f")) for v in body.split(",")], dtype=np.float32)


def test_kernel_coefficients_are_the_fitted_ones():
    np.testing.assert_allclose(_kernel_coefficients(), gelu_poly_fit.fit(4.5, 8), rtol=2e-6, atol=1e-9)


def test_float32_error_bound():
    coef = _kernel_coefficients()
    x = np.concatenate([np.linspace(-40, 40, 800001), np.linspace(-1e-2, 1e-2, 20001)])
    exact = x * 0.5 * (1 + erf(x / np.sqrt(2)))
    got = gelu_poly_fit.gelu_f32(x, coef, 4.5).astype(np.float64)
    assert np.all(np.abs(got - exact) <= 4.3e-5 + 3.5e-6 * np.abs(x))
    # small arguments: relative accuracy (GELU ~ x / 2 there), far inside bf16's 2^-9
    small = (np.abs(x) < 1e-2) & (x != 0)
    assert np.all(np.abs(got[small] - exact[small]) <= 2e-5 * np.abs(exact[small]))
