"""Test infrastructure: floating-point tolerances for comparing two fp64
evaluations of the same step (device vs oracle / golden vector).

K rows and y_new are compared to ~1e-13 relative.  The embedded error
`h * sum_j E_j K_j` is a CANCELLING sum (for Pr9 at h = 0.2 on y' = y the terms
are O(0.1) and the result O(1e-11)), so two correct summation orders differ by
~eps * sum_j |E_j K_j|, which is far more than 1e-13 relative to the result.
The reference shows the same sensitivity against its own BLAS thread count
(SURVEY.md §7).  The bound below is the forward-error bound of an s-term
dot product, 4 * s * eps * rms(|h| * |K|^T |E| / scale).
"""
import numpy as np

EPS = np.finfo(float).eps


def error_norm_atol(E, K, h, scale):
    """absolute tolerance on the weighted RMS error norm"""
    m = min(len(E), K.shape[0])
    while m > 0 and E[m - 1] == 0:
        m -= 1
    mag = abs(h) * (np.abs(K[:m]).T @ np.abs(E[:m]))
    ratio = mag / scale
    cond = (np.real(ratio @ ratio) / max(ratio.size, 1)) ** 0.5
    return 4.0 * m * EPS * cond + 1e-300


def step_scale(rtol, atol, y, y_new):
    return atol + rtol * np.maximum(np.abs(y), np.abs(y_new))


def check_step(dev, ref_K, ref_y_new, ref_err, ref_h_abs, y_old, h, rtol, atol,
               cls=None, k_rtol=1e-13, lipschitz=0.0):
    """assert that a device solver `dev` that just took ONE step from the same
    (t, y, f, h) agrees with reference values; returns nothing.

    `lipschitz`: Lipschitz constant of the RHS.  A stage derivative is
    f(y_stage); two correctly rounded y_stage differ by ~eps*|y|, which the RHS
    amplifies by up to L (4e6 for the Brusselator at N = 2236), so K can only
    agree to k_rtol*max|K| + 8*eps*L*max|y|."""
    cls = cls or type(dev)
    assert dev.h_previous == h, "the compared steps used different h"
    kmax = np.abs(ref_K).max()
    k_atol = k_rtol * kmax + 8 * EPS * lipschitz * np.abs(ref_y_new).max()
    np.testing.assert_allclose(dev.K, ref_K, rtol=0, atol=k_atol)
    np.testing.assert_allclose(dev.y, ref_y_new, rtol=k_rtol,
                               atol=abs(h) * k_atol)
    scale = step_scale(rtol, atol, y_old, ref_y_new)
    tol = error_norm_atol(cls.E, ref_K, h, scale)
    if lipschitz:
        # rounding in K, amplified, enters the error sum as well: up to
        # |h| * sum|E| * k_atol per element, weighted like the norm itself --
        # rms(1 / scale), not 1 / scale.min() (on the
        # Brusselator the smallest scale is 30 x below the rms weight)
        tol += abs(h) * np.abs(cls.E).sum() * k_atol * np.sqrt(np.mean(1.0 / scale ** 2))
    assert abs(dev.error_norm_old - ref_err) <= tol, (
        dev.error_norm_old, ref_err, tol)
    # the next step size is a smooth function of the error norm
    rel = tol / max(ref_err, 1e-300)
    np.testing.assert_allclose(dev.h_abs, ref_h_abs, rtol=min(1.0, rel) + 1e-12)
