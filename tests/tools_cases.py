"""Seeded inputs shared by tools/gen_golden.py's fixtures and the tests (the
generator defines the same functions; a fixture stores only numbers)."""
import numpy as np

from oracle import problems as pb


def bruss1d(N=32, seed=7):
    rng = np.random.default_rng(seed)
    y0 = np.concatenate([1.0 + 0.5 * rng.random(N), 3.0 + 0.5 * rng.random(N)])
    d = 0.02 * N * N

    def fun(t, y):
        u, v = y[:N], y[N:]
        lap = lambda w: np.roll(w, 1) + np.roll(w, -1) - 2.0 * w  # noqa: E731
        return np.concatenate([1.0 + u * u * v - 4.0 * u + d * lap(u),
                               3.0 * u - u * u * v + d * lap(v)])
    return fun, y0


def single_step_cases():
    rng = np.random.default_rng(1031)
    lam = -rng.random(1031) * 3.0
    y_lin = rng.standard_normal(1031)
    fb, yb = bruss1d()
    return {
        "exp": (lambda t, y: y, 0.0, np.array([1.0]), 0.2),
        "decay3": (lambda t, y: -0.5 * y, 0.0, np.array([2.0, 4.0, 8.0]), 0.37),
        "duffing": (pb.duffing_rhs, 0.3, np.array([0.4, -0.2]), 0.11),
        "bruss1d": (fb, 0.0, yb, 2e-3),
        "lin1031": (lambda t, y: lam * y, 1.0, y_lin, 0.05),
    }


def compare_trajectory(res, nfs, gold, rtol_used, t_rtol=1e-7, noisy=False):
    """solve_ivp result vs a golden run of the reference.

    Normally: identical status / nfev / failed steps and t_k to `t_rtol`.
    `noisy` (duffing_tight: rtol 1e-9 from y = 0, thousands of steps for the
    low-order pairs): the reference's own error estimates are rounding noise at
    the start and its step sequence changes with the BLAS thread count, so
    only the work (nfev within 0.5 %) and the end state are compared."""
    from numpy.testing import assert_allclose
    assert res.status == gold["status"]
    y_end = np.array(gold["y_end_re"]) + 1j * np.array(gold["y_end_im"])
    y_end = y_end if np.iscomplexobj(res.y) else y_end.real
    if noisy:
        assert abs(res.nfev - gold["nfev"]) <= max(3, 0.005 * gold["nfev"])
        assert abs(nfs - gold["nfs"]) <= max(2, 0.05 * gold["nfs"])
        if len(res.t) == len(gold["t"]):
            assert_allclose(res.t, gold["t"], rtol=5e-3)
        assert_allclose(res.y[:, -1], y_end, rtol=1e3 * rtol_used, atol=1e-12)
        return
    assert res.nfev == gold["nfev"]
    assert nfs == gold["nfs"]
    assert_allclose(res.t, gold["t"], rtol=t_rtol)
    assert_allclose(res.y[:, -1], y_end, rtol=1e-3 * rtol_used, atol=1e-12)


def ckdisc_cases():
    """problems for the variable-order CKdisc fixtures"""
    def sawtooth(t, y):                 # discontinuous forcing (non-smooth)
        return np.array([np.sign(np.sin(5.0 * t)) - y[0], y[0] - 0.5 * y[1]])

    def kink(t, y):                     # derivative jump at y = 0.5
        return np.array([-abs(y[0] - 0.5) - 0.1, y[0]])
    fb, yb = bruss1d()
    return {
        "readme": (lambda t, y: -0.5 * y, [0, 10], [2, 4, 8], {}),
        "duffing": (pb.duffing_rhs, [0.0, 20.0], [0.0, 0.0], {}),
        "rational_bwd": (pb.rational_rhs, [5, 1], [1 / 3, 2 / 9],
                         dict(rtol=1e-3, atol=1e-6)),
        "complex": (lambda t, y: -y, [0, 1], [0.5 + 1j],
                    dict(rtol=1e-3, atol=1e-6)),
        "sawtooth": (sawtooth, [0.0, 4.0], [0.0, 1.0], dict(rtol=1e-5, atol=1e-8)),
        "kink": (kink, [0.0, 3.0], [1.0, 0.0], dict(rtol=1e-6, atol=1e-9)),
        "bruss1d": (fb, [0, 0.5], yb, dict(rtol=1e-6, atol=1e-9)),
    }
