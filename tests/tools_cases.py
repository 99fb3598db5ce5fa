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
