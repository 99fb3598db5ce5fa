"""Seeded problems for the stiffness-diagnosis fixtures (shared by
tools/gen_golden.py and the GPU tests; a fixture stores only numbers)."""
import numpy as np


def stiffness_cases():
    lam = -np.logspace(0, 3.3, 20)              # -1 ... -2000

    def stiff_real(t, y):
        return lam * (y - np.cos(t))

    w = 150.0

    def oscillator(t, y):                       # eigenvalues +- i w
        return np.array([y[1], -w * w * y[0], y[3], -0.25 * w * w * y[2]])

    def mild(t, y):
        return np.array([-0.5 * y[0] + np.sin(t), -0.1 * y[1] + y[0]])

    def spiral(t, y):                           # eigenvalues -40 +- 300 i
        return np.array([-40.0 * y[0] - 300.0 * y[1], 300.0 * y[0] - 40.0 * y[1]])

    return {
        "stiff_real": (stiff_real, [0.0, 6.0], np.ones(20), {}),
        "oscillator": (oscillator, [0.0, 8.0], np.array([1.0, 0.0, 0.5, 1.0]),
                       dict(rtol=1e-5, atol=1e-8)),
        "mild": (mild, [0.0, 200.0], np.array([1.0, 0.0]),
                 dict(nfev_stiff_detect=300)),
        "spiral": (spiral, [0.0, 12.0], np.array([1.0, 0.0]), {}),
    }
