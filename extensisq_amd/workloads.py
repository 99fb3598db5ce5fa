"""Synthetic inputs of the BASELINE.json workloads: initial
states and step-size choices.  Host-side input synthesis only -- the right-hand
sides themselves are the device plugins in extensisq_amd.device."""
import numpy as np

from .device import Brusselator2D, Diffusion3D, Heat2D


def heat2d_y0(N, seed=1234):
    """sin(pi x) sin(pi y) + 0.1 * N(0,1), interior grid x_i = i/(N+1)"""
    x = np.arange(1, N + 1) / (N + 1)
    rng = np.random.default_rng(seed)
    u0 = np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
    return (u0 + 0.1 * rng.standard_normal((N, N))).ravel()


def bruss2d_y0(N, shard=0):
    """Hairer-Wanner BRUSS-2D start: u = 22 y (1-y)^1.5, v = 27 x (1-x)^1.5 on
    cell centres; `shard` > 0 scales the amplitudes (independent IVPs of a
    lock-step batch)"""
    c = (np.arange(N) + 0.5) / N
    yy, xx = np.meshgrid(c, c, indexing="ij")
    amp = 1.0 + 0.02 * shard
    u0 = amp * 22.0 * yy * (1.0 - yy) ** 1.5
    v0 = amp * 27.0 * xx * (1.0 - xx) ** 1.5
    return np.concatenate([u0.ravel(), v0.ravel()])


def diff3d_y0(N):
    x = np.arange(1, N + 1) / (N + 1)
    s = np.sin(np.pi * x)
    return (s[:, None, None] * s[None, :, None] * s[None, None, :]).ravel()


def pr8_brusselator(N=2236, shard=0):
    """north-star workload: (rhs, y0, h) with h = 1/rho so that every step is
    accepted and stability-safe"""
    rhs = Brusselator2D(N)
    return rhs, bruss2d_y0(N, shard), 1.0 / rhs.spectral_radius()


def ts5_heat(N=1000, seed=1234):
    rhs = Heat2D(N)
    return rhs, heat2d_y0(N, seed), 1.0 / rhs.spectral_radius()


def rkc_diffusion(N=159, m_target=100):
    """SSV2stab config: first step chosen so that m = 1+int(sqrt(1.54 h rho + 1))
    is about m_target"""
    rhs = Diffusion3D(N)
    rho = rhs.spectral_radius()
    h = ((m_target - 1) ** 2 - 1 + 0.5 * (2 * m_target - 1)) / (1.54 * rho)
    return rhs, diff3d_y0(N), h, rho
