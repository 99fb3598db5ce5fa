"""extensisq_amd -- MI355X-native explicit Runge-Kutta solvers for
`scipy.integrate.solve_ivp`.

Drop-in for the explicit-RK hot path of extensisq (BS5, Ts5, Pr7, Pr8, Pr9,
SSV2stab): the same `OdeSolver` classes, Butcher-tableau class attributes and
counters, with the per-step vector arithmetic running as hand-written HIP
kernels (gfx950) behind a ctypes C ABI (include/extensisq_amd.h).
"""
from .common import NFS, NFI, NLS, LockstepGroup, RungeKutta  # noqa: F401
from .tsitouras import Ts5  # noqa: F401
from .bogacki import BS5  # noqa: F401
from .prince import Pr7, Pr8, Pr9  # noqa: F401
from .sommeijer import SSV2stab  # noqa: F401
from .cash import CK5, CKdisc  # noqa: F401
from .merson import Me4  # noqa: F401
from .calvo import CFMR7osc  # noqa: F401
from .device import (Brusselator2D, CFunctionRHS, DeviceContext,  # noqa: F401
                     DeviceRHS, DiagonalLinear, Diffusion3D, Heat2D)
from ._lib import DeviceError  # noqa: F401

__version__ = "0.1.0"
__all__ = ["BS5", "Ts5", "Pr7", "Pr8", "Pr9", "SSV2stab", "CK5", "CKdisc", "Me4",
           "CFMR7osc", "RungeKutta",
           "NFS", "DeviceRHS", "Heat2D", "Brusselator2D", "Diffusion3D",
           "DiagonalLinear", "CFunctionRHS", "LockstepGroup", "DeviceError"]
