"""The reference's two demo problems of docs/Demo_SSV2stab.ipynb as user plugins of
extensisq_amd (examples/ssv2stab_demo_plugins.hip: two pointwise functors on
csrc/esq_stencil3d.hpp).  `build()` compiles the plugin library with hipcc (once
per source version, into a cache directory); `tanh_heat(N)` and `combustion(N)` return
the device right-hand sides together with the initial states the notebook uses:

    import examples.ssv2stab_demo_plugins as demo
    rhs, y0, rho_jac = demo.tanh_heat(39)
    sol = solve_ivp(rhs, (0, 0.7), y0, method=esq.SSV2stab, rtol=1e-3, atol=1e-3,
                    rho_jac=rho_jac, const_jac=True)

Both objects are also callable on host arrays (`rhs(t, y)`), like every DeviceRHS.
"""
import ctypes as C
import hashlib
import os
import shutil
import subprocess
import tempfile

import numpy as np

import extensisq_amd as esq

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, ".."))
SRC = os.path.join(HERE, "ssv2stab_demo_plugins.hip")
_LIB = {}


def build(cache_dir=None):
    """-> path of the compiled plugin library (hipcc, gfx950)"""
    csrc = os.path.join(ROOT, "extensisq_amd", "csrc")
    digest = hashlib.sha256()
    for path in [SRC] + sorted(os.path.join(csrc, f) for f in os.listdir(csrc)
                               if f.endswith(".hpp")):
        with open(path, "rb") as fh:
            digest.update(fh.read())
    cache_dir = cache_dir or os.environ.get("ESQ_EXAMPLE_CACHE") or os.path.join(
        tempfile.gettempdir(), "extensisq_amd_examples")
    os.makedirs(cache_dir, exist_ok=True)
    so = os.path.join(cache_dir, f"libssv2stab_demo_{digest.hexdigest()[:16]}.so")
    if not os.path.exists(so):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        tmp = so + f".{os.getpid()}.tmp"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                        "-ffp-contract=off", "-I", csrc, SRC, "-o", tmp], check=True)
        os.replace(tmp, so)
    return so


def _lib():
    if "lib" not in _LIB:
        _LIB["lib"] = C.CDLL(build())
    return _LIB["lib"]


def _ptr(fn):
    return C.cast(fn, C.c_void_p)


class _DemoPlugin(esq.CFunctionRHS):
    """esq_rhs_fn + the fused entry (every epilogue kind) + the Chebyshev stage entry of
    one of the two demo problems"""
    _fuse_default = True
    _fuse_query = True

    def __init__(self, prefix, user, n):
        lib = _lib()
        self._entries = {k: getattr(lib, f"{prefix}_{k}") for k in ("rhs", "fused", "rkc")}
        self._user_struct = user                      # kept alive with the object
        super().__init__(_ptr(self._entries["rhs"]).value, C.addressof(user), n)

    def _fused_entry(self, lib_):
        return _ptr(self._entries["fused"])

    def _rkc_entry(self, lib_):
        return _ptr(self._entries["rkc"])


class _ChainPlugin(_DemoPlugin):
    """... + the chain entries of a one-field homogeneous functor"""
    _chain_caps = esq._lib.CHAIN_CAP_QUERY | 1 | 2      # from the state, rows unwritten
    _rkc_chain_depth = 4

    def __init__(self, prefix, user, n):
        super().__init__(prefix, user, n)
        lib = _lib()
        self._entries["chain"] = getattr(lib, f"{prefix}_chain")
        self._entries["rkc_chain"] = getattr(lib, f"{prefix}_rkc_chain")

    def _chain_entry(self, lib_):
        return _ptr(self._entries["chain"])

    def _rkc_chain_entry(self, lib_):
        depth = int(os.environ.get("ESQ_RKC_MAXDEPTH", self._rkc_chain_depth))
        return (_ptr(self._entries["rkc_chain"]),
                depth | esq._lib.RKC_CHAIN_FIRST | esq._lib.RKC_CHAIN_LAST)


class _AnisoFn(C.Structure):
    _fields_ = [("ci", C.c_double), ("cj", C.c_double), ("cl", C.c_double)]


class _AnisoUser(C.Structure):
    _fields_ = [("N", C.c_int), ("fn", _AnisoFn)]


def aniso_diffusion(N, cx=1.0, cy=0.5, cz=2.0):
    """u_t = cx u_xx + cy u_yy + cz u_zz on N^3 interior points, homogeneous Dirichlet: a
    user functor with the header's fast sweeps and chain sweeps.
    -> (device RHS, NumPy twin with the same operation order, spectral radius)"""
    h2 = (N + 1.0) ** 2
    ci, cj, cl = cy * h2, cx * h2, cz * h2          # array axes (i, j, l) <-> (y, x, z)
    rhs = _ChainPlugin("aniso3d", _AnisoUser(N, _AnisoFn(ci, cj, cl)), N ** 3)

    def twin(t, y):
        w = np.zeros((N + 2,) * 3)
        w[1:-1, 1:-1, 1:-1] = y.reshape(N, N, N)
        c = w[1:-1, 1:-1, 1:-1]
        di = (w[:-2, 1:-1, 1:-1] + w[2:, 1:-1, 1:-1]) - 2.0 * c
        dj = (w[1:-1, :-2, 1:-1] + w[1:-1, 2:, 1:-1]) - 2.0 * c
        dl = (w[1:-1, 1:-1, :-2] + w[1:-1, 1:-1, 2:]) - 2.0 * c
        return ((ci * di + cj * dj) + cl * dl).reshape(-1)
    return rhs, twin, 4.0 * (ci + cj + cl)


class _TanhFn(C.Structure):
    _fields_ = [("inv_h2", C.c_double), ("step", C.c_double)]


class _TanhUser(C.Structure):
    _fields_ = [("N", C.c_int), ("fn", _TanhFn)]


class _CombFn(C.Structure):
    _fields_ = [("inv_h2", C.c_double), ("damkohler", C.c_double), ("delta", C.c_double),
                ("alpha", C.c_double), ("lewis", C.c_double)]


class _CombUser(C.Structure):
    _fields_ = [("N", C.c_int), ("fn", _CombFn)]


def tanh_heat(N=39):
    """3-D heat equation with a travelling tanh front and time-dependent Dirichlet
    data (Demo_SSV2stab.ipynb, "heat problem"; published table :350-356).
    -> (device RHS, y0, rho_jac)"""
    step = 1.0 / (N + 1)
    inv_h2 = (N + 1.0) ** 2
    rhs = _DemoPlugin("tanh3d", _TanhUser(N, _TanhFn(inv_h2, step)), N ** 3)
    x = np.linspace(0.0, 1.0, N + 2)
    X, Y, Z = np.meshgrid(x, x, x)
    y0 = np.tanh(5 * X + 10 * Y + 7.5 * Z - 2.5)[1:-1, 1:-1, 1:-1].copy().reshape(-1)

    def rho_jac(t, y):
        return 12.0 * inv_h2
    return rhs, y0, rho_jac


def combustion(N=40, lewis=0.9, alpha=1.0, delta=20.0, rate=5.0):
    """3-D combustion benchmark of the RKC paper (Demo_SSV2stab.ipynb cells 1-3;
    published table :207-211): concentration and temperature on N^3 cells.
    -> (device RHS, y0)"""
    damkohler = rate * np.exp(delta) / (alpha * delta)
    inv_h2 = (N + 0.5) ** 2
    rhs = _DemoPlugin("comb3d", _CombUser(N, _CombFn(inv_h2, damkohler, delta, alpha, lewis)),
                      2 * N ** 3)
    return rhs, np.ones(2 * N ** 3)
