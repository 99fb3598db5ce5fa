"""ctypes binding of libextensisq_amd.so (the C ABI in include/extensisq_amd.h).

The library is the ONLY compute back end of this package: if it is missing or
a call fails, an exception is raised -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ESQ_LIB: another build of the same library (A/B runs of compile-time variants,
# `make VARIANT=...` in csrc/); must sit next to the package like the default one
LIB_PATH = os.environ.get("ESQ_LIB") or os.path.join(_HERE, "libextensisq_amd.so")

ABI_VERSION = 9
OP_SUM, OP_MAX, OP_MIN = 0, 1, 2
EPI_STAGE, EPI_BLOCK, EPI_SOLERR, EPI_ERRNORM = 1, 2, 3, 4
EPI_RKCERR = 6
FUSE_ALL = 0x5e
FUSE_SRC = 0x20
FUSE_QUERY = 0x80
RKC_CHAIN_FIRST, RKC_CHAIN_LAST = 0x100, 0x200
CHAIN_CAP_ALL, CHAIN_CAP_QUERY = 15, 16
CHAIN_CAP_PRE, CHAIN_CAP_ERRNORM = 32, 64
CREATE_HOST_SLAB = 1
SLOT_K, SLOT_Y, SLOT_YNEW, SLOT_YSTAGE, SLOT_ATOL, SLOT_WORK = range(6)
PROF_STAGE, PROF_RHS, PROF_SOLERR, PROF_RKC = range(4)
VEC_NONE, VEC_Y, VEC_YNEW, VEC_YSTAGE, VEC_WORK = -1, -2, -3, -4, -5

RHS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                     C.c_size_t, C.c_void_p)

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol include/extensisq_amd.h declares
SIGNATURES = {
    "esq_abi_version": (C.c_int, []),
    "esq_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "esq_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "esq_create": (C.c_int, [_vpp, C.c_int, C.c_size_t, C.c_int, C.c_int]),
    "esq_create2": (C.c_int, [_vpp, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int,
                              C.c_char_p]),
    "esq_option_level": (C.c_int, [C.c_char_p]),
    "esq_rhs_set_options": (C.c_int, [_vp, C.c_char_p]),
    "esq_destroy": (C.c_int, [_vp]),
    "esq_last_error": (C.c_char_p, [_vp]),
    "esq_synchronize": (C.c_int, [_vp]),
    "esq_vector_len": (C.c_size_t, [_vp]),
    "esq_upload": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "esq_download": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "esq_snapshot_begin": (C.c_int, [_vp, C.c_int, C.c_int, _vpp]),
    "esq_snapshot_copy": (C.c_int, [_vp, _vp, C.c_int]),
    "esq_release_cached_memory": (C.c_int, [_vp]),
    "esq_copy_lane_info": (C.c_int, [C.c_int, _vp, _vp, _vp]),
    "esq_host_pin": (C.c_int, [_vp, C.c_size_t]),
    "esq_host_unpin": (C.c_int, [_vp]),
    "esq_copy": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "esq_rk_set_tableau": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    "esq_set_tol": (C.c_int, [_vp, C.c_double, _vp, C.c_size_t]),
    "esq_set_rhs": (C.c_int, [_vp, _vp, _vp]),
    "esq_set_rhs_fused": (C.c_int, [_vp, _vp, C.c_int]),
    "esq_set_rhs_rkc": (C.c_int, [_vp, _vp]),
    "esq_set_rhs_chain": (C.c_int, [_vp, _vp, C.c_int]),
    "esq_set_rhs_rkc_chain": (C.c_int, [_vp, _vp, C.c_int]),
    "esq_rk_stage_accumulate": (C.c_int, [_vp, C.c_int, C.c_double]),
    "esq_rk_block_plan": (C.c_int, [_vp, C.POINTER(C.c_int), C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "esq_rk_eval_rhs": (C.c_int, [_vp, C.c_int, C.c_double, C.c_int, C.c_int]),
    "esq_rk_stages": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double]),
    "esq_rk_solution": (C.c_int, [_vp, C.c_double]),
    "esq_rk_error_norm": (C.c_int, [_vp, C.c_double, _dp]),
    "esq_rk_solution_error": (C.c_int, [_vp, C.c_double, C.c_double, _dp]),
    "esq_rk_solution_error_ahead": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, _dp]),
    "esq_rk_set_launch_ahead": (C.c_int, [_vp, C.c_int]),
    "esq_rk_launch_ahead_stats": (C.c_int, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "esq_rk_pre_error": (C.c_int, [_vp, C.c_double, _vp, _vp, C.c_int, _dp]),
    "esq_rk_attempt": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, _dp, _dp]),
    "esq_rk_set_pre": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "esq_rk_pre_result": (C.c_int, [_vp, _dp]),
    "esq_rk_custom_sol_err": (C.c_int, [_vp, C.c_double, _vp, _vp, C.c_int, C.c_int,
                                        _dp]),
    "esq_rk_accept": (C.c_int, [_vp, C.c_double, C.c_int, C.c_double]),
    "esq_rk_error_vector": (C.c_int, [_vp, C.c_double, C.c_int]),
    "esq_rk_row_id": (C.c_int, [_vp, C.c_int, C.c_int]),
    "esq_rk_download_last_K": (C.c_int, [_vp, C.c_int, _vp]),
    "esq_rk_lazy_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "esq_plan_describe": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _vp, _vp, _vp, _vp,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    _vp, _vp, C.c_int, C.c_char_p, C.c_size_t]),
    "esq_step_dry_run": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _vp, _vp, _vp, _vp,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _vp, _vp, C.c_int, _vp, C.c_int, C.c_char_p,
                                   C.c_size_t]),
    "esq_dense_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_double, C.c_int,
                                   _vpp]),
    "esq_dense_create_vecs": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vpp]),
    "esq_dense_eval": (C.c_int, [_vp, C.c_double, _vp]),
    "esq_dense_download": (C.c_int, [_vp, _vp]),
    "esq_dense_destroy": (C.c_int, [_vp]),
    "esq_rk_dense_stage": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_double]),
    "esq_rk_dense_eval": (C.c_int, [_vp, C.c_int, C.c_double]),
    "esq_rk_upload_last_K": (C.c_int, [_vp, C.c_int, _vp]),
    "esq_rkc_first_stage": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double]),
    "esq_rkc_stage": (C.c_int, [_vp] + [C.c_int] * 6 + [C.c_double] * 4),
    "esq_rkc_stages": (C.c_int, [_vp] + [C.c_int] * 6 + [C.c_double, C.c_int, _vp,
                                                         C.POINTER(C.c_int)]),
    "esq_rkc_error_norm": (C.c_int, [_vp] + [C.c_int] * 4 + [C.c_double, _dp]),
    "esq_rkc_stages_end": (C.c_int, [_vp] + [C.c_int] * 6 + [C.c_double, C.c_int, _vp,
                                    C.c_double, C.c_double, _vp, _vp, _vp]),
    "esq_rkc_guess_next": (C.c_int, [_vp, C.c_double, C.c_int, _vp]),
    "esq_rkc_plan_describe": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.c_size_t]),
    "esq_rkc_end_error": (C.c_int, [_vp] + [C.c_int] * 4 + [C.c_double, C.c_double,
                                                            _dp]),
    "esq_rkc_eval_rhs": (C.c_int, [_vp, C.c_int, C.c_double, C.c_int]),
    "esq_vec_sumsq": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "esq_vec_axpbmc": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]),
    "esq_vec_wdiff_sumsq": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp]),
    "esq_aux_rows": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "esq_vec_wdot": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_double, _dp]),
    "esq_vec_fill": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double]),
    "esq_vec_copy": (C.c_int, [_vp, C.c_int, C.c_int]),
    "esq_vec_eval_rhs": (C.c_int, [_vp, C.c_int, C.c_double, C.c_int]),
    "esq_vec_upload": (C.c_int, [_vp, C.c_int, _vp]),
    "esq_vec_download": (C.c_int, [_vp, C.c_int, _vp]),
    "esq_hs_log_etol": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "esq_hs_select": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double]),
    "esq_set_comm": (C.c_int, [_vp, _vp]),
    "esq_comm_unique_id": (C.c_int, [_vp]),
    "esq_comm_init_rank": (C.c_int, [_vpp, C.c_int, _vp, C.c_int, C.c_int]),
    "esq_comm_destroy": (C.c_int, [_vp]),
    "esq_comm_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "esq_comm_abort": (C.c_int, [_vp]),
    "esq_comm_is_aborted": (C.c_int, [_vp]),
    "esq_allreduce_scalars": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "esq_rhs_diag_create": (C.c_int, [_vpp, C.c_int, _vp, C.c_size_t, C.c_double]),
    "esq_rhs_cdiag_create": (C.c_int, [_vpp, C.c_int, _vp, C.c_size_t, C.c_double,
                                       C.c_double]),
    "esq_rhs_heat2d_create": (C.c_int, [_vpp, C.c_int]),
    "esq_rhs_bruss2d_create": (C.c_int, [_vpp, C.c_int, C.c_double, C.c_double,
                                         C.c_double]),
    "esq_rhs_diff3d_create": (C.c_int, [_vpp, C.c_int]),
    "esq_rhs_free": (C.c_int, [_vp]),
    "esq_rhs_diag": (C.c_int, [_vp, C.c_double, _vp, _vp, C.c_size_t, _vp]),
    "esq_rhs_cdiag": (C.c_int, [_vp, C.c_double, _vp, _vp, C.c_size_t, _vp]),
    "esq_rhs_heat2d": (C.c_int, [_vp, C.c_double, _vp, _vp, C.c_size_t, _vp]),
    "esq_rhs_bruss2d": (C.c_int, [_vp, C.c_double, _vp, _vp, C.c_size_t, _vp]),
    "esq_rhs_diff3d": (C.c_int, [_vp, C.c_double, _vp, _vp, C.c_size_t, _vp]),
    "esq_rhs_heat2d_rkc": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, _vp] + [C.c_double] * 5 + [_vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_diff3d_rkc": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, _vp] + [C.c_double] * 5 + [_vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_heat2d_rkc_chain": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_diff3d_rkc_chain": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_bruss2d_fused": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_heat2d_fused": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_bruss2d_chain": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_heat2d_chain": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_diff3d_chain": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_diff3d_fused": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_diag_fused": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_rhs_cdiag_fused": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "esq_profile_enable": (C.c_int, [_vp, C.c_int]),
    "esq_profile_sampling": (C.c_int, [_vp, C.c_int]),
    "esq_profile_read": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(C.c_long), _dp]),
    "esq_profile_read_moved": (C.c_int, [_vp, C.c_int, _dp]),
    "esq_profile_kernels": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "esq_profile_reset": (C.c_int, [_vp]),
}

_lib = None


class DeviceError(RuntimeError):
    """A call into libextensisq_amd.so failed (HIP/RCCL error or misuse)."""


def load():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeviceError(
            f"{LIB_PATH} not found: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C extensisq_amd/csrc`.  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    ver = lib.esq_abi_version()
    if ver != ABI_VERSION:
        raise DeviceError(f"{LIB_PATH}: ABI version {ver}, this package binds {ABI_VERSION}")
    _lib = lib
    return lib


class Options:
    """The tuning switches of ONE solver (`esq_options=` of a constructor, and so of
    `solve_ivp(..., method=Pr8, esq_options={"chain_depth": 1})`): the `ESQ_*` switches of
    DESIGN.md §3.4, lower case without the prefix.  Nothing here writes the process
    environment (until round 5 the keyword did, for the duration of the constructor):
    the library's own switches travel as an options string to `esq_create2` /
    `esq_rhs_set_options`, the package's are looked up with `get`.  A key that is
    neither is refused.  A switch the caller does not give takes the process default
    `ESQ_<KEY>` -- read, never written."""

    # switches this package reads itself (the library does not know them)
    PACKAGE_KEYS = frozenset((
        "chain", "fuse", "rkc_chain", "rkc_maxdepth", "host_slab", "lazy_y", "prelaunch",
        "launch_ahead", "pre_whole", "chain_errnorm", "chain_pre", "lockstep_debug"))

    def __init__(self, options=None):
        if isinstance(options, Options):
            options = options.values
        self.values = {}
        for key, value in dict(options or {}).items():
            if not isinstance(key, str) or not key.replace("_", "").isalnum():
                raise ValueError(f"esq_options: bad key {key!r}")
            k = key.lower()
            k = k[4:] if k.startswith("esq_") else k
            if isinstance(value, bool):
                value = int(value)
            self.values[k] = str(value)
        lib = None
        self._level = {}
        for k in self.values:
            if k in self.PACKAGE_KEYS:
                self._level[k] = 0
                continue
            lib = lib or load()
            level = lib.esq_option_level(k.encode())
            if level == 0:
                raise ValueError(f"esq_options: unknown switch {k!r}")
            self._level[k] = level

    def get(self, key, default=None):
        """a switch of this package: the explicit value, else ESQ_<KEY>, else default"""
        if key in self.values:
            return self.values[key]
        return os.environ.get("ESQ_" + key.upper(), default)

    def _string(self, level):
        return ";".join(f"{k}={v}" for k, v in sorted(self.values.items())
                        if self._level[k] == level).encode()

    @property
    def context_string(self):
        """the switches of an `esq_ctx` (esq_create2)"""
        return self._string(1)

    @property
    def plugin_string(self):
        """the switches of a built-in plugin object (esq_rhs_set_options)"""
        return self._string(2)


def check(code, ctx=None, what=""):
    if code == 0:
        return
    msg = ""
    if ctx:
        raw = load().esq_last_error(ctx)
        msg = raw.decode(errors="replace") if raw else ""
    raise DeviceError(f"{what or 'libextensisq_amd'} failed with code {code}"
                      + (f": {msg}" if msg else ""))


def device_count():
    """GPUs visible to this process"""
    out = C.c_int(0)
    check(load().esq_device_count(C.byref(out)), None, "esq_device_count")
    return out.value


def device_pci_bus_id(device):
    """PCI address of HIP device `device`, lower case ("0000:c1:00.0"); None if the
    runtime cannot tell"""
    buf = C.create_string_buffer(64)
    if load().esq_device_pci_bus_id(int(device), buf, len(buf)) != 0:
        return None
    return buf.value.decode().strip().lower() or None


def release_cached_memory():
    """free the device memory the library keeps from destroyed solvers for the next ones
    (esq_release_cached_memory); -> bytes freed"""
    held = C.c_size_t(0)
    check(load().esq_release_cached_memory(C.byref(held)), None, "esq_release_cached_memory")
    return held.value


def copy_lane_info(device=0):
    """the record of the large downloads of `device` (esq_copy_lane_info): the fastest and
    the latest one (GB/s), and how many the DMA engines have made"""
    best, last, count = C.c_double(0.0), C.c_double(0.0), C.c_long(0)
    if load().esq_copy_lane_info(int(device), C.byref(best), C.byref(last),
                                 C.byref(count)) != 0:
        return None
    return {"best_gbs": best.value, "last_gbs": last.value, "engine_copies": count.value}


def as_ptr(arr):
    """borrowed pointer to a C-contiguous float64 / complex128 ndarray"""
    return arr.ctypes.data_as(C.c_void_p)


def contiguous(x, dtype):
    return np.ascontiguousarray(x, dtype=dtype)
