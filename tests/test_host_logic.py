"""CPU tests of the product's HOST logic (no GPU, no compute calls): the
step-size controller, the look-ahead rule, tolerance validation, the starting
step, the Chebyshev scalar recurrences -- each against the oracle, which is
itself pinned to the reference's golden vectors (tests/test_oracle_golden.py)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from extensisq_amd import common as dev_common
from extensisq_amd import sommeijer as dev_rkc
from extensisq_amd.bogacki import BS5
from extensisq_amd.prince import Pr7, Pr8, Pr9
from extensisq_amd.tsitouras import Ts5
from oracle import problems as pb
from oracle import rk_oracle, rkc_oracle

from extensisq_amd.calvo import CFMR7osc
from extensisq_amd.cash import CK5
from extensisq_amd.merson import Me4

PAIRS = [(BS5, rk_oracle.BS5), (Ts5, rk_oracle.Ts5), (Pr7, rk_oracle.Pr7),
         (Pr8, rk_oracle.Pr8), (Pr9, rk_oracle.Pr9), (CK5, rk_oracle.CK5),
         (Me4, rk_oracle.Me4), (CFMR7osc, rk_oracle.CFMR7osc)]


def bare(cls, sc_params=None):
    """a product solver object WITHOUT device state: only the scalar fields
    the controller reads"""
    s = object.__new__(cls)
    s._dev = None
    s._y_host = np.zeros(3)
    s.error_exponent = -1 / (min(cls.order_secondary, cls.order) + 1)
    s.h_min_a, s.h_min_b = s._init_min_step_parameters()
    s.tiny_err = s.h_min_b
    s._init_sc_control(sc_params)
    return s


@pytest.mark.parametrize("dev_cls,ref_cls", PAIRS)
@pytest.mark.parametrize("sc", [None, "G", "S", "standard", (0.5, -0.1, 0.1, 0.8)])
def test_controller_matches_oracle(dev_cls, ref_cls, sc):
    ref = ref_cls(lambda t, y: -y, 0.0, np.ones(3), 10.0, sc_params=sc,
                  first_step=0.1)
    dev = bare(dev_cls, sc)
    for name in ("minbeta1", "minbeta2", "minalpha", "safety", "safety_sc",
                 "h_min_a", "h_min_b", "tiny_err", "error_exponent"):
        assert getattr(dev, name) == getattr(ref, name), name
    rng = np.random.default_rng(0)
    dev.h_previous = ref.h_previous = 0.07
    dev.error_norm_old = ref.error_norm_old = 0.3
    for k in range(200):
        err = float(10 ** rng.uniform(-170, 0)) if k % 7 else 1e-200
        h = float(rng.uniform(0.01, 0.2))
        rejected = bool(rng.integers(2))
        f_dev = dev._accept_factor(err, h, rejected)
        f_ref = ref._growth_after_accept(err, h, rejected)
        assert f_dev == f_ref
        assert dev.standard_sc == ref.standard_sc
        assert dev.max_factor == ref.max_factor
        big = float(10 ** rng.uniform(0, 6))
        assert dev._reject_factor(big) == max(
            ref.min_factor, ref.safety * big ** ref.error_exponent)
        dev.error_norm_old = ref.error_norm_old = err
        dev.h_previous = ref.h_previous = h


@pytest.mark.parametrize("dev_cls,ref_cls", PAIRS[:2])
def test_step_limits_match_oracle(dev_cls, ref_cls):
    rng = np.random.default_rng(1)
    for direction, t_bound in ((1, 3.0), (-1, -3.0)):
        ref = ref_cls(lambda t, y: -y, 0.0, np.ones(3), t_bound, first_step=0.1,
                      max_step=0.8)
        dev = bare(dev_cls)
        dev.max_step, dev.t_bound = ref.max_step, ref.t_bound
        for _ in range(300):
            t = direction * float(rng.uniform(0, 3.0))
            dev.h_abs = ref.h_abs = float(10 ** rng.uniform(-20, 1))
            dev.standard_sc = ref.standard_sc = bool(rng.integers(2))
            assert dev._reassess_stepsize(t) == ref._limit_step(t)
            assert dev.standard_sc == ref.standard_sc


def test_validate_tol_matches_oracle():
    y = np.ones(4)
    for rtol, atol in ((1e-3, 1e-6), (1e-20, 0.0), (0.5, np.array([1e-3, 0, 1e-9, 1])),
                       (1e-10, 1e-300)):
        a = dev_common.validate_tol(rtol, atol, y)
        b = rk_oracle.check_tolerances(rtol, atol, y)
        assert a[0] == b[0]
        assert_allclose(a[1], b[1], rtol=0)
    for bad in ((-1.0, 1e-6), (1e-3, -1.0), (1, 1e-6), (1e-3, np.ones(3))):
        with pytest.raises(ValueError):
            dev_common.validate_tol(bad[0], bad[1], y)


def test_h_start_golden(golden_dir):
    """the oracle's starting step (the one tests/test_gpu_parity.py::test_device_h_start
    holds the device procedure to) against the first step of the reference's
    8-shard lock-step run (tools/gen_golden.py)"""
    import os
    g = np.load(os.path.join(golden_dir, "lockstep.npz"))
    N = int(g["N"])
    n = N * N
    f1 = pb.heat2d_rhs(N)
    y0 = np.concatenate([pb.heat2d_y0(N, seed=int(s)) for s in g["seeds"]])

    def fun(t, y):
        return np.concatenate([f1(t, y[k * n:(k + 1) * n]) for k in range(8)])
    rtol, atol = dev_common.validate_tol(1e-6, 1e-9, y0)
    h = rk_oracle.first_step_size(fun, 0.0, float(g["t_end"]), y0, fun(0.0, y0),
                                  Pr9.order_secondary, rtol, atol)
    assert_allclose(abs(h), float(g["h0"]), rtol=1e-12)


@pytest.mark.parametrize("m", [2, 3, 7, 50, 132, 600])
def test_chebyshev_scalars_match_oracle(m):
    t, h = 0.3, 0.0123
    hmus1, table = dev_rkc.chebyshev_scalars(m, t, h)
    mus1, rows = rkc_oracle.chebyshev_stage_scalars(m)
    assert hmus1 == h * mus1
    assert table.shape == (m - 1, 5)
    for got, (mu, nu, mus, ajm1, thjm1) in zip(table, rows):
        assert tuple(got) == (mu, nu, h * mus, ajm1, t + h * thjm1)


def test_dense_output_classes_match_oracle():
    rng = np.random.default_rng(3)
    Q = rng.standard_normal((5, 4))
    y_old = rng.standard_normal(5)
    a = dev_common.HornerDenseOutput(1.0, 1.5, y_old, Q.copy())
    b = rk_oracle.HornerInterpolant(1.0, 1.5, y_old, Q.copy())
    tt = np.linspace(1.0, 1.5, 9)
    assert_allclose(a(tt), b(tt), rtol=1e-15)
    assert_allclose(a(1.25), b(1.25), rtol=1e-15)
    y, f_old, f = rng.standard_normal((3, 5))
    c = dev_common.CubicDenseOutput(1.0, 1.5, y_old, y, f_old, f)
    d = rk_oracle.HermiteInterpolant(1.0, 1.5, y_old, y, f_old, f)
    assert_allclose(c(tt), d(tt), rtol=1e-14)


def test_rms_from_sumsq_lockstep_bookkeeping():
    s = bare(Pr9)
    s._n_norm = 8 * 100
    assert s._rms_from_sumsq(800.0 * 4.0) == 2.0


def test_an_aborted_communicator_is_dropped_by_every_attached_context():
    """when the library has aborted (and freed) the communicator on ONE
    context, LockstepGroup.sync_aborted must clear the pointer on every other
    context attached to it -- their next reduction would otherwise call
    ncclAllReduce on a dead handle"""
    from extensisq_amd.common import LockstepGroup

    class FakeLib:
        def __init__(self):
            self.aborted, self.cleared = set(), []

        def esq_comm_is_aborted(self, handle):
            return 1 if handle in self.aborted else 0

        def esq_set_comm(self, handle, comm):
            assert comm is None
            self.cleared.append(handle)
            return 0

    class FakeDev:
        def __init__(self, lib, handle):
            self.lib, self.handle = lib, handle

    lib = FakeLib()
    grp = LockstepGroup(comm=0xdead, n_total=30)
    devs = [FakeDev(lib, h) for h in (11, 12, 13)]
    for d in devs:
        grp.attach(d)
    assert grp.sync_aborted() is False and lib.cleared == []
    lib.aborted.add(12)
    devs[2].handle = None                     # a closed context is left alone
    assert grp.sync_aborted() is True
    assert lib.cleared == [11] and grp.comm is None
    assert grp.sync_aborted() is True and lib.cleared == [11]


def test_esq_options_travel_as_arguments_not_through_the_environment(monkeypatch):
    """`esq_options=` of a solver constructor (`_lib.Options`): this solver's switches.
    The library's travel as strings to esq_create2 (a context's) / esq_rhs_set_options (a
    plugin object's), the package's are looked up; a switch that was not given takes the
    process default ESQ_<KEY> -- READ, never written: the environment is untouched, so
    two threads may construct solvers with different switches.  A key that is not a
    switch is refused (until round 5 it was silently put into the environment)."""
    import os
    from extensisq_amd._lib import Options
    monkeypatch.setenv("ESQ_CHAIN_DEPTH", "3")
    monkeypatch.setenv("ESQ_LAZY_Y", "0")
    monkeypatch.delenv("ESQ_LAUNCH_AHEAD", raising=False)
    before = dict(os.environ)
    seen = {}

    class Probe:
        @dev_common._with_esq_options
        def __init__(self, a, b=2):
            seen.update(opts=self._esq_options, a=a, b=b)

    Probe(1, b=5, esq_options={"chain_depth": 1, "ESQ_LAZY_ROWS": False, "lazy_y": "always",
                               "launch_ahead": False, "chain_rows": 12, "rkc_force": True})
    o = seen["opts"]
    assert (seen["a"], seen["b"]) == (1, 5)
    assert dict(os.environ) == before                       # nothing written
    # library switches by the object they steer, package switches by lookup
    assert o.context_string == b"chain_depth=1;lazy_rows=0"
    assert o.plugin_string == b"chain_rows=12;rkc_force=1"
    assert o.get("lazy_y") == "always" and o.get("launch_ahead") == "0"
    assert o.get("pre_whole", "1") == "1"                   # not given, not in the environment
    Probe(7)                                                # no options: the process defaults
    o = seen["opts"]
    assert o.context_string == b"" and o.plugin_string == b"" and seen["a"] == 7
    assert o.get("lazy_y", "1") == "0"                      # ESQ_LAZY_Y, read
    for bad in ({"chain depth; rm": 1}, {"chian_depth": 1}, {"d2h_mode": "auto"},
                {"plan_greedy": 1}, {3: 1}):
        with pytest.raises(ValueError):
            Probe(1, esq_options=bad)
    assert isinstance(Options(o), Options) and Options(o).values == o.values


def test_option_keys_are_known_to_the_library_by_level():
    """esq_option_level: 1 = a context's switch, 2 = a plugin object's, 0 = unknown;
    keys in any case, with or without the prefix"""
    from extensisq_amd import _lib
    lib = _lib.load()
    for key in (b"chain_depth", b"CHAIN_DEPTH", b"ESQ_CHAIN_DEPTH", b"lazy_rows", b"lazy_end",
                b"src", b"block_acc", b"rkc_depth", b"rkc_first", b"rkc_last", b"epi_nt",
                b"comm_timeout_s", b"chain_from_rows"):
        assert lib.esq_option_level(key) == 1, key
    for key in (b"chain_rows", b"rkc_force", b"rkc_planes", b"diff3d_r"):
        assert lib.esq_option_level(key) == 2, key
    # closed experiments (retired in round 6) and nonsense
    for key in (b"d2h_mode", b"plan_greedy", b"chain_split", b"row_stagger", b"block_fold",
                b"chain_caps", b"rhs_variant", b"plan_write_cost", b"", b"chain"):
        assert lib.esq_option_level(key) == 0, key
    # a plugin object takes its own switches only (-1: ESQ_EINVAL)
    import ctypes
    user = ctypes.c_void_p()
    assert lib.esq_rhs_heat2d_create(ctypes.byref(user), 64) == 0
    assert lib.esq_rhs_set_options(user, b"chain_rows=12;rkc_planes=3") == 0
    assert lib.esq_rhs_set_options(user, b"chain_depth=1") == -1
    assert lib.esq_rhs_set_options(user, b"no_such=1") == -1
    assert lib.esq_rhs_set_options(user, None) == 0
    assert lib.esq_rhs_free(user) == 0
    # ... and so does a context: refused before anything touches a device (no GPU here),
    # the offending key in the message
    ctx = ctypes.c_void_p()
    for bad in (b"chain_rows=12", b"no_such=1", b"chain_depth=1;d2h_mode=auto"):
        assert lib.esq_create2(ctypes.byref(ctx), 0, 100, 3, 0, 0, bad) == -1
        msg = lib.esq_last_error(ctx).decode()
        assert "not a context option" in msg and bad.split(b"=")[-2].split(b";")[-1].decode().upper() in msg, msg
        lib.esq_destroy(ctx)


def test_chebyshev_scalars_cached_base_is_bit_identical():
    """sommeijer.chebyshev_scalars keeps the m-dependent part of the coefficients of the
    last stage counts; the table of a step must equal the scalar loop's (ref
    sommeijer.py:278-314: h*mus and t + h*theta formed per stage) bit for bit"""
    from math import cosh, log, sinh, sqrt

    from extensisq_amd.sommeijer import chebyshev_scalars

    def scalar_loop(m, t, h):
        w0 = 1.0 + 2.0 / (13.0 * m ** 2)
        sq = w0 ** 2 - 1.0
        rt = sqrt(sq)
        arg = m * log(w0 + rt)
        w1 = sinh(arg) * sq / (cosh(arg) * m * rt - w0 * sinh(arg))
        bj1 = bj2 = 1.0 / (2.0 * w0) ** 2
        mus1 = w1 * bj1
        thj2, thj1 = 0.0, mus1
        zj1, zj2, dzj1, dzj2, d2zj1, d2zj2 = w0, 1.0, 1.0, 0.0, 0.0, 0.0
        rows = []
        for _ in range(2, m + 1):
            zj = 2.0 * w0 * zj1 - zj2
            dzj = 2.0 * w0 * dzj1 - dzj2 + 2.0 * zj1
            d2zj = 2.0 * w0 * d2zj1 - d2zj2 + 4.0 * dzj1
            bj = d2zj / dzj ** 2
            ajm1 = 1.0 - zj1 * bj1
            mu = 2.0 * w0 * bj / bj1
            nu = -bj / bj2
            mus = mu * w1 / w0
            rows.append((mu, nu, h * mus, ajm1, t + h * thj1))
            thj = mu * thj1 + nu * thj2 + mus * (1.0 - ajm1)
            thj2, thj1, bj2, bj1 = thj1, thj, bj1, bj
            zj2, zj1, dzj2, dzj1, d2zj2, d2zj1 = zj1, zj, dzj1, dzj, d2zj1, d2zj
        return h * mus1, np.array(rows).reshape(-1, 5)

    rng = np.random.default_rng(5)
    for m in (1, 2, 3, 17, 99, 100, 431):
        for _ in range(3):                       # (second and third call: from the cache)
            t, h = rng.normal() * 10.0, rng.normal() * 10.0 ** rng.uniform(-6, 1)
            want, got = scalar_loop(m, t, h), chebyshev_scalars(m, t, h)
            assert want[0] == got[0]
            assert want[1].shape == got[1].shape
            assert np.array_equal(want[1].view(np.uint64), got[1].view(np.uint64)), (m, t, h)
