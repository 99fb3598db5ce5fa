"""GPU tests of the scipy plugin surface, re-expressing the hot-path cases of
the reference's tests/test_ivp.py: events (:369-541), t_eval (:668-782),
args (:988-1078), dense output after a terminal event -- each run through
`solve_ivp` with the device classes (host-RHS mode: Python callables) and
compared with the oracle's run of the same call."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_equal
from scipy.integrate import solve_ivp

import extensisq_amd as esq
from oracle import problems as pb
from oracle import rk_oracle

pytestmark = pytest.mark.gpu

NAMES = ["BS5", "Ts5", "Pr7", "Pr8", "Pr9", "CK5", "Me4", "CFMR7osc", "CKdisc"]


def ev1(t, y):
    return y[0] - y[1] ** 0.7


def ev2(t, y):
    return y[1] ** 0.6 - y[0]


def ev3(t, y):
    return t - 7.4


def both(name, *args, **kw):
    got = solve_ivp(*args, method=getattr(esq, name), **kw)
    ref = solve_ivp(*args, method=rk_oracle.METHODS[name], **kw)
    return got, ref


def same_events(got, ref):
    assert got.status == ref.status
    assert len(got.t_events) == len(ref.t_events)
    for a, b, ya, yb in zip(got.t_events, ref.t_events, got.y_events,
                            ref.y_events):
        assert a.shape == b.shape and np.shape(ya) == np.shape(yb)
        assert_allclose(a, b, rtol=1e-7)
        if a.size:
            assert_allclose(ya, yb, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", NAMES)
def test_events(name):
    y0 = [1 / 3, 2 / 9]
    for d1, d2 in ((0, 0), (1, 1), (-1, -1)):
        ev1.direction, ev2.direction = d1, d2
        got, ref = both(name, pb.rational_rhs, [5, 8], y0, events=(ev1, ev2))
        same_events(got, ref)
    ev1.direction = ev2.direction = 0
    ev3.terminal = True
    got, ref = both(name, pb.rational_rhs, [5, 8], y0, events=(ev1, ev2, ev3),
                    dense_output=True)
    same_events(got, ref)
    assert got.status == 1 and 7.3 < got.t_events[2][0] < 7.5
    tc = np.linspace(got.t[0], got.t[-1])
    assert_allclose(got.sol(tc), ref.sol(tc), rtol=1e-6, atol=1e-9)
    # backward
    got, ref = both(name, pb.rational_rhs, [8, 5], [4 / 9, 20 / 81],
                    events=(ev1, ev2))
    same_events(got, ref)


@pytest.mark.parametrize("name", NAMES)
def test_t_eval(name):
    y0 = [1 / 3, 2 / 9]
    for t_span, t_eval in (([5, 9], np.linspace(5, 9, 7)),
                           ([5, 1], np.linspace(5, 1, 5)),
                           ([5, 9], [5.5, 7.25]), ([5, 9], [5.01, 7, 8.01, 9])):
        got, ref = both(name, pb.rational_rhs, t_span, y0, rtol=1e-3, atol=1e-6,
                        t_eval=t_eval)
        assert_equal(got.t, t_eval)
        assert got.success and got.nfev == ref.nfev
        assert_allclose(got.y, ref.y, rtol=1e-7, atol=1e-10)
    with pytest.raises(ValueError):
        solve_ivp(pb.rational_rhs, [5, 9], y0, method=getattr(esq, name),
                  t_eval=[4, 6])


@pytest.mark.parametrize("name", NAMES)
def test_args_and_tight_tolerance(name):
    """tests/test_ivp.py:988-1078 (the falling-object event problem)"""
    def fun(t, z, omega):
        x, v = z
        return [v, -omega ** 2 * x]

    def hit(t, z, omega):
        return z[0]
    kw = dict(rtol=1e-10, atol=1e-13, args=(2.0,), events=hit)
    got, ref = both(name, fun, [0, 3.0], [1.0, 0.0], **kw)
    assert got.success
    assert_allclose(got.t_events[0], [np.pi / 4, 3 * np.pi / 4], rtol=1e-8)
    assert_allclose(got.t_events[0], ref.t_events[0], rtol=1e-9)
    assert_allclose(got.y[:, -1], ref.y[:, -1], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("name", NAMES)
def test_string_free_drop_in(name):
    """the classes are accepted by solve_ivp's `method=` validation and report
    the OdeSolver bookkeeping scipy expects"""
    res = solve_ivp(lambda t, y: -y, [0, 1], [1.0, 2.0], method=getattr(esq, name))
    assert res.success and res.njev == 0 and res.nlu == 0
    assert_allclose(res.y[:, -1], np.exp(-1) * np.array([1.0, 2.0]), rtol=2e-3)
