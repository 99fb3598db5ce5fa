"""Tableau data checks (CPU): rooted-tree order conditions up to the FULL
nominal order (the reference stops at 7, tests/test_rk.py:14-42), the
coefficient identities of tests/test_rk.py:45-72, and bit-equality of the
product's class attributes with the data file the oracle reads."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from extensisq_amd.bogacki import BS5
from extensisq_amd.prince import Pr7, Pr8, Pr9
from extensisq_amd.tsitouras import Ts5
from extensisq_amd.cash import CK5
from extensisq_amd.merson import Me4
from extensisq_amd.calvo import CFMR7osc
from oracle import rk_oracle
from rooted_trees import max_residual

METHODS = [BS5, Ts5, Pr7, Pr8, Pr9, CK5, Me4, CFMR7osc]


@pytest.mark.parametrize("cls", METHODS)
def test_order_conditions(cls):
    s = cls.n_stages
    tol = s * 1e-14
    for k in range(1, cls.order + 1):
        assert max_residual(k, cls.B, cls.A) < tol
    assert max_residual(cls.order + 1, cls.B, cls.A) > 1e-10   # not higher
    # embedded method b_hat = E + B on the FSAL-extended tableau
    A = np.zeros((s + 1, s + 1))
    A[:s, :s] = cls.A
    A[s, :s] = cls.B
    bh = cls.E.copy()
    bh[:s] += cls.B
    for k in range(1, cls.order_secondary + 1):
        assert max_residual(k, bh, A) < tol
    assert max_residual(cls.order_secondary + 1, bh, A) > 1e-10


@pytest.mark.parametrize("cls", METHODS)
def test_coefficient_identities(cls):
    assert_allclose(np.sum(cls.B), 1, rtol=1e-15)
    assert_allclose(np.sum(cls.E), 0, atol=1e-15)
    assert_allclose(np.sum(cls.A, axis=1), cls.C, rtol=1e-13)
    assert np.all(np.triu(cls.A) == 0)
    assert cls.E.shape == (cls.n_stages + 1,)
    P = cls.P
    Ps = np.sum(P, axis=1)
    Ps[:cls.B.size] -= cls.B
    assert_allclose(Ps, 0, atol=1e-12)             # C0 at the end
    Pc = np.sum(P, axis=0)
    Pc[0] -= 1
    assert_allclose(Pc, 0, atol=1e-12)             # C1 at the start
    dP = (P * (np.arange(P.shape[1]) + 1)).sum(axis=1)
    dP[-1] -= 1
    assert_allclose(dP, 0, atol=2e-12)             # C1 at the end


@pytest.mark.parametrize("cls", METHODS)
def test_matches_oracle_data(cls):
    ref = rk_oracle.METHODS[cls.__name__]
    for name in ("A", "B", "C", "E", "P"):
        assert_array_equal(getattr(cls, name), getattr(ref, name))
    for name in ("n_stages", "order", "order_secondary", "sc_params",
                 "tanang", "stbrad"):
        assert getattr(cls, name) == getattr(ref, name)
    assert (cls.E[cls.n_stages] != 0) == (cls.__name__ in ("BS5", "Ts5"))


def test_published_values_spot_check():
    """a few coefficients typed from the papers, as an independent anchor"""
    assert_array_equal(BS5.C, [0, 1 / 6, 2 / 9, 3 / 7, 2 / 3, 3 / 4, 1])
    assert BS5.B[0] == 587 / 8064 and BS5.A[2, 1] == 4 / 27
    assert_array_equal(Pr7.C, [0, 1 / 6, 1 / 4, 1 / 2, 1 / 2, 3 / 16, 3 / 16,
                               3 / 5, 6 / 7, 1])
    assert Pr7.B[0] == 179 / 3240
    assert_array_equal(Ts5.C, [0, 0.161, 0.327, 0.9, 0.9800255409045097, 1])
    assert Ts5.B[1] == 0.01
    assert_array_equal(Pr8.C, [0, 7 / 75, 7 / 50, 7 / 25, 7 / 25, 19 / 40,
                               19 / 40, 7 / 50, 2 / 25, 8 / 15, 4 / 5, 22 / 25, 1])
    assert Pr9.n_stages == 17 and Pr9.C[-1] == 1
    nnz = [int(np.count_nonzero(c.A)) for c in (Ts5, BS5, Pr7, Pr8, Pr9)]
    assert nnz == [15, 21, 40, 69, 122]
