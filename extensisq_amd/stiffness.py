"""Stiffness diagnosis for the explicit RK solvers (RKSuite's STIFF, after
L.F. Shampine, "Diagnosing Stiffness for Runge-Kutta Methods", SIAM J. Sci.
Stat. Comput. 12 (1991) 260-272), device-resident.

Reference counterpart: `_diagnose_stiffness` + `stiff_a/b/c/d`
(extensisq/common.py:370-516, 824-1204).  A nonlinear power iteration on
finite-difference Jacobian-vector products estimates the two dominant
eigenvalues of havg*J; all vectors (three iterates, the probe point) live in
auxiliary HBM rows, every inner product is RKSuite's weighted one
(`esq_vec_wdot`), and only scalars return to the host, where the 2x2
least-squares fits and the decision logic run.
"""
import ctypes
import logging
from math import sqrt
from warnings import warn

import numpy as np

from ._lib import VEC_NONE, VEC_WORK, VEC_Y, VEC_YNEW, as_ptr

_LARGE = 1.0e10
_MAX_TRIES = 8


def _real_root_test(new_new, old_new, old_old, rayleigh_prev):
    """Has the iteration collapsed onto one dominant REAL eigenvalue?
    (ref stiff_b, common.py:1105-1137).  Returns
    (rayleigh, rho, root1, root2, is_real)."""
    r = old_new / old_old
    rho = abs(r)
    det = old_old * new_new - old_new ** 2
    res = abs(det / old_old)
    is_real = det == 0.0 or (res <= 1e-6 * new_new
                             and abs(r - rayleigh_prev) <= 0.001 * rho)
    return r, rho, [r if is_real else 0.0, 0.0], [0.0, 0.0], is_real


def _quadratic_roots(alpha, beta):
    """roots of x^2 + alpha x + beta as [re, im] pairs, |r1| >= |r2|
    (ref stiff_c, common.py:1140-1175)"""
    half = alpha / 2
    disc = half ** 2 - beta
    if disc == 0.0:
        return [-half, 0.0], [-half, 0.0]
    root = sqrt(abs(disc))
    if disc < 0.0:
        return [-half, root], [-half, -root]
    big = -half - root if half > 0.0 else -half + root
    return [big, 0.0], [beta / big, 0.0]


class _Workspace:
    """device vectors and primitives of one diagnosis"""

    def __init__(self, solver):
        self.s = solver
        self.lib, self.ctx = solver._lib, solver._ctx
        if getattr(solver, "_stiff_rows", None) is None:
            first = ctypes.c_int()
            solver._chk(self.lib.esq_aux_rows(self.ctx, 5, ctypes.byref(first)),
                        "esq_aux_rows")
            solver._stiff_rows = first.value
        base = solver._stiff_rows
        self.v = [base, base + 1, base + 2, base + 3]
        self.probe = base + 4
        rid = self.lib.esq_rk_row_id(self.ctx, 0, 0)          # f(t, y)
        if rid < 0:      # an error code, not a vector id (ids of rows are >= 0; the
            solver._chk(rid, "esq_rk_row_id")    # negative ids name the fixed slots)
        self.f_row = rid
        self.floor = sqrt(np.finfo(np.float64).tiny)

    def dot(self, a, b):
        out = ctypes.c_double()
        self.s._chk(self.lib.esq_vec_wdot(self.ctx, a, b, VEC_Y, VEC_YNEW,
                                          self.floor, ctypes.byref(out)),
                    "esq_vec_wdot")
        # weighted inner products of the WHOLE batch in a lock-step group
        return self.s._group_reduce([out.value], "sum")[0]

    def axpbmc(self, dst, a, alpha, b, c=VEC_NONE):
        self.s._chk(self.lib.esq_vec_axpbmc(self.ctx, dst, a, float(alpha), b, c),
                    "esq_vec_axpbmc")

    def rhs(self, dst, t, src):
        s = self.s
        if s._device_rhs is not None:
            s._chk(self.lib.esq_vec_eval_rhs(self.ctx, dst, float(t), src),
                   "esq_vec_eval_rhs")
            s.nfev += 1
            return
        arg = np.empty(s.n, dtype=s._dev.dtype)
        s._chk(self.lib.esq_vec_download(self.ctx, src, as_ptr(arg)),
               "esq_vec_download")
        val = np.ascontiguousarray(s.fun(t, arg), dtype=s._dev.dtype)
        s._chk(self.lib.esq_vec_upload(self.ctx, dst, as_ptr(val)),
               "esq_vec_upload")

    def jac_times(self, dst, v, vv, t, havg, scale):
        """dst = havg * J v by a one-sided difference (ref stiff_d,
        common.py:1178-1204); returns <dst, dst>"""
        eps = scale / sqrt(vv)
        self.axpbmc(self.probe, VEC_Y, eps, v)
        self.rhs(dst, t, self.probe)
        self.axpbmc(dst, VEC_NONE, havg / eps, dst, self.f_row)
        return self.dot(dst, dst)


def dominant_roots(solver, hnow, havg):
    """(stif, rootre, roots) like the reference's stiff_a (common.py:824-1103);
    `roots` = (root1, root2, rho) when the power iteration converged."""
    t, xend = solver.t, solver.t_bound
    if abs(hnow / havg) > 5 or abs(hnow / havg) < 0.2:
        return False, None, None
    if solver.n_stages * abs((xend - t) / havg) <= solver.nfev_stiff_detect:
        return False, None, None

    w = _Workspace(solver)
    lib, ctx = w.lib, w.ctx
    epsneg = np.finfo(np.float64).epsneg
    v0, v1, v2, v3 = w.v
    # start vector: the embedded error estimate of the step just taken
    solver._chk(lib.esq_rk_error_vector(ctx, float(solver.h_previous), 1),
                "esq_rk_error_vector")
    solver._chk(lib.esq_vec_copy(ctx, v0, VEC_WORK), "esq_vec_copy")

    scale = sqrt(w.dot(VEC_Y, VEC_Y)) * sqrt(epsneg)
    if scale == 0.0:
        scale = sqrt(w.dot(v0, v0)) * sqrt(epsneg)
        if scale == 0.0:
            return None, None, None
    vv0 = w.dot(v0, v0)
    if vv0 == 0.0:
        solver._chk(lib.esq_vec_fill(ctx, v0, 1.0, 1.0), "esq_vec_fill")
        vv0 = w.dot(v0, v0)
    w.axpbmc(v0, VEC_NONE, 1.0 / sqrt(vv0), v0)
    vv0 = 1.0

    rayleigh = None
    root1 = root2 = rho = None
    for attempt in range(_MAX_TRIES):
        vv1 = w.jac_times(v1, v0, vv0, t, havg, scale)
        if sqrt(vv1) > _LARGE * sqrt(vv0):
            return None, None, None
        v0v1 = w.dot(v0, v1)
        if attempt == 0:
            rayleigh = v0v1 / vv0
            if abs(rayleigh) < epsneg ** (1 / 3):
                return False, None, None
        else:
            rayleigh, rho, root1, root2, real = _real_root_test(
                vv1, v0v1, vv0, rayleigh)
            if real:
                return None, True, (root1, root2, rho)
        vv2 = w.jac_times(v2, v1, vv1, t, havg, scale)
        v0v2 = w.dot(v0, v2)
        v1v2 = w.dot(v1, v2)
        rayleigh, rho, root1, root2, real = _real_root_test(vv2, v1v2, vv1,
                                                            rayleigh)
        if real:
            return None, True, (root1, root2, rho)
        # quadratic fitted to (v0, v1, v2), then to (v1, v2, v3)
        det1 = vv0 * vv1 - v0v1 ** 2
        alpha1 = (-vv0 * v1v2 + v0v1 * v0v2) / det1
        beta1 = (v0v1 * v1v2 - vv1 * v0v2) / det1
        vv3 = w.jac_times(v3, v2, vv2, t, havg, scale)
        v1v3 = w.dot(v1, v3)
        v2v3 = w.dot(v2, v3)
        rayleigh, rho, root1, root2, real = _real_root_test(vv3, v2v3, vv2,
                                                            rayleigh)
        if real:
            return None, True, (root1, root2, rho)
        det2 = vv1 * vv2 - v1v2 ** 2
        alpha2 = (-vv1 * v2v3 + v1v2 * v1v3) / det2
        beta2 = (v1v2 * v2v3 - vv2 * v1v3) / det2
        res2 = abs(vv3 + vv2 * alpha2 ** 2 + vv1 * beta2 ** 2
                   + 2 * v2v3 * alpha2 + 2 * v1v3 * beta2
                   + 2 * v1v2 * alpha2 * beta2)
        if res2 <= 1e-6 * vv3:
            r1, r2 = _quadratic_roots(alpha1, beta1)
            root1, root2 = _quadratic_roots(alpha2, beta2)
            rho = sqrt(root1[0] ** 2 + root1[1] ** 2)
            d1 = (root1[0] - r1[0]) ** 2 + (root1[1] - r1[1]) ** 2
            d2 = (root1[0] - r2[0]) ** 2 + (root1[1] - r2[1]) ** 2
            if sqrt(min(d1, d2)) <= 0.001 * rho:
                return None, False, (root1, root2, rho)
        w.axpbmc(v0, VEC_NONE, 1.0 / sqrt(vv3), v3)
        vv0 = 1.0
    return None, None, None


def diagnose(solver, lotsfl):
    """decide and report (ref common.py:410-516); returns (stif, rootre, roots)
    for the tests"""
    stif, rootre, roots = dominant_roots(solver, solver.h_previous, solver.havg)
    if roots is not None:
        root1, root2, rho = roots
        rootre = root1[1] == 0.0
        if root1[0] > 0.0:
            stif = False
        else:
            rho2 = sqrt(root2[0] ** 2 + root2[1] ** 2)
            if rho2 >= 0.9 * rho and root2[0] > 0.0:
                stif = False
            elif abs(root1[1]) > abs(root1[0]) * solver.tanang:
                stif = None
            else:
                stif = rho >= 0.9 * solver.stbrad
    solver._last_stiffness = (stif, rootre, roots)
    if stif is None:
        if rootre is None:
            logging.info('Stiffness detection did not converge')
        if not rootre:
            if lotsfl:
                warn('Your problem has a complex pair of dominant roots near '
                     'the imaginary axis.  There are many recently failed '
                     'steps.  You should probably change to a code intended '
                     'for oscillatory problems.')
            else:
                logging.info('The problem has a complex pair of dominant roots '
                             'near the imaginary axis.  There are not many '
                             'failed steps.')
        else:
            logging.warning('stif=None, rootre=True; this should not happen')
    elif stif:
        if rootre is None:
            logging.warning('stif=True, rootre=None; this should not happen')
        elif rootre:
            warn('Your problem has a real dominant root and is diagnosed as '
                 'stiff.  You should probably change to a code intended for '
                 'stiff problems.')
        else:
            warn('Your problem has a complex pair of dominant roots and is '
                 'diagnosed as stiff.  You should probably change to a code '
                 'intended for stiff problems.')
    else:
        if rootre is None:
            logging.info('Stiffness detection has diagnosed the problem as '
                         'non-stiff, without performing power iterations')
        elif rootre:
            logging.info('The problem has a real dominant root and is not stiff')
        else:
            logging.info('The problem has a complex pair of dominant roots and '
                         'is not stiff')
    return stif, rootre, roots
