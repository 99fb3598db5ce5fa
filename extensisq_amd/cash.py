"""CK5 and CKdisc: the Cash-Karp methods (J.R. Cash, A.H. Karp, ACM TOMS 16
(1990) 201-222) on the device-resident step.

CK5 is the 5(4) pair as data only (reference: extensisq/cash.py:9-112).

CKdisc is the variable-order (5, 3, 2) method for non-smooth problems
(reference: cash.py:115-416): between stage pairs it assesses embedded
solutions of order 2 and 3 to predict whether the fifth-order step will succeed
and, if not, falls back to a lower-order solution over a shortened step without
further RHS evaluations.  Every assessment is one fused HIP pass
(`esq_rk_custom_sol_err`: solution, scale, error, weighted norm); the "quit" /
"twiddle" bookkeeping below is the host part.
"""
import ctypes

import numpy as np

from ._lib import as_ptr
from ._tableau import install
from .common import NFS, CubicDenseOutput, RungeKutta

SAFETY = 0.9


class CK5(RungeKutta):
    pass


class CKdisc(RungeKutta):

    def __init__(self, fun, t0, y0, t_bound, **extraneous):
        super().__init__(fun, t0, y0, t_bound, nfev_stiff_detect=0,
                         **extraneous)
        self.twiddle = [1.5, 1.1]
        self.quit = [100., 100.]
        self.order_accepted = None

    def _pair_norm(self, h, B, E, rows, store):
        """weighted RMS norm of the embedded pair (B, E) over K[:rows]; with
        `store` the solution goes to the YNEW slot (ref `_comp_sol_err_tol`)"""
        b = np.ascontiguousarray(B[:rows], dtype=np.float64)
        e = np.ascontiguousarray(E[:rows], dtype=np.float64)
        out = ctypes.c_double()
        self._chk(self._lib.esq_rk_custom_sol_err(
            self._ctx, float(h), as_ptr(b), as_ptr(e), rows, int(store),
            ctypes.byref(out)), "esq_rk_custom_sol_err")
        return self._rms_from_sumsq(out.value)

    def _step_impl(self):
        """ref cash.py:253-395"""
        t = self.t
        twiddle, quit = self.twiddle, self.quit
        h_abs, min_step = self._reassess_stepsize(t)
        order_accepted = 0
        rejected = False
        while not order_accepted:
            if h_abs < min_step:
                return False, self.TOO_SMALL_STEP
            h = h_abs * self.direction
            self._run_stages(1, 2, t, h)
            E1 = self._pair_norm(h, self.B_assess[0], self.E_assess[0], 2,
                                 False) ** (1 / 2)
            esttol = E1 / quit[0]
            if E1 < twiddle[0] * quit[0]:
                self._run_stages(2, 4, t, h)
                E2 = self._pair_norm(h, self.B_assess[1], self.E_assess[1], 4,
                                     False) ** (1 / 3)
                esttol = E2 / quit[1]
                if E2 < twiddle[1] * quit[1]:
                    self._run_stages(4, 6, t, h)
                    # the tableau's own pair: `_comp_sol_err` of the base class
                    # (fused into the last stage's sweep where the plugin can)
                    E4 = self._solution_and_error(t, h) ** (1 / 5)
                    E4 = E4 or 1e-160
                    esttol = E4
                    if E4 < 1:
                        order_accepted = 4
                        factor = min(self.max_factor, SAFETY / E4)
                        if rejected:
                            factor = min(1.0, factor)
                        h_abs *= factor
                        q = [E1 / E4, E2 / E4]
                        for j in (0, 1):
                            if q[j] > quit[j]:
                                q[j] = min(q[j], 10 * quit[j])
                            else:
                                q[j] = max(q[j], 2 / 3 * quit[j])
                            quit[j] = max(1., min(10000., q[j]))
                        break
                    if np.isnan(E4) or np.isinf(E4):
                        return False, "Overflow or underflow encountered."
                    for i, e in enumerate((E1, E2)):
                        ratio = e / quit[i]
                        if ratio < twiddle[i]:
                            twiddle[i] = max(1.1, ratio)
                    if E2 < 1:
                        if self._pair_norm(h, self.B_fallback[1],
                                           self.E_fallback[1], 4, True) < 1:
                            order_accepted = 2
                            h_abs *= self.C_fallback[1]
                            h = h_abs * self.direction
                            break
                if E1 < 1:
                    if self._pair_norm(h, self.B_fallback[0], self.E_fallback[0],
                                       2, True) < 1:
                        order_accepted = 1
                        h_abs *= self.C_fallback[0]
                        h = h_abs * self.direction
                        break
                    rejected = True
                    h_abs *= self.C_fallback[0]
                    NFS[()] += 1
                    continue
            rejected = True
            h_abs *= max(self.min_factor, SAFETY / esttol)
            NFS[()] += 1
        # the derivative at the accepted point (next first stage, interpolation)
        t_new = t + h
        self._finish_step(t_new, h, h_abs)
        self.order_accepted = order_accepted
        self.h_previous = h
        self.h_abs = h_abs
        self.t = t_new
        return True, None

    def _dense_output_impl(self):
        if self.order_accepted == 4:
            return self._horner_interpolant(self.P, self.t_old, self.t)
        return CubicDenseOutput(self.t_old, self.t, self.y_old, self.y,
                                self.f_old, self.f)


install(CK5, "CK5")
install(CKdisc, "CKdisc")
