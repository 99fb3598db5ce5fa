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

    # The ladder: stage pairs (1), (2, 3), (4, 5); after each of the first two an
    # embedded pair of order 2 / 3 is assessed, after the last the 5(4) pair of
    # the tableau itself.  (rows of K involved, root that makes the norm
    # comparable between the orders)
    _LADDER = ((2, 1 / 2), (4, 1 / 3), (6, 1 / 5))
    _ORDER_OF_FALLBACK = (1, 2)        # `order_accepted` of fallback level 0 / 1

    def _climb(self, t, h):
        """Evaluate stage pairs for the step size h while the assessed errors
        promise that the fifth-order step will succeed (gates
        `E < twiddle * quit`, ref cash.py:277-303).  Returns the assessed,
        order-normalised errors [E1, E2, E4] as far as the climb got."""
        est = []
        first = 1
        for level, (rows, root) in enumerate(self._LADDER):
            self._run_stages(first, rows, t, h)
            first = rows
            if level < 2:
                e = self._pair_norm(h, self.B_assess[level], self.E_assess[level],
                                    rows, False) ** root
                est.append(e)
                if not e < self.twiddle[level] * self.quit[level]:
                    break
            else:
                # the tableau's own pair: `_comp_sol_err` of the base class
                # (fused into the last stage's sweep where the plugin can)
                est.append(self._solution_and_error(t, h) ** root or 1e-160)
        return est

    def _tune_quit(self, est):
        """after an accepted fifth-order step (ref cash.py:318-326)"""
        for j in (0, 1):
            q = est[j] / est[2]
            q = min(q, 10 * self.quit[j]) if q > self.quit[j] else \
                max(q, 2 / 3 * self.quit[j])
            self.quit[j] = max(1., min(10000., q))

    def _tune_twiddle(self, est):
        """after a failed fifth-order step (ref cash.py:334-339)"""
        for j in (0, 1):
            ratio = est[j] / self.quit[j]
            if ratio < self.twiddle[j]:
                self.twiddle[j] = max(1.1, ratio)

    def _descend(self, est, h):
        """The climb did not end in an accepted fifth-order step: try the
        lower-order solutions over the shortened step, highest first
        (ref cash.py:341-375).  Level 1 (third order) is only looked at after a
        failed fifth-order attempt, level 0 (second order) whenever the first
        gate was passed.  Returns ("accept", level), ("shrink", 0) -- non-smooth
        behaviour, retry with the shortened step -- or ("reject", None)."""
        depth = len(est)
        for level in (1, 0):
            reached = depth == 3 if level == 1 else depth >= 2
            if not reached or not est[level] < 1:
                continue
            rows = self._LADDER[level][0]
            if self._pair_norm(h, self.B_fallback[level], self.E_fallback[level],
                               rows, True) < 1:
                return "accept", level
            if level == 0:
                return "shrink", 0
        return "reject", None

    def _step_impl(self):
        """variable-order step (ref cash.py:253-395)"""
        t = self.t
        h_abs, min_step = self._reassess_stepsize(t)
        rejected = False
        while True:
            if h_abs < min_step:
                return False, self.TOO_SMALL_STEP
            h = h_abs * self.direction
            est = self._climb(t, h)
            if len(est) == 3:
                if est[2] < 1:                       # fifth order accepted
                    order_accepted = 4
                    factor = min(self.max_factor, SAFETY / est[2])
                    h_abs *= min(1.0, factor) if rejected else factor
                    self._tune_quit(est)
                    break
                if np.isnan(est[2]) or np.isinf(est[2]):
                    return False, "Overflow or underflow encountered."
                self._tune_twiddle(est)
            verdict, level = self._descend(est, h)
            if verdict == "accept":
                # the fallback solution belongs to the SHORTENED step
                order_accepted = self._ORDER_OF_FALLBACK[level]
                h_abs *= self.C_fallback[level]
                h = h_abs * self.direction
                break
            rejected = True
            NFS[()] += 1
            if verdict == "shrink":
                h_abs *= self.C_fallback[0]
            else:
                depth = len(est)
                esttol = est[-1] / self.quit[depth - 1] if depth < 3 else est[-1]
                h_abs *= max(self.min_factor, SAFETY / esttol)
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
