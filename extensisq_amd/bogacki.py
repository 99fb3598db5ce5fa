"""BS5: the Bogacki-Shampine 5(4) pair (Comput. Math. Appl. 32 (1996) 15-28;
coefficients as in RKSUITE), 7 effective stages, FSAL, with TWO error
estimates: an early one after six stages that can reject a step before the
last two RHS evaluations, and the usual one at the end.  Reference counterpart:
extensisq/bogacki.py:103-393.

Device mapping: stages and both error norms are HIP kernels
(`esq_rk_stages`, `esq_rk_pre_error`, `esq_rk_solution_error`); the controller
below is host scalar arithmetic."""
import numpy as np

from ._lib import SLOT_K, SLOT_WORK, SLOT_YNEW, SLOT_YSTAGE, as_ptr  # noqa: F401
from ._tableau import install
from .common import NFS, RungeKutta


class BS5(RungeKutta):
    _extra_rows = 3      # room for the interpolants' extra stages (rows 8..10)

    def __init__(self, fun, t0, y0, t_bound, nfev_stiff_detect=5000,
                 sc_params='standard', interpolant='low', **extraneous):
        if interpolant not in ('best', 'low', 'free'):
            raise ValueError(
                "interpolant should be one of: 'best', 'low', 'free'")
        super().__init__(fun, t0, y0, t_bound,
                         nfev_stiff_detect=nfev_stiff_detect,
                         sc_params=sc_params, **extraneous)
        self.interpolant = interpolant

    # ref bogacki.py:340-346
    def _estimate_error_norm_pre(self, y, h):
        return self._rms_from_sumsq(
            self._dev.rk_pre_error_sumsq(h, self.E_pre, self.B_scale_pre))

    def _step_impl(self):
        """ref bogacki.py:238-338"""
        t = self.t
        s = self.n_stages
        h_abs, min_step = self._reassess_stepsize(t)
        rejected = False
        while True:
            if h_abs < min_step:
                return False, self.TOO_SMALL_STEP
            h = h_abs * self.direction
            t_new = t + h
            self._run_stages(1, s - 1, t, h)
            pre = self._estimate_error_norm_pre(None, h)
            if pre > 1:
                # early rejection: the last two evaluations are saved
                rejected = True
                h_abs *= self._reject_factor(pre)
                NFS[()] += 1
                if self.nfev_stiff_detect:
                    self.jflstp += 1
                continue
            self._run_stages(s - 1, s, t, h)
            error_norm = self._solution_and_error(t, h)
            if error_norm < 1:
                h_abs *= self._accept_factor(error_norm, h, rejected)
                break
            if np.isnan(error_norm) or np.isinf(error_norm):
                return False, "Overflow or underflow encountered."
            rejected = True
            h_abs *= self._reject_factor(error_norm)
            NFS[()] += 1
            self.jflstp += 1
        self._finish_step(t_new, h, h_abs)
        self.h_previous = h
        self.h_abs = h_abs
        self.error_norm_old = error_norm
        self.t = t_new
        self._diagnose_stiffness()
        return True, None

    # ------------------------------------------------------------ interpolants
    def _extra_stage(self, row, a_row, c, h):
        """K_last[row] = f(t_old + c*h, y_old + h * sum_j a_row[j] K_last[j])
        (ref bogacki.py:356-368); runs on the device rows of the finished step
        via the scratch tableau slot of `esq_rk_dense_stage`."""
        a = np.ascontiguousarray(a_row[:row], dtype=np.float64)
        self._chk(self._lib.esq_rk_dense_stage(self._ctx, row, as_ptr(a), row,
                                               float(h)), "esq_rk_dense_stage")
        t_stage = self.t_old + c * h
        if self._device_rhs is not None:
            self._chk(self._lib.esq_rk_dense_eval(self._ctx, row, t_stage),
                      "esq_rk_dense_eval")
            self.nfev += 1
        else:
            y_stage = self._dev.download(SLOT_YSTAGE)
            k = np.ascontiguousarray(self.fun(t_stage, y_stage),
                                     dtype=self._dev.dtype)
            self._chk(self._lib.esq_rk_upload_last_K(self._ctx, row, as_ptr(k)),
                      "esq_rk_upload_last_K")

    def _dense_output_impl(self):
        h = self.h_previous
        s = self.n_stages
        if self.interpolant == 'free':
            return self._horner_interpolant(self.P, self.t_old, self.t)
        if self.interpolant == 'low':
            self._extra_stage(s + 1, self.A_extra[0], self.C_extra[0], h)
            return self._horner_interpolant(self.Plow, self.t_old, self.t)
        for k, (a, c) in enumerate(zip(self.A_extra, self.C_extra)):
            self._extra_stage(s + 1 + k, a, c, h)
        # RKSUITE's 'best' interpolant looks back from the END of the step
        # (ref bogacki.py:370-393): Q[:, 0] = K[7], higher columns from Pbest
        # (the device sums each column in ascending row order; the reference
        # groups the terms by magnitude -- differences are O(1e-16) relative).
        P = self.Pbest.copy()
        P[:, 0] = 0.0
        P[7, 0] = 1.0
        return self._horner_interpolant(P, self.t, self.t + h, from_end=True)


install(BS5, "BS5")
