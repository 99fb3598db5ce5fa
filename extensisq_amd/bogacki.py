"""BS5: the Bogacki-Shampine 5(4) pair (Comput. Math. Appl. 32 (1996) 15-28;
coefficients as in RKSUITE), 7 effective stages, FSAL, with TWO error
estimates: an early one after six stages that can reject a step before the
last two RHS evaluations, and the usual one at the end.  Reference counterpart:
extensisq/bogacki.py:103-393.

Device mapping: a whole attempt is TWO chain sweeps -- stages 1..5 with the early
estimate as their last target, then stage 6, y_new, the FSAL stage and the error
norm -- enqueued in one go (`RungeKutta._step_impl_early`, `esq_rk_set_pre`); with
a Python right-hand side the round-5 sequence of pieces (`esq_rk_stages`,
`esq_rk_pre_error`, `esq_rk_solution_error`).  The controller is host scalar
arithmetic."""
import numpy as np

from ._lib import SLOT_K, SLOT_WORK, SLOT_YNEW, SLOT_YSTAGE, as_ptr  # noqa: F401
from ._tableau import install
from .common import RungeKutta


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

    def _early_estimate(self):
        return self.E_pre, self.B_scale_pre

    def _step_impl(self):
        """ref bogacki.py:238-338"""
        return self._step_impl_early(nan_check_first=True)

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
