"""CFMR7osc: the 7(5) pair of Calvo, Franco, Montijano & Randez for
oscillatory problems (J. Comput. Appl. Math. 76 (1996) 195-212), 9 stages,
non-FSAL.  Like BS5 it tests an early error estimate -- after 8 stages, with
the weights `E[:8]` and the scale built from `y + h*K[:8].T@A[8,:8]` -- and
skips the last stage of a step that is going to be rejected (reference
counterpart: extensisq/calvo.py:152-261).  Both estimates are HIP kernels
(`esq_rk_pre_error`, `esq_rk_solution_error`)."""
import numpy as np

from ._tableau import install
from .common import NFS, RungeKutta


class CFMR7osc(RungeKutta):

    def _estimate_error_norm_pre(self, y, h):
        s = self.n_stages
        return self._rms_from_sumsq(self._dev.rk_pre_error_sumsq(
            h, self.E[:s - 1], self.A[s - 1, :s - 1]))

    def _step_impl(self):
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
            rejected = True
            h_abs *= self._reject_factor(error_norm)
            NFS[()] += 1
            self.jflstp += 1
            if np.isnan(error_norm) or np.isinf(error_norm):
                return False, "Overflow or underflow encountered."
        self._finish_step(t_new, h, h_abs)
        self.h_previous = h
        self.h_abs = h_abs
        self.error_norm_old = error_norm
        self.t = t_new
        self._diagnose_stiffness()
        return True, None


install(CFMR7osc, "CFMR7osc")
