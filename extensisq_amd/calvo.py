"""CFMR7osc: the 7(5) pair of Calvo, Franco, Montijano & Randez for
oscillatory problems (J. Comput. Appl. Math. 76 (1996) 195-212), 9 stages,
non-FSAL.  Like BS5 it tests an early error estimate -- after 8 stages, with
the weights `E[:8]` and the scale built from `y + h*K[:8].T@A[8,:8]` -- and
skips the last stage of a step that is going to be rejected (reference
counterpart: extensisq/calvo.py:152-261).  Both estimates are HIP kernels
(`esq_rk_pre_error`, `esq_rk_solution_error`)."""
from ._tableau import install
from .common import RungeKutta


class CFMR7osc(RungeKutta):

    def _estimate_error_norm_pre(self, y, h):
        s = self.n_stages
        return self._rms_from_sumsq(self._dev.rk_pre_error_sumsq(
            h, self.E[:s - 1], self.A[s - 1, :s - 1]))

    def _early_estimate(self):
        s = self.n_stages
        return self.E[:s - 1], self.A[s - 1, :s - 1]

    def _step_impl(self):
        """ref calvo.py:152-253"""
        return self._step_impl_early(nan_check_first=False)


install(CFMR7osc, "CFMR7osc")
