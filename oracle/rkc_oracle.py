"""CPU ORACLE (test infrastructure, NOT product code).

NumPy restatement of the reference's Runge-Kutta-Chebyshev solver
`SSV2stab` (extensisq/sommeijer.py:17-406, itself a translation of rkc.f by
Sommeijer, Shampine & Verwer).  Only tests/, smoke() and bench.py's
cpu_baseline leg may import it.

Parity pinned: `tests/test_oracle_golden.py` checks this module against stage
vectors / step traces generated from the real reference
(`tools/gen_golden.py`) and against the published integer table of
docs/Demo_SSV2stab.ipynb:350-356 (steps(failed)/nfev/s-max of the 3-D heat
problem).  The reference's own tests/ hold no SSV2stab test.
"""
from math import cosh, log, sinh, sqrt
from warnings import warn

import numpy as np
from scipy.integrate._ivp.base import OdeSolver
from scipy.integrate._ivp.common import (validate_first_step,
                                         validate_max_step, warn_extraneous)

from .rk_oracle import (NFS, HermiteInterpolant, check_tolerances, error_scale,
                       rms)

nrejct = NFS                  # sommeijer.py:12
nfesig = np.array(0)          # sommeijer.py:13
maxm = np.array(0)            # sommeijer.py:14


def chebyshev_stage_scalars(m):
    """Scalar recurrences of sommeijer.py:278-314, returned per stage so the
    device path can be checked against the same numbers.

    Returns (mus1, [(mu, nu, mus, ajm1, thjm1) for j = 2..m]) where `thjm1` is
    the time fraction at which stage j evaluates the RHS."""
    w0 = 1.0 + 2.0 / (13.0 * m ** 2)
    temp1 = w0 ** 2 - 1.0
    temp2 = sqrt(temp1)
    arg = m * log(w0 + temp2)
    w1 = sinh(arg) * temp1 / (cosh(arg) * m * temp2 - w0 * sinh(arg))
    bjm1 = bjm2 = 1.0 / (2.0 * w0) ** 2
    mus1 = w1 * bjm1
    thjm2, thjm1 = 0.0, mus1
    zjm1, zjm2 = w0, 1.0
    dzjm1, dzjm2 = 1.0, 0.0
    d2zjm1, d2zjm2 = 0.0, 0.0
    rows = []
    for j in range(2, m + 1):
        zj = 2.0 * w0 * zjm1 - zjm2
        dzj = 2.0 * w0 * dzjm1 - dzjm2 + 2.0 * zjm1
        d2zj = 2.0 * w0 * d2zjm1 - d2zjm2 + 4.0 * dzjm1
        bj = d2zj / dzj ** 2
        ajm1 = 1.0 - zjm1 * bjm1
        mu = 2.0 * w0 * bj / bjm1
        nu = -bj / bjm2
        mus = mu * w1 / w0
        rows.append((mu, nu, mus, ajm1, thjm1))
        thj = mu * thjm1 + nu * thjm2 + mus * (1.0 - ajm1)
        if j < m:
            thjm2, thjm1 = thjm1, thj
            bjm2, bjm1 = bjm1, bj
            zjm2, zjm1 = zjm1, zj
            dzjm2, dzjm1 = dzjm1, dzj
            d2zjm2, d2zjm1 = d2zjm1, d2zj
    return mus1, rows


class SSV2stab(OdeSolver):
    def __init__(self, fun, t0, y0, t_bound, max_step=np.inf, rtol=1e-3,
                 atol=1e-6, vectorized=False, first_step=None,
                 const_jac=False, rho_jac=None, **extraneous):
        # sommeijer.py:93-145
        warn_extraneous(extraneous)
        super().__init__(fun, t0, y0, t_bound, vectorized,
                         support_complex=False)
        self.absh = (None if first_step is None
                     else validate_first_step(first_step, t0, t_bound))
        self.hold = None
        if not isinstance(const_jac, bool):
            raise TypeError('`const_jac` should be True or False')
        if rho_jac is not None:
            if not callable(rho_jac):
                raise TypeError('`rho_jac` should be None or a function: '
                                '`sprad = rho_jac(t, y)`')
            if not isinstance(rho_jac(self.t, self.y), float):
                raise TypeError('`rho_jac` should return a float')
            if rho_jac(self.t, self.y) <= 0:
                raise ValueError('`rho_jac` should return a positive float')
        self.const_jac = const_jac
        self.rho_jac = rho_jac
        self.max_step = validate_max_step(max_step)
        self.rtol, self.atol = check_tolerances(rtol, atol, self.y)
        info = np.finfo(self.y.dtype)
        self.uround = np.nextafter(info.epsneg, 1)
        self.sqrtu = sqrt(self.uround)
        self.sqrtmin = sqrt(info.tiny)
        self.W = np.empty((4, self.n), self.y.dtype)
        self.V = None
        nrejct[()] = 0
        nfesig[()] = 0
        maxm[()] = 0
        self.nstsig = 0
        self.mlim = 0
        self.mmax = max(int(round(sqrt(self.rtol / (10.0 * self.uround)))), 2)
        self.newspc = True
        self.jacatt = False
        self.W[0] = self.y
        self.W[1] = self.fun(self.t, self.y)
        max_step = min(self.max_step, abs(self.t_bound - self.t))
        self.max_step = min(max_step, sqrt(info.max))
        hmin = abs(self.t)
        if self.t_bound != np.inf:
            hmin = max(hmin, abs(self.max_step))
        self.hmin = max(self.sqrtmin, 10.0 * self.uround * hmin)
        self.trace = []     # (t_new, h, m, err, accepted)

    # the two places where a lock-step batch exchanges a scalar (overridden by
    # tests/test_lockstep_cpu.py; the arithmetic here is the reference's)
    def _err_norm(self, r):
        return rms(r)

    def _rho_user(self, t, yn):
        return self.rho_jac(t, yn)

    def _initial_step(self, t, yn, fn, vtemp1, vtemp2):      # :147-160
        absh = self.max_step
        if self.sprad * absh > 1.0:
            absh = 1.0 / self.sprad
        absh = max(absh, self.hmin)
        vtemp1[:] = yn + absh * fn
        vtemp2[:] = self.fun(t + absh, vtemp1)
        wt = self.atol + self.rtol * np.abs(yn)
        est = absh * self._err_norm((vtemp2 - fn) / wt)
        if 0.1 * absh < self.max_step * sqrt(est):
            return max(0.1 * absh / sqrt(est), self.hmin)
        return self.max_step

    def _step_impl(self):                                    # :162-271
        t = self.t
        absh = self.absh
        y = self.y.copy()
        yn, fn, vtemp1, vtemp2 = self.W
        while True:
            if self.newspc:
                if self.rho_jac is not None:
                    self.sprad = self._rho_user(t, yn)
                else:
                    self.sprad = self._spectral_radius(t, yn, fn, vtemp1,
                                                       vtemp2)
                    if self.sprad is None:
                        return False, (
                            "The method to estimate the spectral radius "
                            "of the Jacobian did not converge")
                self.jacatt = True
            if absh is None:
                absh = self._initial_step(t, yn, fn, vtemp1, vtemp2)
            if 1.1 * absh >= abs(self.t_bound - t):
                absh = abs(self.t_bound - t)
            m = 1 + int(sqrt(1.54 * absh * self.sprad + 1.0))
            if m > self.mmax:
                m = self.mmax
                absh = (m ** 2 - 1) / (1.54 * self.sprad)
                self.mlim += 1
                if self.mlim == 15:
                    warn('Your problem is too stiff for this method.')
            else:
                self.mlim = 0
            maxm[()] = max(m, maxm[()])
            h = self.direction * absh
            hmin = max(self.sqrtmin,
                       13.3 * self.uround * (abs(t) + absh) * (m ** 2 - 1))
            self._stages(t, yn, fn, h, m, y, vtemp1, vtemp2)
            vtemp1[:] = self.fun(t + h, y)
            wt = error_scale(self.atol, self.rtol, y, yn)
            est = 0.8 * (yn - y) + 0.4 * h * (fn + vtemp1)
            err = self._err_norm(est / wt)
            self.trace.append((t + h, h, m, float(err), err < 1.0))
            if err < 1.0:
                break
            if np.isnan(err) or np.isinf(err):
                return False, "Overflow or underflow encountered."
            nrejct[()] += 1
            absh = 0.8 * absh / err ** (1 / 3)
            if absh < hmin:
                return False, self.TOO_SMALL_STEP
            self.newspc = not self.jacatt
            self.absh = absh

        t += h
        self.jacatt = self.const_jac
        self.nstsig = (self.nstsig + 1) % 25
        self.newspc = False
        if self.rho_jac is not None or self.nstsig == 0:
            self.newspc = not self.jacatt
        ylast = yn.copy()
        yplast = fn.copy()
        yn[:] = y
        fn[:] = vtemp1
        vtemp1[:] = ylast
        vtemp2[:] = yplast
        fac = 10.0
        if self.hold is None:
            temp2 = err ** (1 / 3)
            if 0.8 < fac * temp2:
                fac = 0.8 / temp2
        else:
            temp1 = 0.8 * absh * self.errold ** (1 / 3)
            temp2 = abs(self.hold) * err ** (2 / 3)
            if temp1 < fac * temp2:
                fac = temp1 / temp2
        absh = max(0.1, fac) * absh
        self.absh = max(hmin, min(self.max_step, absh))
        self.errold = err
        self.hold = h
        self.y = y
        self.t = t
        return True, None

    def _stages(self, t, yn, fn, h, m, y, yjm1, yjm2):       # :273-329
        mus1, rows = chebyshev_stage_scalars(m)
        yjm2[:] = yn
        yjm1[:] = yn + h * mus1 * fn
        for idx, (mu, nu, mus, ajm1, thjm1) in enumerate(rows):
            y[:] = self.fun(t + h * thjm1, yjm1)
            y[:] = (mu * yjm1 + nu * yjm2 + (1.0 - mu - nu) * yn +
                    h * mus * (y - ajm1 * fn))
            if idx < len(rows) - 1:
                yjm2[:] = yjm1
                yjm1[:] = y

    def _spectral_radius(self, t, yn, fn, v, fv):            # :331-398
        small = 1.0 / self.max_step
        if self.V is None:
            self.V = fn.copy()
        v[:] = self.V
        ynrm = np.linalg.norm(yn)
        vnrm = np.linalg.norm(v)
        if ynrm != 0.0 and vnrm != 0.0:
            dynrm = ynrm * self.sqrtu
            v[:] = yn + v * (dynrm / vnrm)
        elif ynrm != 0.0:
            dynrm = ynrm * self.sqrtu
            v[:] *= 1.0 + self.sqrtu
        elif vnrm != 0.0:
            dynrm = self.uround
            v[:] *= dynrm / vnrm
        else:
            dynrm = self.uround
            v[:] = dynrm
        sigma = 0.0
        for it in range(50):
            fv[:] = self.fun_single(t, v)
            nfesig[()] += 1
            dfnrm = np.linalg.norm(fv - fn)
            sigmal = sigma
            sigma = dfnrm / dynrm
            sprad = 1.2 * sigma
            if it and abs(sigma - sigmal) <= max(sigma, small) * 0.01:
                self.V[:] = v - yn
                return sprad
            if dfnrm != 0.0:
                v[:] = yn + (fv - fn) * (dynrm / dfnrm)
            else:
                index = it % self.n
                v[index] = -v[index]
        return None

    def _dense_output_impl(self):                            # :400-406
        y, f, y_old, f_old = self.W[:4].copy()
        return HermiteInterpolant(self.t_old, self.t, y_old, y, f_old, f)
