"""Pr7, Pr8, Pr9: P.J. Prince's 7(5) 10-stage, 8(6) 13-stage and 9(7) 17-stage
non-FSAL pairs with free interpolants ("Parallel Derivation of Efficient
Continuous/Discrete Explicit Runge-Kutta Methods", 2018).  Tableaux only; the
step is the generic device-resident `RungeKutta` step.  Reference counterpart:
extensisq/prince.py:79-128, 205-372, 449-746 (controllers "S", "G",
"standard").  Pr8 at n = 1e7 is the north-star workload of BASELINE.json."""
from ._tableau import install
from .common import RungeKutta


class Pr7(RungeKutta):
    pass


class Pr8(RungeKutta):
    pass


class Pr9(RungeKutta):
    pass


install(Pr7, "Pr7")
install(Pr8, "Pr8")
install(Pr9, "Pr9")
