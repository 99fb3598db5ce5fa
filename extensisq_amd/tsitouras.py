"""Ts5: Tsitouras' 5(4) pair, 6 effective stages, FSAL, free 4th-order
interpolant (Tsitouras, Comput. Math. Appl. 62 (2011) 770-775).  Tableau only:
the step is the generic device-resident `RungeKutta` step.  Reference
counterpart: extensisq/tsitouras.py:83-115 (default controller "G")."""
from ._tableau import install
from .common import RungeKutta


class Ts5(RungeKutta):
    pass


install(Ts5, "Ts5")
