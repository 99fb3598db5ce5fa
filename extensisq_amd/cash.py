"""CK5: the Cash-Karp 5(4) pair (ACM TOMS 16 (1990) 201-222), 6 stages,
non-FSAL, free 4th-order interpolant.  Tableau only -- it rides on the generic
device-resident `RungeKutta` step (reference counterpart: extensisq/cash.py:
9-112).  The variable-order `CKdisc` of the same file is out of scope."""
from ._tableau import install
from .common import RungeKutta


class CK5(RungeKutta):
    pass


install(CK5, "CK5")
