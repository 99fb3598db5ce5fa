"""Me4: Merson's 4(3) pair, 5 stages, non-FSAL, with a free interpolant.
Tableau only -- generic device-resident `RungeKutta` step (reference
counterpart: extensisq/merson.py:5-122)."""
from ._tableau import install
from .common import RungeKutta


class Me4(RungeKutta):
    pass


install(Me4, "Me4")
