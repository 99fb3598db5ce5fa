"""Order conditions of Runge-Kutta methods from rooted trees (Butcher):
for every rooted tree t with |t| <= p:   sum_i b_i Phi_i(t) = 1 / gamma(t).

Own implementation (the reference checks orders <= 7 with tabulated
expressions, tests/order_conditions.py); trees are nested sorted tuples."""
from functools import lru_cache
from itertools import combinations_with_replacement

import numpy as np


@lru_cache(maxsize=None)
def trees(order):
    """all rooted trees with `order` vertices"""
    if order == 1:
        return ((),)
    out = set()
    for parts in _partitions(order - 1):
        pools = [trees(k) for k in parts]
        for combo in _product_sorted(pools):
            out.add(tuple(sorted(combo)))
    return tuple(sorted(out))


def _partitions(n, largest=None):
    largest = largest or n
    if n == 0:
        yield ()
        return
    for k in range(min(n, largest), 0, -1):
        for rest in _partitions(n - k, k):
            yield (k,) + rest


def _product_sorted(pools):
    if not pools:
        yield ()
        return
    for head in pools[0]:
        for rest in _product_sorted(pools[1:]):
            yield (head,) + rest


def order_of(t):
    return 1 + sum(order_of(c) for c in t)


def gamma(t):
    g = order_of(t)
    for c in t:
        g *= gamma(c)
    return g


def phi(t, A):
    """elementary weight vector Phi(t) (one entry per stage)"""
    out = np.ones(A.shape[0])
    for c in t:
        out = out * (A @ phi(c, A))
    return out


def max_residual(order, b, A):
    """largest |b.Phi(t) - 1/gamma(t)| over the trees of exactly `order`"""
    worst = 0.0
    for t in trees(order):
        worst = max(worst, abs(b @ phi(t, A) - 1.0 / gamma(t)))
    return worst
