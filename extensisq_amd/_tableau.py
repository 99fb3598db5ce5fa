"""Butcher-tableau constants, stored as IEEE-754 hex in data/tableaus.json so
that the class attributes are bit-identical to the published coefficient sets
the reference carries (Bogacki & Shampine 1996 / RKSUITE; Tsitouras 2011;
Prince 2018 -- extensisq/bogacki.py:103-215, tsitouras.py:83-115,
prince.py:79-128, 205-372, 449-746).  tests/test_tableaus.py re-derives the
order conditions independently."""
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                     "tableaus.json")
_cache = None


def _matrix(d):
    M = np.zeros(d["shape"])
    for i, j, s in d["nz"]:
        M[i, j] = float.fromhex(s)
    return M


def tableau(name):
    """dict of ndarrays / scalars for method `name`"""
    global _cache
    if _cache is None:
        with open(_PATH) as fh:
            _cache = json.load(fh)
    out = {}
    for key, val in _cache[name].items():
        if isinstance(val, dict):
            out[key] = _matrix(val)
        elif isinstance(val, list):
            out[key] = np.array([float.fromhex(s) for s in val])
        elif isinstance(val, str) and key not in ("sc_params",):
            out[key] = float.fromhex(val)
        else:
            out[key] = val
    return out


def install(cls, name):
    """set the tableau of `name` as class attributes of `cls`"""
    for key, val in tableau(name).items():
        setattr(cls, key, val)
    return cls
