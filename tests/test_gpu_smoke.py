"""The driver's own first contact: `__graft_entry__.smoke()` (three Pr8 steps with
chain sweeps against the oracle) -- in the suite, so that a change which breaks it
shows up with the other GPU tests and not only at the end of a round."""
import os
import sys

import pytest

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(__file__), "..")))


@pytest.mark.gpu
def test_smoke_entry_point(monkeypatch):
    import __graft_entry__ as entry
    monkeypatch.setenv("ESQ_CHAIN_ROWS", "12")
    entry.smoke()
