"""The driver's smoke() entry point, run as a test so that a change of the grid's stored order (which smoke() restates for
its oracle replay) cannot break it unnoticed."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_smoke_entry_point():
    import __graft_entry__ as entry
    entry.smoke()
