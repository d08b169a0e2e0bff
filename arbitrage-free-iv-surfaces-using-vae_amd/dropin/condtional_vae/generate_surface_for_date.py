"""Drop-in for the function the reference's older flat script ``generate_surface_for_date``
defines: ``generate_iv_surface(params, T_grid, logm_grid, spot, r, q) -> (iv, success)``
(generate_surface_for_date.py:462-504) and the lattice it is called on (:458-459).

That function is the same computation, line for line in behaviour, as
generate_iv_surface_by_date.py:267-322 (0.99 x intrinsic guard, brentq on [0.001, 3.0] with
xtol 1e-6, success when fewer than 10 % of the points are NaN), so both names resolve to one GPU
implementation.  The rest of that script (checkpoint loading, BigQuery fetch, plots) runs at
import time in the reference and has no counterpart here: use ``generate_iv_surface_by_date.main``
for the command line.
"""
from __future__ import annotations

import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from generate_iv_surface_by_date import (  # noqa: E402,F401
    T_GRID as T_grid,
    LOGM_GRID as logm_grid,
    generate_iv_surface,
    generate_iv_surfaces,
)

__all__ = ["T_grid", "logm_grid", "generate_iv_surface", "generate_iv_surfaces"]
