"""oracle -- CPU checker for the nr3d hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this package.
Nothing under nr3d_lib_amd/ imports it (tests/test_boundary.py greps for that).

The arithmetic lives in the C restatement (lotd_oracle.c, occ_grid_oracle.c, pack_ops_oracle.c), each
function citing the reference file:line it follows; this module is a numpy/ctypes veneer over
``liboracle.so`` (build: ``make -C oracle``).
"""
from .api import *  # noqa: F401,F403
