"""The oracle's restatement of the selection loop of matchCandidates (orc_select_matches, reprojector.cpp:342-382) against an
independent one in Python (tests/select_helpers.py): visited flags, grid, counters, where the loop ends.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from select_helpers import sequential, make_lists


@pytest.mark.parametrize("n_cells", [416, 1, 5000])
def test_oracle_selection_equals_the_python_loop(n_cells):
    lib = oracle.load()
    lib.orc_select_matches.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_select_matches.restype = C.c_int
    rng = np.random.RandomState(100 + n_cells)
    n_cut = 0
    for c, s, o, mx, nb in make_lists(rng, 40, n_cells):
        v, o2, tr, ma, i, nf = sequential(c, s, o, mx, nb)
        ov = np.zeros(max(1, c.size), np.uint8); oo = o.copy(); onf = C.c_int(nb); otr = C.c_int(); oma = C.c_int()
        oi = lib.orc_select_matches(c.size, c.ctypes.data, s.ctypes.data, n_cells, oo.ctypes.data, mx, C.byref(onf), ov.ctypes.data, C.byref(otr), C.byref(oma))
        assert (oi, otr.value, oma.value, onf.value) == (i, tr, ma, nf)
        assert np.array_equal(ov[:c.size], v) and np.array_equal(oo, o2)
        n_cut += int(c.size > 0 and i < c.size)
    assert n_cells == 1 or n_cut > 0
