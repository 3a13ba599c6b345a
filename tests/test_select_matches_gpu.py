"""svoh_select_matches_batch (reprojector.cpp:342-382: the loop of matchCandidates over given matches -- first
success per free cell, the max_n cut, which candidates were tried) against the sequential loop, restated twice: in the oracle (C)
and here in Python.  Integer work: everything exact."""
import ctypes as C

import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi
from oracle import oracle

pytestmark = pytest.mark.gpu


from select_helpers import sequential, make_lists   # noqa: E402


@pytest.mark.parametrize("n_cells", [416, 1, 5000])
def test_selection_equals_the_sequential_loop(gpu_ctx, n_cells):
    ctx = gpu_ctx
    rng = np.random.RandomState(100 + n_cells)
    lists = make_lists(rng, 40, n_cells)
    lib = oracle.load()
    lib.orc_select_matches.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_select_matches.restype = C.c_int
    begin = np.zeros(len(lists) + 1, np.int32)
    begin[1:] = np.cumsum([l[0].size for l in lists])
    cell = np.concatenate([l[0] for l in lists]).astype(np.int32)
    success = np.concatenate([l[1] for l in lists]).astype(np.uint8)
    occ = np.concatenate([l[2] for l in lists]).astype(np.uint8)
    max_n = np.array([l[3] for l in lists], np.int32)
    n_feat = np.array([l[4] for l in lists], np.int32)
    visited = np.full(max(1, cell.size), 7, np.uint8)
    trials = np.zeros(len(lists), np.int32); matches = np.zeros(len(lists), np.int32); consumed = np.zeros(len(lists), np.int32)
    ctx._check(ctx.lib.svoh_select_matches_batch(ctx.h, len(lists), begin.ctypes.data, cell.ctypes.data, success.ctypes.data, n_cells, occ.ctypes.data,
                                                 max_n.ctypes.data, n_feat.ctypes.data, visited.ctypes.data, trials.ctypes.data, matches.ctypes.data, consumed.ctypes.data))
    some_cut = some_through = False
    for l, (c, s, o, mx, nb) in enumerate(lists):
        v, o2, tr, ma, i, nf = sequential(c, s, o, mx, nb)
        # the oracle's C restatement agrees with the Python one ...
        ov = np.zeros(max(1, c.size), np.uint8); oo = o.copy(); onf = C.c_int(nb); otr = C.c_int(); oma = C.c_int()
        oi = lib.orc_select_matches(c.size, c.ctypes.data, s.ctypes.data, n_cells, oo.ctypes.data, mx, C.byref(onf), ov.ctypes.data, C.byref(otr), C.byref(oma))
        assert (oi, otr.value, oma.value, onf.value) == (i, tr, ma, nf) and np.array_equal(ov[:c.size], v) and np.array_equal(oo, o2)
        # ... and the device with both
        lo, hi = begin[l], begin[l + 1]
        assert np.array_equal(visited[lo:hi], v), l
        assert np.array_equal(occ[l * n_cells:(l + 1) * n_cells], o2), l
        assert (trials[l], matches[l], consumed[l], n_feat[l]) == (tr, ma, i, nf), l
        some_cut = some_cut or (c.size and i < c.size)
        some_through = some_through or (c.size and i == c.size and ma > 0)
    assert n_cells == 1 or (some_cut and some_through)   # (one cell: at most one match, the case is about the degenerate grid)
    # misuse: max_n = 0 (the reference's loop ignores the grid then), a decreasing begin
    bad = max_n.copy(); bad[3] = 0
    assert ctx.lib.svoh_select_matches_batch(ctx.h, len(lists), begin.ctypes.data, cell.ctypes.data, success.ctypes.data, n_cells, occ.ctypes.data, bad.ctypes.data,
                                             n_feat.ctypes.data, visited.ctypes.data, trials.ctypes.data, matches.ctypes.data, consumed.ctypes.data) != 0
    b2 = begin.copy(); b2[2] = b2[1] - 1 if b2[1] > 0 else -1
    assert ctx.lib.svoh_select_matches_batch(ctx.h, len(lists), b2.ctypes.data, cell.ctypes.data, success.ctypes.data, n_cells, occ.ctypes.data, max_n.ctypes.data,
                                             n_feat.ctypes.data, visited.ctypes.data, trials.ctypes.data, matches.ctypes.data, consumed.ctypes.data) != 0
