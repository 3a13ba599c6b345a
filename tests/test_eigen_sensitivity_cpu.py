"""tests/golden/eigen_sensitivity.json (made by tests/golden/make_eigen_sensitivity.py): how many outputs of the hot
path depend on third-party arithmetic the reference links but does not contain -- Eigen's vectorised 4x4 float
inverse vs the generic one in align2D, and the summation order inside its LDLT.  The committed counts say "none on
the benchmark's workloads"; this test re-derives the small subset and pins the alternative inverse itself."""
import ctypes as C
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _study():
    spec = importlib.util.spec_from_file_location("make_eigen_sensitivity", os.path.join(HERE, "golden", "make_eigen_sensitivity.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_vectorised_inverse_is_an_inverse_and_differs_only_in_rounding(oracle_lib):
    lib = oracle_lib.load()
    rng = np.random.RandomState(3)
    n_diff = 0
    for t in range(500):
        J = rng.randn(64, 4).astype(np.float32)
        J[:, 2] = 1.0
        H = np.ascontiguousarray((J.T @ J).astype(np.float32).ravel())
        r0 = np.zeros(16, np.float32); r1 = np.zeros(16, np.float32)
        lib.orc_mat4f_inverse(H.ctypes.data_as(C.c_void_p), r0.ctypes.data_as(C.c_void_p), 0)
        lib.orc_mat4f_inverse(H.ctypes.data_as(C.c_void_p), r1.ctypes.data_as(C.c_void_p), 1)
        ref = np.linalg.inv(H.reshape(4, 4).astype(np.float64))
        for r in (r0, r1):
            assert np.abs(r.reshape(4, 4) - ref).max() <= 2e-6 * np.abs(ref).max()
        n_diff += int(not np.array_equal(r0, r1))
    assert n_diff > 400      # the two paths really round differently (else the study below would be vacuous)


def test_committed_counts_and_subset_reproduces(oracle_lib):
    d = json.load(open(os.path.join(HERE, "golden", "eigen_sensitivity.json")))
    big = d["seeds_offset_only_64x3000"]
    assert big["seeds"] == 192000 and big["refined_2d"] > 100000
    # the statement DESIGN.md makes: no result code / success flag / feature type of the benchmark's 192 000 seeds
    # depends on which of Eigen's two 4x4 inverses runs; states agree to 1e-4 relative
    assert big["result_code_changed"] == 0 and big["success_changed"] == 0 and big["type_changed"] == 0
    assert big["max_rel_state_diff_same_code"] < 1e-4
    assert d["match_direct_8x2000"]["result_code_changed"] == 0 and d["match_direct_8x2000"]["max_px_diff_same_code"] <= 1e-4
    assert d["ldlt_align_32x2000"]["iteration_counts_changed"] == 0 and d["ldlt_align_32x2000"]["max_pose_diff"] < 1e-12
    m = _study()
    got = m.subset()
    for sect, want in d["subset"].items():
        for k, v in want.items():
            if isinstance(v, float):
                assert abs(got[sect][k] - v) <= 1e-9 * max(1.0, abs(v)) + 1e-12, (sect, k)
            else:
                assert got[sect][k] == v, (sect, k)
