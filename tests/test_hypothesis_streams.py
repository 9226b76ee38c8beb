"""CPU: the hypothesis streams of all ten RANSAC / PROSAC solvers, product (rpe_host_hypotheses: the drop-in headers' samplers +
minimal solvers, no GPU) against the oracle (orc_hypotheses: the CPU restatement's), same seed.  Every hypothesis -- quaternion
and translation in Tp -- and every per-iteration count must be IDENTICAL, bit for bit: this is what makes whole pipelines
vote / Iter / mask exact on noisy scenes (tests/test_gpu_pipelines.py), because the scoring kernels are bit-exact given equal
hypotheses (tests/test_gpu_kernels.py)."""
import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

METHODS = [  # name, method id (same ids in api and oracle), arrays
    ("shinji_ransac", 0, ("xw", "xc", "bv")), ("shinji_ransac2", 1, ("xw", "xc")), ("shinji_prosac", 2, ("xw", "xc")),
    ("kneip_ransac", 3, ("xw", "bv")), ("kneip_prosac", 4, ("xw", "bv")), ("shinji_kneip_ransac", 5, ("xw", "xc", "bv")),
    ("shinji_kneip_prosac", 6, ("xw", "xc", "bv")), ("nl_kneip_ransac", 7, ("xw", "xc", "bv", "nw", "nc")),
    ("nl_shinji_ransac", 8, ("xw", "xc", "bv", "nw", "nc")), ("nl_shinji_kneip_ransac", 9, ("xw", "xc", "bv", "nw", "nc")),
]


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("m", METHODS, ids=[m[0] for m in METHODS])
def test_hypothesis_stream_is_the_oracles(oracle, m, f64):
    name, method, arrays = m
    dt = np.float64 if f64 else np.float32
    sc = util.scene_full(77 + method, 3000, dt, n2d=1.0, n3d=0.05, nnl_deg=2.0, outliers=0.2, nan_frac=0.05)
    data = dict(xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    sel = {k: data[k] for k in arrays}
    q1, f1 = api.host_hypotheses(method, L.F64 if f64 else L.F32, weights=sc.weights, iters=300, seed=9, **sel)
    q2, f2 = oracle.hypotheses(oracle.Problem(f64, weights=sc.weights, **sel), method, 300, seed=9)
    assert len(q1) >= 200
    assert np.array_equal(f1, f2)
    assert np.array_equal(q1, q2, equal_nan=True)


def test_stream_layout_and_errors(oracle):
    sc = util.scene_full(5, 500, np.float32)
    q, first = api.host_hypotheses(api.M_SK_RANSAC, L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, iters=50, seed=3)
    assert first[0] == 0 and first[-1] == len(q) and np.all(np.diff(first) >= 0) and np.all(np.diff(first) <= 2)
    assert np.allclose(np.linalg.norm(q[:, :4], axis=1), 1.0, atol=1e-4)   # rotations (not renormalised: SO3(Matrix3) semantics)
    # a different seed gives a different stream; the same seed the same one
    q2, _ = api.host_hypotheses(api.M_SK_RANSAC, L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, iters=50, seed=4)
    q3, _ = api.host_hypotheses(api.M_SK_RANSAC, L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, iters=50, seed=3)
    assert not np.array_equal(q[:10], q2[:10]) and np.array_equal(q, q3)
    prob, n, keep = api._problem(L.F32, sc.Q, sc.P, sc.U, None, None, None, 585.0)
    import ctypes as C
    out, f = np.zeros((4, 7)), np.zeros(51, np.int32)
    assert L.lib().rpe_host_hypotheses(api.M_SK_RANSAC, C.byref(prob), 50, 3, out.ctypes.data_as(C.c_void_p), 4, f.ctypes.data_as(C.c_void_p)) == L.RPE_ERR_ARG
