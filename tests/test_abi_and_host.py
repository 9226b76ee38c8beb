"""CPU: librgbdpose_hip.so loads without a GPU, exports every symbol include/rgbd_pose_hip.h declares, fails loudly
when asked to compute without a device, and its host-side solver pieces (sampling, minimal solvers, small algebra)
agree with the golden fixtures and with the oracle -- sampled index streams bit for bit."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def G():
    g = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    g["arr"] = dict(np.load(os.path.join(HERE, "golden", "golden_inputs.npz")))
    return g


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "rgbd_pose_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(?:int|void|double|const char\*)\s+(\w+)\s*\(", hdr))
    declared.discard("rpe_status")
    assert {"ao", "ao_ransac", "py2c", "rpe_create", "rpe_normal_eq", "rpe_score", "rpe_run"} <= declared
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert declared <= exported, sorted(declared - exported)
    assert declared == set(L.SYMBOLS), sorted(declared ^ set(L.SYMBOLS))
    assert L.lib().rpe_abi_version() == 1


def test_no_gpu_means_loud_failure_not_fallback():
    if L.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(L.RpeError) as e:
        api.Context(0)
    assert e.value.code == L.RPE_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    sc = util.scene33(1, 50)
    with pytest.raises(L.RpeError) as e:
        api.run(api.M_SHINJI_RANSAC2, L.F32, xw=sc.Q, xc=sc.P, thre_3d=0.2, iters=10)
    assert e.value.code == L.RPE_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    bad = []
    for base in (os.path.join(ROOT, "rgbd_pose_estimation_amd"), os.path.join(ROOT, "include")):
        for d, _, fs in os.walk(base):
            for f in fs:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(d, f)).read()
                    if re.search(r"oracle_lib|liboracle|orc_\w+\(|#include\s+\"[^\"]*orc_|oracle/", txt):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def host_random_elements(n, m, seed, draws):
    out = np.zeros((draws, m), np.int32)
    L.lib().rpe_host_random_elements(n, m, seed, draws, _p(out))
    return out


def host_prosac(dtype, m, n, seed, draws):
    out = np.zeros((draws, m), np.int32)
    L.lib().rpe_host_prosac_samples(dtype, m, n, seed, draws, _p(out))
    return out


def test_samplers_bit_exact_with_golden_and_oracle(oracle, G):
    assert host_random_elements(100, 4, 3, 20).tolist() == G["random_elements_n100_m4_seed3"]
    assert host_random_elements(5, 3, 3, 10).tolist() == G["random_elements_n5_m3_seed3"]
    assert host_prosac(L.F64, 4, 100, 9, 300).tolist() == G["prosac_f64_m4_n100_seed9"]
    assert host_prosac(L.F32, 3, 50, 9, 300).tolist() == G["prosac_f32_m3_n50_seed9"]
    for n, m, seed in ((307200, 3, 1), (1000, 4, 77), (4, 4, 5)):
        assert np.array_equal(host_random_elements(n, m, seed, 50), oracle.random_elements(n, m, seed, 50))
    assert np.array_equal(host_prosac(L.F32, 4, 1000, 3, 2000), oracle.prosac_samples(False, 4, 1000, 3, 2000))
    assert np.array_equal(host_prosac(L.F64, 3, 12, 5, 30000), oracle.prosac_samples(True, 3, 12, 5, 30000))


def test_update_num_iters_and_sort(oracle, G):
    for p, ep, mp, mx, want in G["update_num_iters_f64"]:
        assert L.lib().rpe_host_update_num_iters(L.F64, p, ep, mp, mx) == want
        assert L.lib().rpe_host_update_num_iters(L.F32, p, ep, mp, mx) == oracle.ransac_update_num_iters(False, p, ep, mp, mx)
    w = np.random.default_rng(0).uniform(size=500)
    out = np.zeros(500, np.int32)
    L.lib().rpe_host_sort_indexes(_p(w), 500, _p(out))
    assert np.array_equal(out, oracle.sort_indexes(w))


def host_svd3(A):
    A = np.ascontiguousarray(A, np.float64)
    U, s, V = np.zeros(9), np.zeros(3), np.zeros(9)
    L.lib().rpe_host_svd3(_p(A), _p(U), _p(s), _p(V))
    return U.reshape(3, 3), s, V.reshape(3, 3)


def test_host_svd_and_pose_from_moments(G):
    rng = np.random.default_rng(1)
    for A in [rng.standard_normal((3, 3)) for _ in range(20)] + [np.outer([1, 2, 3.0], [0.5, -1, 2.0]), np.zeros((3, 3)), np.diag([2.0, 1.0, 0.0])]:
        U, s, V = host_svd3(A)
        assert np.allclose(U @ np.diag(s) @ V.T, A, atol=1e-12)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-12) and np.allclose(V.T @ V, np.eye(3), atol=1e-12)
        assert np.allclose(s, np.linalg.svd(A)[1], atol=1e-12)
    a = G["arr"]
    for name in ("pure_translation", "rot180_z", "rot180_axis", "planar", "mirror"):
        xw, xc = a[f"kab_{name}_xw"], a[f"kab_{name}_xc"]
        m = np.concatenate([[len(xw)], xw.sum(0), xc.sum(0), (xc.T @ xw).reshape(9), [np.sum(xc * xc)], [len(xw)]])
        R, t = api.pose_from_moments(m)
        k = G["kabsch"][name]
        assert util.rot_err(R, np.array(k["R"])) < 1e-9 and np.linalg.norm(t - np.array(k["t"])) < 1e-9, name
    with pytest.raises(L.RpeError) as e:
        api.pose_from_moments(np.zeros(18))
    assert e.value.code == L.RPE_ERR_DEGENERATE


def test_host_gn_solve_and_exp(oracle, G):
    rng = np.random.default_rng(2)
    J = rng.standard_normal((40, 6))
    r = rng.standard_normal(40)
    H, g = J.T @ J, J.T @ r
    rec = np.zeros(32); k = 0
    for i in range(6):
        for j in range(i, 6):
            rec[k] = H[i, j]; k += 1
    rec[21:27] = g
    d = api.gn_solve(rec)
    assert np.allclose(d, np.linalg.solve(H, -g), atol=1e-10)
    assert np.allclose(d, oracle.gn_solve(rec[:29])[0], atol=1e-12)
    with pytest.raises(L.RpeError):
        api.gn_solve(np.zeros(32))
    for e in G["se3_exp"]:
        R, t = np.zeros(9), np.zeros(3)
        a = np.array(e["a"])
        L.lib().rpe_host_se3_exp(_p(a), _p(R), _p(t))
        assert np.allclose(R.reshape(3, 3), e["R"], atol=1e-13) and np.allclose(t, e["t"], atol=1e-13)
    p0 = oracle.pose12(np.array(G["full_R"]), np.array(G["full_t"]))
    a = np.array(G["se3_exp"][2]["a"])
    assert np.allclose(api.gn_apply(a, p0), oracle.gn_apply(a, p0), atol=1e-13)


def _cm(x):  # (k,3) rows -> 3 x k column-major doubles
    return np.ascontiguousarray(x, np.float64)


@pytest.mark.parametrize("dtype", [L.F64, L.F32])
def test_host_minimal_solvers_match_oracle(oracle, G, dtype):
    """H1: the host-side minimal solvers evaluate the reference's expressions in Tp in the reference's operation order (the same
    published Jacobi SVD, the same quartic), so on every sample they return the SAME Tp values as the CPU restatement -- compared
    bit for bit, float and double; and they contain the truth on the noise-free golden samples."""
    a = G["arr"]
    f64 = dtype == L.F64
    R, t = np.array(G["full_R"]), np.array(G["full_t"])
    for i in range(0, 40, 4):
        Q, U = _cm(a["p3p_Q"][i:i + 4]), _cm(a["p3p_U"][i:i + 4])
        sols = np.zeros((4, 12))
        cnt = L.lib().rpe_host_kneip_main(dtype, _p(Q), _p(U), _p(sols))
        ref = oracle.kneip_main(Q, U, f64)
        assert cnt == len(ref)
        for k in range(cnt):  # same branch order as the reference's loop over the quartic's roots
            assert np.array_equal(sols[k, :9].reshape(3, 3), ref[k][0]) and np.array_equal(sols[k, 9:], ref[k][1])
        R9, t3 = np.zeros(9), np.zeros(3)
        assert L.lib().rpe_host_kneip(dtype, _p(Q), _p(U), _p(R9), _p(t3)) == 1
        assert util.rot_err(R9.reshape(3, 3), R) < (1e-8 if f64 else 3e-2)   # float P3P on a noise-free sample: conditioning, not parity
    for i in range(0, 40, 2):
        v = _cm(np.stack([a["p3p_P"][i], a["p3p_N"][i], a["p3p_P"][i + 1], a["p3p_Q"][i], a["p3p_M"][i], a["p3p_Q"][i + 1]]))
        R9, t3 = np.zeros(9), np.zeros(3)
        L.lib().rpe_host_nl_2p(dtype, _p(v), _p(R9), _p(t3))
        Ro, to = oracle.nl_2p(*v, is_f64=f64)
        assert np.array_equal(R9.reshape(3, 3), Ro) and np.array_equal(t3, to)
    valid = ~np.isnan(a["full_P"]).all(1)
    xw, xc = _cm(a["full_Q"][valid][:3]), _cm(a["full_P"][valid][:3])
    R9, t3 = np.zeros(9), np.zeros(3)
    L.lib().rpe_host_shinji(dtype, _p(xw), _p(xc), 3, _p(R9), _p(t3))
    k = G["kabsch"]["first3"]
    assert util.rot_err(R9.reshape(3, 3), np.array(k["R"])) < (1e-9 if f64 else 1e-4)
    Ro, to, _ = oracle.shinji(xw, xc, 3, f64)
    assert np.array_equal(R9.reshape(3, 3), Ro) and np.array_equal(t3, to)


@pytest.mark.parametrize("f64", [False, True])
def test_host_minimal_solvers_bit_exact_on_noisy_samples(oracle, f64):
    """1000 noisy 4-point samples (with 10 % gross outliers) per dtype: shinji(K = 3), kneip_main, kneip, nl_2p product == oracle, every bit."""
    dtype = L.F64 if f64 else L.F32
    dt = np.float64 if f64 else np.float32
    sc = util.scene_full(4242, 4000, dt, n2d=3.0, n3d=0.05, nnl_deg=2.0, outliers=0.1)
    for i in range(0, 4000, 4):
        Q, U, P, M, N = (_cm(x[i:i + 4]) for x in (sc.Q, sc.U, sc.P, sc.M, sc.N))
        R9, t3 = np.zeros(9), np.zeros(3)
        L.lib().rpe_host_shinji(dtype, _p(Q), _p(P), 3, _p(R9), _p(t3))
        Ro, to, _ = oracle.shinji(Q[:3], P[:3], 3, f64)
        assert np.array_equal(R9.reshape(3, 3), Ro) and np.array_equal(t3, to), i
        sols = np.zeros((4, 12))
        cnt = L.lib().rpe_host_kneip_main(dtype, _p(Q), _p(U), _p(sols))
        ref = oracle.kneip_main(Q, U, f64)
        assert cnt == len(ref), i
        for k in range(cnt):
            assert np.array_equal(sols[k, :9].reshape(3, 3), ref[k][0]) and np.array_equal(sols[k, 9:], ref[k][1]), (i, k)
        v = _cm(np.stack([P[0], N[0], P[1], Q[0], M[0], Q[1]]))
        L.lib().rpe_host_nl_2p(dtype, _p(v), _p(R9), _p(t3))
        Ro, to = oracle.nl_2p(*v, is_f64=f64)
        assert np.array_equal(R9.reshape(3, 3), Ro, equal_nan=True) and np.array_equal(t3, to, equal_nan=True), i


def test_host_error_metrics(G):
    e = G["calc_err"]
    a = [np.array(x, np.float64) for x in (e["gt"]["R"], e["gt"]["t"], e["se"]["R"], e["se"]["t"])]
    err, pct = np.zeros(2), np.zeros(2)
    L.lib().rpe_host_calc_err(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(err), _p(pct))
    assert abs(err[0] - e["te_re"][0]) < 1e-12 and abs(err[1] - e["te_re"][1]) < 1e-12
    assert abs(pct[0] - G["calc_percentage_err"]["te"]) < 1e-9 and abs(pct[1] - G["calc_percentage_err"]["re_sign_aligned"]) < 1e-9


def test_pose7_helper_matches_oracle(oracle):
    rng = np.random.default_rng(3)
    from rgbd_pose_estimation_amd import simulator as S
    for _ in range(50):
        R, t = S.random_pose(rng)
        for dt, f64 in ((L.F32, False), (L.F64, True)):
            assert np.array_equal(api.pose7_from_Rt(R, t, dt), oracle.pose7_from_Rt(R, t, f64))
    R = np.diag([-1.0, -1.0, 1.0])  # trace <= 0 branch
    assert np.array_equal(api.pose7_from_Rt(R, np.zeros(3), L.F64), oracle.pose7_from_Rt(R, np.zeros(3), True))


def test_sqrt_cut_is_the_exact_threshold_of_the_rooted_test():
    """The exact 3D vote is sqrt(s) < thr (Eigen norm(), reference AbsoluteOrientation.hpp:137-138); the kernels evaluate s < cut.
    numpy's sqrt is the correctly rounded one: the two predicates must agree for every s, in particular in the ulps around the cut."""
    lib = L.lib()
    rng = np.random.default_rng(0)
    for dt, code in ((np.float32, L.F32), (np.float64, L.F64)):
        thrs = np.concatenate([np.array([0.2, 0.1, 1.0, 3.0, 1e-3, 1e-20, 1e18, 0.05, 2.0 ** -10], dt), rng.uniform(1e-4, 50.0, 300).astype(dt)])
        for thr in thrs:
            cut = dt(lib.rpe_host_sqrt_cut(code, float(thr)))
            s = cut
            near = [s]
            for _ in range(40):
                s = np.nextafter(s, dt(0)); near.append(s)
            s = cut
            for _ in range(40):
                s = np.nextafter(s, dt(np.inf)); near.append(s)
            near = np.array(near + list(rng.uniform(0, 4 * float(thr) ** 2, 200).astype(dt)) + [dt(0), dt(np.inf), dt(np.nan)], dt)
            with np.errstate(invalid="ignore"):
                assert np.array_equal(np.sqrt(near) < thr, near < cut), (dt, thr, cut)
    # degenerate thresholds: nothing passes for thr <= 0 or NaN; every finite s passes for thr = inf
    assert lib.rpe_host_sqrt_cut(L.F32, 0.0) == 0.0 and lib.rpe_host_sqrt_cut(L.F32, -1.0) == 0.0
    assert np.isnan(lib.rpe_host_sqrt_cut(L.F64, float("nan"))) and lib.rpe_host_sqrt_cut(L.F64, float("inf")) == float("inf")


def test_run_replay_refuses_a_malformed_index_array():
    """first[i] .. first[i + 1] index the hypothesis array: a negative or decreasing entry would make the copy run wild; refused before
    anything else happens (no GPU needed to see it)."""
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    xw = np.random.default_rng(0).normal(size=(50, 3)).astype(np.float32)
    poses = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (4, 1))
    for first in ([0, 2, 1, 4], [1, 2, 3, 4], [0, -1, 2, 4]):
        with pytest.raises(L.RpeError) as e:
            api.run_replay(0, poses, first, xw=xw, xc=xw, thre_3d=0.1, iters=3)
        assert "first[" in str(e.value), str(e.value)


def test_hypothesis_capture_is_per_thread():
    """rpe_host_hypotheses captures on the calling thread only: two threads capturing different streams at once get their own streams."""
    import threading
    import numpy as np
    from rgbd_pose_estimation_amd import api
    rng = np.random.default_rng(3)
    xw = rng.normal(size=(400, 3)).astype(np.float32)
    xc = (xw + 0.01 * rng.normal(size=xw.shape)).astype(np.float32)
    want = {seed: api.host_hypotheses(0, xw=xw, xc=xc, iters=40, seed=seed) for seed in (11, 12)}
    got = {}

    def work(seed):
        for _ in range(20):
            q, first = api.host_hypotheses(0, xw=xw, xc=xc, iters=40, seed=seed)
            if not (np.array_equal(q, want[seed][0]) and np.array_equal(first, want[seed][1])):
                got[seed] = False
                return
        got[seed] = True
    th = [threading.Thread(target=work, args=(s,)) for s in (11, 12)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == {11: True, 12: True}
