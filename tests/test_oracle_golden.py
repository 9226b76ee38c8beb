"""CPU: pin the oracle (oracle/*.hpp, the restatement of the reference) against the independent numpy/scipy fixtures of
tests/golden/make_golden.py and against analytic known-answer cases.  No GPU, no /root/reference."""
import json
import math
import os

import numpy as np
import pytest

import util

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    g = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    g["arr"] = dict(np.load(os.path.join(HERE, "golden", "golden_inputs.npz")))
    return g


def _scene(G):
    a = G["arr"]
    return dict(xw=a["full_Q"], xc=a["full_P"], bv=a["full_U"], nw=a["full_M"], nc=a["full_N"])


def test_rand31_is_pcg32(oracle, G):
    # reference vector of pcg32 (seed 42, stream 54) >> 1, then the seeded streams of the fixtures
    assert list(oracle.rand31(7, 16)) == G["rand31_seed7"]
    import ctypes as C
    lib = oracle.lib()
    # seed 42 through the same entry point
    got = oracle.rand31(42, 6)
    assert list(got) == [x >> 1 for x in G["pcg32_seed42_seq54"]]


def test_random_elements_bit_exact(oracle, G):
    assert oracle.random_elements(100, 4, 3, 20).tolist() == G["random_elements_n100_m4_seed3"]
    assert oracle.random_elements(5, 3, 3, 10).tolist() == G["random_elements_n5_m3_seed3"]
    d = oracle.random_elements(50, 5, 1, 200)
    assert all(len(set(r)) == 5 for r in d.tolist()) and d.min() >= 0 and d.max() < 50


def test_prosac_bit_exact(oracle, G):
    assert oracle.prosac_samples(True, 4, 100, 9, 300).tolist() == G["prosac_f64_m4_n100_seed9"]
    assert oracle.prosac_samples(False, 3, 50, 9, 300).tolist() == G["prosac_f32_m3_n50_seed9"]


def test_prosac_progresses_from_top_ranked(oracle):
    s = oracle.prosac_samples(True, 4, 1000, 5, 50)
    assert s[:5].max() < 10  # early samples come from the best few
    # run long enough that n reaches N: indices stay in range (the reference would read out of bounds)
    s = oracle.prosac_samples(True, 3, 12, 5, 30000)
    assert s.min() >= 0 and s.max() <= 11


def test_update_num_iters_table(oracle, G):
    for p, ep, mp, mx, want in G["update_num_iters_f64"]:
        assert oracle.ransac_update_num_iters(True, p, ep, mp, mx) == want, (p, ep, mp, mx)
    # float instantiation: same unless the float rounding of log() crosses .5 -- check a few hand values
    assert oracle.ransac_update_num_iters(False, 0.99, 0.5, 3, 1000) == 34
    assert oracle.ransac_update_num_iters(False, 0.99, 0.0, 3, 1000) == 0     # no outliers: denom = 1 - 1 < eps -> 0
    assert oracle.ransac_update_num_iters(False, 0.99, 1.0, 3, 1000) == 1000  # all outliers: log(denom) = log(1) = 0 >= 0 -> maxIters


def test_sort_indexes_descending(oracle):
    w = np.array([0.3, 2.0, -1.0, 2.5, 0.0])
    assert oracle.sort_indexes(w).tolist() == [3, 1, 0, 4, 2]


def test_svd3_and_quaternion(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        A = rng.standard_normal((3, 3))
        U, s, V = oracle.svd3(A)
        assert np.allclose(U @ np.diag(s) @ V.T, A, atol=1e-13)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-13) and np.allclose(V.T @ V, np.eye(3), atol=1e-13)
        assert np.allclose(s, np.linalg.svd(A)[1], atol=1e-13) and s[0] >= s[1] >= s[2]
    # rank deficient: full orthonormal bases anyway
    for A in (np.outer([1, 2, 3.0], [0.5, -1, 2.0]), np.zeros((3, 3)), np.diag([2.0, 1.0, 0.0])):
        U, s, V = oracle.svd3(A)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-12) and np.allclose(U @ np.diag(s) @ V.T, A, atol=1e-12)


@pytest.mark.parametrize("f64", [True, False])
def test_shinji_matches_kabsch(oracle, G, f64):
    a = G["arr"]
    valid = ~np.isnan(a["full_P"]).all(1)
    tolr, tolt = (1e-12, 1e-12) if f64 else (5e-6, 5e-6)
    R, t, rc = oracle.shinji(a["full_Q"][valid], a["full_P"][valid], is_f64=f64)
    k = G["kabsch"]["noisy_valid"]
    assert rc == 0 and util.rot_err(R, np.array(k["R"])) < tolr and util.trans_rel_err(t, k["t"]) < tolt
    R, t, rc = oracle.shinji(a["full_Q"][valid], a["full_P"][valid], K=3, is_f64=f64)
    k = G["kabsch"]["first3"]
    assert rc == 0 and util.rot_err(R, np.array(k["R"])) < (1e-9 if f64 else 1e-3)
    for name in ("pure_translation", "rot180_z", "rot180_axis", "planar", "mirror"):
        R, t, rc = oracle.shinji(a[f"kab_{name}_xw"], a[f"kab_{name}_xc"], is_f64=f64)
        k = G["kabsch"][name]
        assert rc == 0, name
        assert util.rot_err(R, np.array(k["R"])) < (1e-10 if f64 else 1e-5), name
        assert np.linalg.norm(t - np.array(k["t"])) < (1e-10 if f64 else 2e-5), name
        assert abs(np.linalg.det(R) - 1) < 1e-5


def test_ao_is_float_shinji_ls2_and_less_accurate_than_f64(oracle):
    """Library.cpp ao() restated: float arithmetic.  At 307200 points its own fp32 accumulation error is visible
    against the fp64 evaluation of the same data (SURVEY.md section 7 'Precision')."""
    sc = util.scene33(3, 307200, np.float32)
    Rf, tf = oracle.ao(sc.Q, sc.P)
    Rd, td, _ = oracle.shinji_f32in_f64(sc.Q, sc.P)
    e = util.rot_err(Rf.astype(np.float64), Rd)
    assert 1e-8 < e < 5e-3


def test_votes_match_numpy(oracle, G):
    a = G["arr"]
    prob = oracle.Problem(True, **_scene(G))
    thr3, cthr, cnl = G["vote_thresholds"]
    kinds = {"33": oracle.V_33, "23": oracle.V_23, "33_23": oracle.V_33_23, "nn_23": oracle.V_NN_23, "nn_33": oracle.V_NN_33,
             "nn_33_23": oracle.V_NN_33_23}
    for name, k in kinds.items():
        v = oracle.votes(prob, k, a["hyp_q7"], thr3, cthr, cnl)
        assert v.tolist() == G["votes"][name], name
    # matrix-product variant of the 2D vote (kneip_ransac) counts the same in double
    assert oracle.votes(prob, oracle.V_23_MATRIX, a["hyp_q7"], thr3, cthr, cnl).tolist() == G["votes"]["23"]
    # masks of the first (ground-truth) hypothesis
    v, m = oracle.votes(prob, oracle.V_NN_33_23, a["hyp_q7"][:1], thr3, cthr, cnl, mask_for=0)
    assert np.array_equal(m, a["mask_truth"])
    # NaN camera points: never a 3D or N-N inlier, but they still get the 2D vote (no validity test there)
    nan = np.isnan(a["full_P"]).all(1)
    assert m[1][nan].sum() == 0 and m[2][nan].sum() == 0 and m[0][nan].sum() > 0


def test_gn_normal_equations_match_numerical_jacobians(oracle, G):
    a = G["arr"]
    pose = oracle.pose12(G["gn_pose"]["R"], G["gn_pose"]["t"])
    sl = slice(0, 200)
    for kind, (x, b, c) in {0: ("full_Q", "full_P", None), 1: ("full_Q", "full_P", "full_N"), 2: ("full_Q", "full_U", None),
                            4: ("full_Q", "full_U", None)}.items():
        rec = oracle.gn_normal_eq(kind, a[x][sl], a[b][sl], None if c is None else a[c][sl], pose=pose, in_f64=True)
        H, g, cost, cnt = util.unpack_ne(rec)
        ref = G["gn_numeric_first200"][str(kind)]
        assert cnt == ref["count"]
        assert np.allclose(H, np.array(ref["H"]), rtol=1e-6, atol=1e-6 * np.abs(ref["H"]).max())
        assert np.allclose(g, np.array(ref["g"]), rtol=1e-6, atol=1e-7 * max(1.0, np.abs(ref["g"]).max()))
        assert abs(cost - ref["cost"]) < 1e-9 * max(1.0, ref["cost"])


def test_gn_converges_to_closed_form_and_truth(oracle):
    # F1: GN p2p == shinji on the same set; noise-free scene: every objective's minimiser is the true pose
    sc = util.scene_full(5, 2000, np.float64, n2d=0.0, n3d=0.0, nnl_deg=0.0, outliers=0.0)
    p0 = oracle.pose12(*util.perturbed_pose(np.random.default_rng(0), sc.R, sc.t, 0.05, 0.1))
    for terms in ([dict(kind=oracle.GN_P2P, a=sc.Q, b=sc.P)], [dict(kind=oracle.GN_P2PLANE, a=sc.Q, b=sc.P, c=sc.N)],
                  [dict(kind=oracle.GN_BEARING, a=sc.Q, b=sc.U)]):
        p, its, step, cost = oracle.gn_refine(terms, 2000, p0, max_iter=50, tol=1e-12, in_f64=True)
        assert 0 < its < 50 and util.rot_err(p[:9].reshape(3, 3), sc.R) < 1e-9 and np.linalg.norm(p[9:] - sc.t) < 1e-8
    sc = util.scene33(6, 5000, np.float64, outliers=0.0)
    Rk, tk, _ = oracle.shinji(sc.Q, sc.P)
    p, its, _, _ = oracle.gn_refine([dict(kind=oracle.GN_P2P, a=sc.Q, b=sc.P)], 5000, oracle.pose12(np.eye(3), np.zeros(3)), 40, 1e-12, True)
    assert its > 0 and util.rot_err(p[:9].reshape(3, 3), Rk) < 1e-10 and util.trans_rel_err(p[9:], tk) < 1e-10


def test_se3_exp_log(oracle, G):
    for e in G["se3_exp"]:
        R, t = oracle.se3_exp(e["a"])
        assert np.allclose(R, e["R"], atol=1e-13) and np.allclose(t, e["t"], atol=1e-13)
        if np.linalg.norm(e["a"][3:]) < 3.0:
            assert np.allclose(oracle.se3_log(R, t), e["a"], atol=1e-9)


def test_find_opt_cc_and_nl_ls(oracle, G):
    a = G["arr"]
    R0, t0 = np.array(G["gn_pose"]["R"]), np.array(G["gn_pose"]["t"])
    Pz = np.nan_to_num(a["full_P"])
    arrs = dict(_scene(G), xc=Pz)
    prob = oracle.Problem(True, **arrs)
    assert np.allclose(oracle.find_opt_cc(prob, R0, a["mask_truth"][0]), G["find_opt_cc"], atol=1e-9)
    # degenerate: fewer than two rays -> |det AA| < 1e-4 -> NaN
    one = np.zeros(1000, np.int16); one[3] = 1
    assert np.isnan(oracle.find_opt_cc(prob, R0, one)).all()
    for bug in (True, False):
        for weighted in (False, True):
            p = oracle.Problem(True, weights=a["full_W"] if weighted else None, **arrs)
            r = oracle.run(p, oracle.M_NONE, ls=oracle.LS_NL_BUGCOMPAT if bug else oracle.LS_NL_FIXED, mask_in=a["mask_truth"], pose_in=(R0, t0))
            ref = G["nl_shinji_kneip_ls"][f"bug{int(bug)}_w{int(weighted)}"]
            assert util.rot_err(r["R"], np.array(ref["R"])) < 1e-9, (bug, weighted)
            assert np.linalg.norm(r["t"] - np.array(ref["t"])) < 1e-8, (bug, weighted)
    # the two variants really differ (the accumulate-across-rounds quirk is observable)
    b = G["nl_shinji_kneip_ls"]
    assert util.rot_err(np.array(b["bug1_w0"]["R"]), np.array(b["bug0_w0"]["R"])) > 1e-7


def test_error_metrics(oracle, G):
    e = G["calc_err"]
    te, re = oracle.calc_err(e["gt"]["R"], e["gt"]["t"], e["se"]["R"], e["se"]["t"])
    assert abs(te - e["te_re"][0]) < 1e-12 and abs(re - e["te_re"][1]) < 1e-12
    pt, pr = oracle.calc_percentage_err(e["gt"]["R"], e["gt"]["t"], e["se"]["R"], e["se"]["t"])
    assert abs(pt - G["calc_percentage_err"]["te"]) < 1e-9 and abs(pr - G["calc_percentage_err"]["re_sign_aligned"]) < 1e-9


def test_quartic_roots(oracle, G):
    for q in G["quartics"]:
        assert np.allclose(np.sort(oracle.o4_roots(q["coeffs"])), q["roots"], atol=1e-7)


@pytest.mark.parametrize("f64", [True, False])
def test_p3p_contains_truth(oracle, G, f64):
    a = G["arr"]
    R, t = np.array(G["full_R"]), np.array(G["full_t"])
    tol = 1e-8 if f64 else 3e-2  # P3P in float is badly conditioned; the reference's demos run it in float all the same
    hits = 0
    for i in range(0, 40, 4):
        sols = oracle.kneip_main(a["p3p_Q"][i:i + 4], a["p3p_U"][i:i + 4], f64)
        assert 1 <= len(sols) <= 4
        assert min(util.rot_err(Rs, R) + np.linalg.norm(ts - t) for Rs, ts in sols) < tol
        k = oracle.kneip(a["p3p_Q"][i:i + 4], a["p3p_U"][i:i + 4], f64)
        hits += k is not None and util.rot_err(k[0], R) < tol
    assert hits >= 9
    # collinear world points: no solution
    Q = np.array([[0, 0, 1.0], [0, 0, 2.0], [0, 0, 3.0], [1, 1, 1.0]])
    assert oracle.kneip_main(Q, a["p3p_U"][:4], True) == []


def test_nl_2p_constraints(oracle, G):
    """The reference's 2-point+normal solver always maps normal to normal and point 1 to point 1; it is exact for the
    whole pose only when the in-plane rotation is counter-clockwise (acos loses the sign) -- both facts asserted."""
    a = G["arr"]
    R, t = np.array(G["full_R"]), np.array(G["full_t"])
    exact = 0
    for i in range(0, 40, 2):
        Rn, tn = oracle.nl_2p(a["p3p_P"][i], a["p3p_N"][i], a["p3p_P"][i + 1], a["p3p_Q"][i], a["p3p_M"][i], a["p3p_Q"][i + 1])
        assert np.allclose(Rn @ a["p3p_M"][i], a["p3p_N"][i], atol=1e-9)
        assert np.allclose(Rn @ a["p3p_Q"][i] + tn, a["p3p_P"][i], atol=1e-9)
        assert abs(np.linalg.det(Rn) - 1) < 1e-9
        e = util.rot_err(Rn, R)
        assert e < 1e-8 or e > 1e-4  # either the right branch or the mirrored in-plane angle, never "a bit off"
        exact += e < 1e-8
    assert 0 < exact <= 20


def test_pipelines_recover_pose(oracle):
    """Every RANSAC driver of the restatement, Parameters.yml-like settings, lands near the simulated truth."""
    sc = util.scene_full(11, 300, np.float64, n2d=2.0, n3d=0.05, nnl_deg=2.0, outliers=0.2)
    arrs = dict(xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    runs = [(oracle.M_SHINJI_RANSAC, ("xw", "xc", "bv"), oracle.LS_SHINJI_INLIERS), (oracle.M_SHINJI_RANSAC2, ("xw", "xc"), oracle.LS_SHINJI_INLIERS),
            (oracle.M_SHINJI_PROSAC, ("xw", "xc"), oracle.LS_SHINJI_INLIERS), (oracle.M_KNEIP_RANSAC, ("xw", "bv"), 0),
            (oracle.M_KNEIP_PROSAC, ("xw", "bv"), 0), (oracle.M_SK_RANSAC, ("xw", "xc", "bv"), oracle.LS_SHINJI_INLIERS),
            (oracle.M_SK_PROSAC, ("xw", "xc", "bv"), oracle.LS_SHINJI_INLIERS), (oracle.M_NL_KNEIP_RANSAC, tuple(arrs), 0),
            (oracle.M_NL_SHINJI_RANSAC, tuple(arrs), oracle.LS_NL_BUGCOMPAT), (oracle.M_NL_SK_RANSAC, tuple(arrs), oracle.LS_NL_FIXED)]
    for m, keys, ls in runs:
        r = oracle.run(oracle.Problem(True, weights=sc.weights, **{k: arrs[k] for k in keys}), m, thre_3d=0.2, thre_2d=8.0, thre_nl=0.1,
                       iters=300, confidence=0.99999, seed=2, ls=ls)
        assert r["max_votes"] > 0 and r["iters"] <= 300
        assert util.rot_err(r["R"], sc.R) < 0.05 and np.linalg.norm(r["t"] - sc.t) < 0.5, m


def test_lsq_pnp_is_the_sum_of_sines(oracle, G):
    """R1 lsq_pnp (reference P3P.hpp:472-502): the oracle's total against numpy's sum of |normalize(R Xw + t) x bv| in fp64 -- 1e-12
    for the double instantiation, the float one within its own accumulated rounding -- and term by term against getError(i)."""
    sc = _scene(G)
    xw, bv = sc["xw"], sc["bv"]
    rng = np.random.default_rng(5)
    R, t = util.perturbed_pose(rng, np.eye(3), np.array([0.1, -0.2, 0.3]), 0.3, 0.1)
    q7 = oracle.pose7_from_Rt(R, t, True)
    # (the quaternion the oracle rotates with, back to a matrix: exactly the rotation it applies)
    w, x, y, z = q7[:4]
    Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    p = xw.astype(np.float64) @ Rq.T + q7[4:]
    ph = p / np.linalg.norm(p, axis=1, keepdims=True)
    terms_np = np.linalg.norm(np.cross(ph, bv.astype(np.float64)), axis=1)
    tot64, tot64_again, terms = oracle.lsq_pnp(xw, bv, q7, True, with_terms=True)
    assert abs(tot64 - terms_np.sum()) <= 1e-12 * terms_np.sum() and abs(tot64_again - tot64) <= 1e-13 * tot64
    assert np.max(np.abs(terms - terms_np)) < 1e-13
    q7f = oracle.pose7_from_Rt(R, t, False)
    tot32, tot32_in_double, terms32 = oracle.lsq_pnp(xw, bv, q7f, False, with_terms=True)
    n = len(xw)
    assert abs(tot32 - terms_np.sum()) <= (n * 2.0 ** -24 + 1e-5) * terms_np.sum()
    assert abs(tot32_in_double - terms32.sum()) <= 1e-12 * terms32.sum() and np.max(np.abs(terms32 - terms_np)) < 2e-6
