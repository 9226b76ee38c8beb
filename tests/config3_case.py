"""configs[2] of BASELINE.json exactly as SURVEY.md section 8(d) "Config 3" defines it (after SimpleMain.cpp:213-226 and
AbsoluteOrientation.hpp:367-438): N = 307 200 3D-3D correspondences of which the first 2 000 also carry a bearing (NaN elsewhere),
sigma_3d = 0.05 m, sigma_2d = 15 px, outlier ratio 0.1 (Parameters.yml:5-9), thre_2d = 8 px, thre_3d = 0.2 m (:15-17), f = 585,
Iter = 300 (:13) iterations of shinji(3 points) + kneip generated ON THE HOST from a fixed seeded sample list, scored in batches,
replayed with the reference's best-so-far / adaptive-Iter semantics, then a joint Gauss-Newton refinement (3D-3D + 2D-3D terms) on
the inliers.  CPU restatement (oracle) and GPU consume the SAME hypothesis list: both generate it (and the two lists must be
identical), and each side is fed the OTHER side's list.

TEST / BENCH INFRASTRUCTURE (it drives the oracle): used by tests/test_gpu_fullsize.py and by bench.py's `config3_pipeline` extra."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

N, N2D, ITERS, SEED = 307200, 2000, 300, 20260103
THRE_3D, THRE_2D, F, CONF = 0.2, 8.0, 585.0, 0.99


def scene():
    from rgbd_pose_estimation_amd import simulator as S
    rng = np.random.default_rng(SEED)
    R, t = S.random_pose(rng)
    sc = S.simulate_2d_3d_3d_correspondences(rng, R, t, N, 15.0, 0.05, 0.1).astype(np.float32)
    U = sc.U.copy()
    U[N2D:] = np.nan   # F5: one adapter, one N -- a row without an image measurement has no bearing
    return sc, U


def run(with_cpu=True, f64=False):
    """Returns a dict: equality flags (hypothesis lists, votes, Iter, masks), pose distances, GPU and CPU wall times."""
    from rgbd_pose_estimation_amd import _lib as L, api
    import oracle_lib as O
    import util
    sc, U = scene()
    dtp = L.F64 if f64 else L.F32
    arrs = dict(xw=sc.Q, xc=sc.P, bv=U)
    out = {"n": N, "bearings": N2D, "iterations": ITERS, "dtype": "f64" if f64 else "f32"}
    # ---- the hypothesis list: the product's host generator (no GPU) ...
    t0 = time.perf_counter()
    q7, first = api.host_hypotheses(api.M_SK_RANSAC, dtp, iters=ITERS, seed=SEED, **arrs)
    out["generate_ms_product_host"] = (time.perf_counter() - t0) * 1e3
    out["hypotheses"] = int(len(q7))
    kw = dict(thre_3d=THRE_3D, thre_2d=THRE_2D, iters=ITERS, confidence=CONF)
    # ---- GPU: replay (scoring in batches on the GPU, best-so-far / adaptive Iter on the host), then the joint GN refinement
    api.run_replay(api.M_SK_RANSAC, q7, first, dtp, f=F, ls=api.LS_GN_JOINT, **arrs, **kw)   # warm (uploads, code objects, both stages)
    t0 = time.perf_counter()
    got = api.run_replay(api.M_SK_RANSAC, q7, first, dtp, f=F, ls=api.LS_NONE, **arrs, **kw)
    out["gpu_replay_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ref_gpu = api.run_replay(api.M_SK_RANSAC, q7, first, dtp, f=F, ls=api.LS_GN_JOINT, **arrs, **kw)
    out["gpu_replay_plus_refine_ms"] = (time.perf_counter() - t0) * 1e3
    out["gpu"] = {"max_votes": got["max_votes"], "iters": got["iters"], "inliers_23": int(got["masks"][0].sum()), "inliers_33": int(got["masks"][1].sum()),
                  "rot_err_rad_vs_truth": util.rot_err(ref_gpu["R"], sc.R), "trans_err_m_vs_truth": float(np.linalg.norm(ref_gpu["t"] - sc.t))}
    if not with_cpu:
        return out
    # ---- ... and the oracle's own list from the same seed: must be the same list
    prob = O.Problem(f64, f=F, **arrs)
    t0 = time.perf_counter()
    q7o, firsto = O.hypotheses(prob, O.M_SK_RANSAC, ITERS, seed=SEED)
    out["generate_ms_oracle"] = (time.perf_counter() - t0) * 1e3
    out["lists_identical"] = bool(np.array_equal(first, firsto) and np.array_equal(q7, q7o, equal_nan=True))
    # ---- CPU: the oracle replays the list (the reference's vote loop, 1 thread)
    t0 = time.perf_counter()
    ref = O.run_replay(prob, O.M_SK_RANSAC, q7o, firsto, **kw)
    out["cpu_replay_ms"] = (time.perf_counter() - t0) * 1e3
    # the GPU fed with the ORACLE's list (identical anyway, but this is the cross-feed 8d asks for)
    got_o = api.run_replay(api.M_SK_RANSAC, q7o, firsto, dtp, f=F, ls=api.LS_NONE, **arrs, **kw)
    out["votes_equal"] = bool(got["max_votes"] == ref["max_votes"] == got_o["max_votes"])
    out["iter_equal"] = bool(got["iters"] == ref["iters"] == got_o["iters"])
    out["masks_equal"] = bool(np.array_equal(got["masks"], ref["masks"]) and np.array_equal(got_o["masks"], ref["masks"]))
    out["winner_equal"] = bool(np.array_equal(got["R"], ref["R"]) and np.array_equal(got["t"], ref["t"]))
    out["cpu"] = {"max_votes": ref["max_votes"], "iters": ref["iters"]}
    # ---- joint refinement: oracle fp64 GN of the same objective on the same inliers from the same winner
    p0 = O.pose12(np.asarray(ref["R"]), np.asarray(ref["t"]))
    t0 = time.perf_counter()
    po, itso, _, _ = O.gn_refine([dict(kind=O.GN_P2P, a=sc.Q, b=sc.P, mask=ref["masks"][1]), dict(kind=O.GN_BEARING, a=sc.Q, b=U, mask=ref["masks"][0])],
                                 N, p0, max_iter=20, tol=1e-9, in_f64=f64)
    out["cpu_refine_ms"] = (time.perf_counter() - t0) * 1e3
    out["refine_iters"] = {"gpu": ref_gpu["iters"], "cpu": itso}
    out["refined_pose_vs_cpu"] = {"rot_rad": util.rot_err(ref_gpu["R"], po[:9].reshape(3, 3)), "trans_rel": util.trans_rel_err(ref_gpu["t"], po[9:]),
                                  "tolerance": {"rot_rad": 1e-5, "trans_rel": 1e-4}}
    out["speedup_replay"] = out["cpu_replay_ms"] / out["gpu_replay_ms"]
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run()))
