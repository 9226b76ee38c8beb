"""A/B of builds for K4 scoring on one box: wall time per call of rpe_score over 512 hypotheses x 307 200 correspondences, EXACT and
FAST, by vote kind.  usage: score_ab.py tag=lib.so [tag=lib.so ...]   (alternates twice).  RPE_AB_SCENE: harsh (default: thre_3d 0.05
on 3D noise 0.03 -- the threshold sits in the middle of the inliers' residuals; a bearing for every correspondence), benign
(Parameters.yml: thre_3d 0.2 on noise 0.05), sparse (benign, bearings for the first 2 000 correspondences only: configs[2])."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(tag):
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    import util
    n, H = 307200, 512
    scene = os.environ.get("RPE_AB_SCENE", "harsh")
    thre_3d = 0.05 if scene == "harsh" else 0.2
    sc = util.scene_full(5, n, np.float32, n2d=2.0, n3d=0.03 if scene == "harsh" else 0.05, nnl_deg=2.0, outliers=0.2)
    if scene == "sparse":
        sc.U[2000:] = np.nan
    ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    rng = np.random.default_rng(1)
    q = np.tile(api.pose7_from_Rt(sc.R, sc.t, L.F32), (H, 1))
    q[1:, :4] += 0.002 * rng.standard_normal((H - 1, 4)); q[:, :4] /= np.linalg.norm(q[:, :4], axis=1, keepdims=True)
    q[1:, 4:] += 0.02 * rng.standard_normal((H - 1, 3))
    cos_thr = float(np.cos(np.arctan(8.0 / 585.0)))
    for kind, name in ((L.VOTE_33, "33"), (L.VOTE_23, "23"), (L.VOTE_33_23, "33_23"), (L.VOTE_NN_33, "nn_33"), (L.VOTE_NN_33_23, "nn_33_23")):
        for mode, mname in ((L.SCORE_EXACT, "exact"), (L.SCORE_FAST, "fast")):
            for _ in range(3):
                v = ctx.score(kind, q, thre_3d, cos_thr, 0.999, mode=mode)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter(); v = ctx.score(kind, q, thre_3d, cos_thr, 0.999, mode=mode); ts.append(time.perf_counter() - t0)
            ts.sort()
            print(json.dumps(dict(tag=tag, scene=scene, kind=name, mode=mname, median_us=round(ts[len(ts) // 2] * 1e6, 1), min_us=round(ts[0] * 1e6, 1), votes0=int(v[0]), votes_sum=int(v.sum()))), flush=True)
    ctx.close()


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2])
    else:
        for rep in range(2):
            for spec in sys.argv[1:]:
                tag, path = spec.split("=", 1)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", tag + "#%d" % rep], env=dict(os.environ, RPE_LIBRARY=os.path.abspath(path)), check=False)
