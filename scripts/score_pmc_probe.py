"""K4 by itself: H hypotheses scored over n correspondences, `launches` calls per (kind, mode), for rocprofv3 --kernel-trace / --pmc
passes on the scoring kernel.  Prints one JSON line of wall times per call (upload of the poses + kernel + vote read-out).
usage: score_pmc_probe.py [n] [H] [launches]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rgbd_pose_estimation_amd import _lib as L, api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 20
R, t, Q, P, N = bench.cheap_scene(n)
U = (Q @ R.T + t).astype(np.float32)
U /= np.linalg.norm(U, axis=1, keepdims=True)
rng = np.random.default_rng(5)
q0 = api.pose7_from_Rt(R, t, L.F32)
poses = np.tile(np.asarray(q0, np.float64), (H, 1))
poses[:, :4] += 0.01 * rng.standard_normal((H, 4)); poses[:, :4] /= np.linalg.norm(poses[:, :4], axis=1, keepdims=True)
poses[:, 4:] += 0.05 * rng.standard_normal((H, 3))
ctx = api.Context(0).load(L.F32, xw=Q, xc=P, nc=N, bv=U, nw=(N @ R).astype(np.float32))
cos_thr = float(np.cos(np.arctan(8.0 / 585.0)))
out = {"n": n, "H": H}
for name, kind in (("v33", L.VOTE_33), ("v33_23", L.VOTE_33_23), ("vnn_33_23", L.VOTE_NN_33_23)):
    for mname, mode in (("fast", L.SCORE_FAST), ("exact", L.SCORE_EXACT)):
        for _ in range(3):
            v = ctx.score(kind, poses, thre_3d=0.2, cos_thr=cos_thr, cos_nl=0.95, mode=mode)
        t0 = time.perf_counter()
        for _ in range(launches):
            v = ctx.score(kind, poses, thre_3d=0.2, cos_thr=cos_thr, cos_nl=0.95, mode=mode)
        dt = (time.perf_counter() - t0) / launches
        out[f"{name}_{mname}"] = {"us_per_call": round(dt * 1e6, 2), "corr_hyp_per_s": n * H / dt, "max_votes": int(v.max())}
ctx.close()
print(json.dumps(out))
