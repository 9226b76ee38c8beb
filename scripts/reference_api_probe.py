"""The kernels behind the reference's own least-squares API by themselves, for rocprofv3 --kernel-trace --stats and the --pmc FETCH_SIZE /
WRITE_SIZE passes: K1' moments_kernel (with the 3D-3D inlier mask: shinji_ls / shinji_ls1), K5 nl_round_full_kernel and K4b mask_kernel
of the 3D-3D vote, 30 steady launches each at 307 200 / 1 M / 10 M correspondences.  One size per process (RPE_PROBE_N) so that a kernel
name's rows in the profiler's statistics belong to ONE size.  Prints the event-timed averages as one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rgbd_pose_estimation_amd import _lib as L, api

n = int(os.environ.get("RPE_PROBE_N", "1000000"))
R, t, Q, P, Nc = bench.cheap_scene(n)
bv = (P / np.linalg.norm(P, axis=1, keepdims=True)).astype(np.float32)
Nw = np.ascontiguousarray((Nc.astype(np.float64) @ R).astype(np.float32))
rng = np.random.default_rng(11)
ctx = api.Context(0).load(L.F32, xw=Q, xc=P, bv=bv, nw=Nw, nc=Nc)
for m in range(3):
    ctx.upload_mask(m, (rng.random(n) < 0.8).astype(np.int16))
q7 = api.pose7_from_Rt(R, t, L.F32)
Rwc, c_opt = R.T, -(R.T @ t)
Cw, Cc = Q[:1000].mean(0), P[:1000].mean(0)
out = {"n": n}
for name, bpc, fn in (("K1p_moments", 26, lambda: ctx.p2p_moments(L.USE_MASK)), ("K5_nl_round", 66, lambda: ctx.nl_round(c_opt, Cw, Cc, Rwc)),
                      ("K4b_mask_33", 26, lambda: ctx.inlier_mask(L.VOTE_33, q7, thre_3d=0.2))):
    for _ in range(5):
        fn()
    ctx.timing_enable(30, 1)
    for _ in range(30):
        fn()
    cnt, tot, mn = ctx.timing_collect()
    ctx.timing_enable(0, 1)
    out[name] = {"launches": cnt, "avg_launch_us": tot / cnt * 1e3, "algorithmic_bytes": bpc * n, "achieved_GBs": bpc * n / (tot / cnt * 1e-3) / 1e9}
ctx.close()
print(json.dumps(out))
