"""Host-loop vs device-resident Gauss-Newton: time per iteration (development aid; numbers quoted in DESIGN.md)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
for n in (307200, 1000000):
    sc = S.dense_depth_scene(1, n)
    ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
    ctx.inlier_mask(L.VOTE_33, api.pose7_from_Rt(sc.R, sc.t), thre_3d=0.2)
    p0 = api.pose12(sc.R, sc.t); p0[9] += 0.05
    K = 200
    for mode, f in (("host loop (kernel + publish + host solve)", lambda: ctx.gn_refine_joint([(L.RES_P2P, 1.0)], p0, L.USE_MASK, K, 0.0)),
                    ("host loop, dedicated p2p kernel", lambda: ctx.gn_refine([L.RES_P2P], p0, None, L.USE_MASK, K, 0.0)),
                    ("device-resident (solve + exp-map in the kernel)", lambda: ctx.gn_refine_device([(L.RES_P2P, 1.0)], p0, L.USE_MASK, K, 0.0))):
        f()
        t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
        print(f"n={n} {mode}: {dt / K * 1e6:.2f} us/iteration  {n * K / dt:.3e} corr-res/s  (iters {r[1]})")
