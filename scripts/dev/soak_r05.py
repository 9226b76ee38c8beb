"""Long soak of this round's protocols only: scoring sessions (lazy masks, sessions of two threads taking turns) and the autonomous
loops with a solving workgroup, frame-sized and beyond (the 224-worker cap).  usage: soak_r05.py [seconds]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
import util
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
out = {}
# (1) two threads, a context each: RANSAC-shaped runs through sessions + a masked refinement now and then
sc = util.scene_full(77, 307200, np.float32, n2d=2.0, n3d=0.03, nan_frac=0.02, outliers=0.2)
rng = np.random.default_rng(5)
q = np.tile(api.pose7_from_Rt(sc.R, sc.t), (24, 1)); q[1:, 4:] += 0.02 * rng.standard_normal((23, 3))
q = np.ascontiguousarray(q.astype(np.float32).astype(np.float64))
thr = dict(thre_3d=0.05, cos_thr=float(np.cos(np.arctan(4.0 / 585.0))), cos_nl=2.0)
ctxs = [api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N) for _ in range(2)]
want = ctxs[0].score(L.VOTE_33_23, q, **thr)
ctxs[0].inlier_mask(L.VOTE_33_23, q[3], **thr)
wm = [ctxs[0].download_mask(L.MOD_23).copy(), ctxs[0].download_mask(L.MOD_33).copy()]
pose = api.pose12(*util.perturbed_pose(np.random.default_rng(1), sc.R, sc.t, 0.01, 0.03))
ref = ctxs[0].gn_refine([L.RES_P2P], pose, flags=L.USE_MASK, max_iter=5)[0]
stats = [dict(runs=0, bad=0) for _ in ctxs]
def worker(k):
    c, s, t0 = ctxs[k], stats[k], time.perf_counter()
    while time.perf_counter() - t0 < seconds / 3:
        assert c.score_session_begin(L.VOTE_33_23, **thr)
        v = np.concatenate([c.score(L.VOTE_33_23, q[:8], **thr), c.score(L.VOTE_33_23, q[8:], **thr)])
        c.inlier_mask(L.VOTE_33_23, q[3], **thr)
        s["bad"] += int(not np.array_equal(v, want)); s["runs"] += 1
        if s["runs"] % 8 == k:
            s["bad"] += int(not (np.array_equal(c.download_mask(L.MOD_23), wm[0]) and np.array_equal(c.download_mask(L.MOD_33), wm[1])))
            s["bad"] += int(not np.array_equal(c.gn_refine([L.RES_P2P], pose, flags=L.USE_MASK, max_iter=5)[0], ref))
ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
out["sessions_two_threads"] = dict(threads=stats, states=[c.resident_state() for c in ctxs])
for c in ctxs: c.close()
print(json.dumps(out), flush=True)
# (2) autonomous loops with the solving workgroup beyond a frame: 1.5 M and 3 M correspondences, point-to-plane and point-to-point, masked
out2 = {}
for n in (1_500_000, 3_000_000):
    sc = util.scene_full(900 + n % 13, n, np.float32, n2d=2.0, n3d=0.03, outliers=0.1)
    ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, nc=sc.N, nw=sc.M)
    mask = (np.random.default_rng(n).uniform(size=n) < 0.8).astype(np.int16); ctx.upload_mask(L.MOD_33, mask)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(2), sc.R, sc.t, 0.02, 0.05))
    for kind, name in ((L.RES_P2PLANE, "p2plane"), (L.RES_P2P, "p2p")):
        ctx.gn_refine([kind], p0, None, L.USE_MASK, 3, 0.0)
        first = ctx.gn_refine_device([(kind, 1.0)], p0, L.USE_MASK, 100, 0.0)[0]
        t0, calls, bad, slow = time.perf_counter(), 0, 0, 0
        while time.perf_counter() - t0 < seconds / 6:
            t1 = time.perf_counter()
            p = ctx.gn_refine_device([(kind, 1.0)], p0, L.USE_MASK, 100, 0.0)[0]
            slow += int(time.perf_counter() - t1 > 0.2); bad += int(not np.array_equal(p, first)); calls += 1
        out2[f"{name}_{n}"] = dict(calls=calls, iterations=100 * calls, pose_changed=bad, lost_loops=slow, state=ctx.resident_state())
    ctx.close()
print(json.dumps({"solver_beyond_a_frame": out2}), flush=True)
