"""Soak of the cross-workgroup protocols (development aid, run on the GPU box): the host-driven resident loop, the collecting launches and
the one-launch device loop (granules + double-buffered run records, no host in the loop) must give BITWISE the same result every time (fixed summation order whichever workgroup finishes first) and never lose a granule.
  python scripts/soak.py [seconds]      (total: every leg runs for a fifteenth of it)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
from tail_timeline import scene

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
out = {}
for n, kind in ((307200, 0), (1000000, 1), (20000, 0)):
    R, t, arrs = scene(n)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    first_pose = ctx.gn_refine([kind], p, max_iter=500, tol=0.0)[0]
    first_rec = np.asarray(ctx.normal_eq(kind, p)[0] if isinstance(ctx.normal_eq(kind, p), tuple) else ctx.normal_eq(kind, p))
    t0, calls, iters, bad = time.perf_counter(), 0, 0, 0
    while time.perf_counter() - t0 < seconds / 15:
        q = ctx.gn_refine([kind], p, max_iter=500, tol=0.0)[0]
        bad += int(not np.array_equal(q, first_pose)); calls += 1; iters += 500
    t1, ncalls, nbad = time.perf_counter(), 0, 0
    while time.perf_counter() - t1 < seconds / 15:
        r = ctx.normal_eq(kind, p)
        r = np.asarray(r[0] if isinstance(r, tuple) else r)
        nbad += int(not np.array_equal(r, first_rec)); ncalls += 1
    first_dev = ctx.gn_refine_device([(kind, 1.0)], p, 0, 500, 0.0)[0]
    t2, dcalls, dbad = time.perf_counter(), 0, 0
    while time.perf_counter() - t2 < seconds / 15:
        q = ctx.gn_refine_device([(kind, 1.0)], p, 0, 500, 0.0)[0]
        dbad += int(not np.array_equal(q, first_dev)); dcalls += 1
    out[f"{n}_{kind}"] = dict(state=ctx.resident_state(), resident_calls=calls, resident_iterations=iters, resident_pose_changed=bad, collect_calls=ncalls, collect_record_changed=nbad,
                              device_loop_calls=dcalls, device_loop_iterations=500 * dcalls, device_loop_pose_changed=dbad)
    ctx.close()
print(json.dumps(out), flush=True)

# ---- round 3: the resident forms of the bearing kind and of the joint kernel, and the restructured one-launch device loop on them
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
out3 = {}
for n in (307200, 20000):
    sc = util.scene_full(900 + n, n, np.float32, n2d=2.0, n3d=0.03, nan_frac=0.02)
    p = api.pose12(*util.perturbed_pose(np.random.default_rng(n), sc.R, sc.t, 0.01, 0.03))
    ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    terms = [(L.RES_P2P, 1.0, 0, 1.0), (L.RES_BEARING, 4.0, L.ROBUST_HUBER, 0.01), (L.RES_NORMAL, 0.5, 0, 1.0)]
    legs = {"bearing_resident": lambda: ctx.gn_refine([L.RES_BEARING], p, max_iter=300, tol=0.0)[0],
            "joint_resident": lambda: ctx.gn_refine_joint(terms, p, max_iter=300, tol=0.0)[0],
            "bearing_device_loop": lambda: ctx.gn_refine_device([(L.RES_BEARING, 1.0)], p, 0, 300, 0.0)[0]}
    for name, f in legs.items():
        first = f()
        t0, calls, bad = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < seconds / 15:
            bad += int(not np.array_equal(f(), first)); calls += 1
        out3[f"{name}_{n}"] = dict(calls=calls, iterations=300 * calls, pose_changed=bad)
    out3[f"lost_grids_{n}"] = ctx.resident_state()["lost"]
    ctx.close()
print(json.dumps({"round3": out3}), flush=True)

# ---- round 5: the resident scoring session (batches through the control block, the winner's masks as its last message, verified at
# the next call) and the autonomous loops that hand their sums to the solving workgroup (single kind and joint)
out5 = {}
for n in (307200, 20000):
    sc = util.scene_full(1900 + n, n, np.float32, n2d=2.0, n3d=0.03, nan_frac=0.02, outliers=0.2)
    p = api.pose12(*util.perturbed_pose(np.random.default_rng(n), sc.R, sc.t, 0.01, 0.03))
    ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    rng = np.random.default_rng(5)
    q = np.tile(api.pose7_from_Rt(sc.R, sc.t), (24, 1))
    q[1:, 4:] += 0.02 * rng.standard_normal((23, 3))
    q = np.ascontiguousarray(q.astype(np.float32).astype(np.float64))
    thr = dict(thre_3d=0.05, cos_thr=float(np.cos(np.arctan(4.0 / 585.0))), cos_nl=2.0)
    want_votes = ctx.score(L.VOTE_33_23, q, **thr)
    ctx.inlier_mask(L.VOTE_33_23, q[3], **thr)
    want_masks = [ctx.download_mask(L.MOD_23).copy(), ctx.download_mask(L.MOD_33).copy()]
    t0, runs, bad = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < seconds / 15:
        assert ctx.score_session_begin(L.VOTE_33_23, **thr)
        v = np.concatenate([ctx.score(L.VOTE_33_23, q[:8], **thr), ctx.score(L.VOTE_33_23, q[8:], **thr)])
        ctx.inlier_mask(L.VOTE_33_23, q[3], **thr)            # not waited for
        bad += int(not np.array_equal(v, want_votes))
        if runs % 16 == 0:
            bad += int(not (np.array_equal(ctx.download_mask(L.MOD_23), want_masks[0]) and np.array_equal(ctx.download_mask(L.MOD_33), want_masks[1])))
        runs += 1
    out5[f"score_session_{n}"] = dict(runs=runs, changed=bad)
    terms2 = [(L.RES_P2P, 1.0, 0, 1.0), (L.RES_BEARING, 4.0, 0, 1.0)]
    legs = {"p2p_device_loop_solver": lambda: ctx.gn_refine_device([(L.RES_P2P, 1.0)], p, 0, 300, 0.0)[0],
            "joint_device_loop_solver": lambda: ctx.gn_refine_device(terms2, p, 0, 300, 0.0)[0]}
    for name, f in legs.items():
        first = f()
        t0, calls, bad = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < seconds / 15:
            bad += int(not np.array_equal(f(), first)); calls += 1
        out5[f"{name}_{n}"] = dict(calls=calls, iterations=300 * calls, pose_changed=bad)
    out5[f"lost_grids_{n}"] = ctx.resident_state()["lost"]
    out5[f"solver_still_on_{n}"] = ctx.resident_state()["solver"]
    ctx.close()
print(json.dumps({"round5": out5}), flush=True)
