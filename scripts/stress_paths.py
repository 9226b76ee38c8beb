"""Randomised cross-check of the collecting / resident paths against the arrival-counter / launch-per-step paths (RPE_COLLECT=0,
RPE_RESIDENT=0, RPE_DEVICE_LOOP_RESIDENT=0): sizes around every geometry boundary, fp32 / fp64, all three residual kinds, with and without masks -- records to
rounding, votes exactly, refined poses to 1e-7.  Development aid (run on the GPU box)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np

def worker(seed0, count):
    from rgbd_pose_estimation_amd import _lib as L, api
    import util
    rng = np.random.default_rng(seed0)
    out = []
    for case in range(count):
        n = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 255, 1023, 4097, 20000, 76799, 76800, 76801, 131072 * 4 - 1, 131072 * 4 + 5, 300000, 1048577, 2500001]))
        if rng.random() < 0.5:
            n = int(rng.integers(1, 400000))
        f64 = bool(rng.random() < 0.3)
        dt = np.float64 if f64 else np.float32
        sc = util.scene_full(int(rng.integers(1, 1 << 30)), min(n, 200000), dt, nan_frac=0.05 if n > 20 else 0.0)
        reps = (n + len(sc.Q) - 1) // len(sc.Q)
        tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
        arrs = dict(xw=tile(sc.Q), xc=tile(sc.P), bv=tile(sc.U), nw=tile(sc.M), nc=tile(sc.N))
        ctx = api.Context(0).load(L.F64 if f64 else L.F32, **arrs)
        R, t = util.perturbed_pose(rng, sc.R, sc.t, 0.01, 0.02)
        p = api.pose12(R, t)
        kind = int(rng.integers(0, 3))
        flags = 0
        if rng.random() < 0.5 and n >= 3:
            ctx.inlier_mask(L.VOTE_33_23, api.pose7_from_Rt(sc.R, sc.t, L.F64 if f64 else L.F32), 0.3, 0.999, 2.0)
            flags |= L.USE_MASK
        rec = ctx.normal_eq(kind, p, flags)
        rec = rec[0] if isinstance(rec, tuple) else rec
        mom = ctx.p2p_moments(flags)
        q7 = np.array([api.pose7_from_Rt(*util.perturbed_pose(rng, sc.R, sc.t, 0.003 * h, 0.01 * h), L.F64 if f64 else L.F32) for h in range(int(rng.integers(1, 40)))])
        votes = ctx.score(L.VOTE_33_23, q7, 0.2, 0.9999, 2.0)
        ref = None
        if n >= 6:
            try:
                ref = ctx.gn_refine([kind], p, None, flags, 12, 0.0)[0]
            except L.RpeError as e:
                ref = "ERR " + str(e)[:60]
        dev = None   # the device loop: one launch (new) against one launch per iteration (old, RPE_DEVICE_LOOP_RESIDENT=0)
        if n >= 6:
            try:
                dev = ctx.gn_refine_device([(kind, 1.0)], p, flags, 12, 0.0)[0]
            except L.RpeError as e:
                dev = "ERR " + str(e)[:60]
        jnt = None   # the joint refinement: resident (new) against one launch per iteration
        if n >= 6:
            try:
                jnt = ctx.gn_refine_joint([(L.RES_P2P if kind != 1 else L.RES_P2PLANE, 1.0, 0, 1.0), (L.RES_BEARING, 2.0, 0, 1.0)], p, flags, 10, 0.0)[0]
            except L.RpeError as e:
                jnt = "ERR " + str(e)[:60]
        jdev = None   # and the joint objective in the device loop: one launch (new) against one launch per iteration
        if n >= 6:
            try:
                jdev = ctx.gn_refine_device([(L.RES_P2P if kind != 1 else L.RES_P2PLANE, 1.0), (L.RES_BEARING, 2.0)], p, flags, 10, 0.0)[0]
            except L.RpeError as e:
                jdev = "ERR " + str(e)[:60]
        out.append(dict(n=n, f64=f64, kind=kind, flags=flags, jdev=jdev if isinstance(jdev, str) or jdev is None else np.asarray(jdev).tolist(), jnt=jnt if isinstance(jnt, str) or jnt is None else np.asarray(jnt).tolist(), rec=np.asarray(rec).tolist(), mom=np.asarray(mom).tolist(), votes=votes.tolist(),
                        ref=ref if isinstance(ref, str) or ref is None else np.asarray(ref).tolist(),
                        dev=dev if isinstance(dev, str) or dev is None else np.asarray(dev).tolist()))
        ctx.close()
    print("RESULT " + json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(int(sys.argv[1]), int(sys.argv[2]))
    else:
        bad = 0
        total = 0
        for seed in range(8):
            res = {}
            for tag, env in (("new", {}), ("old", {"RPE_COLLECT": "0", "RPE_RESIDENT": "0", "RPE_DEVICE_LOOP_RESIDENT": "0"})):
                r = subprocess.run([sys.executable, __file__, str(seed), "25"], env=dict(os.environ, RPE_QUIET="1", **env), capture_output=True, text=True, timeout=900)
                if r.returncode != 0:
                    print("worker failed", tag, seed, r.stderr[-1500:]); bad += 1; continue
                res[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
            if len(res) < 2:
                continue
            for a, b in zip(res["new"], res["old"]):
                total += 1
                ra, rb = np.array(a["rec"]), np.array(b["rec"])
                tol = 1e-9 if a["f64"] else 1e-6
                ok = np.allclose(ra, rb, rtol=tol, atol=tol * (1 + np.abs(rb).max()))
                ok &= np.allclose(np.array(a["mom"]), np.array(b["mom"]), rtol=1e-9, atol=1e-9 * (1 + np.nanmax(np.abs(np.array(b["mom"])))), equal_nan=True)   # unmasked sums over NaN columns are NaN on both sides
                ok &= a["votes"] == b["votes"]
                if isinstance(a["ref"], list) and isinstance(b["ref"], list):
                    ok &= np.allclose(np.array(a["ref"]), np.array(b["ref"]), rtol=0, atol=1e-7)
                else:
                    ok &= (type(a["ref"]) == type(b["ref"]))
                if isinstance(a["dev"], list) and isinstance(b["dev"], list):
                    ok &= np.allclose(np.array(a["dev"]), np.array(b["dev"]), rtol=0, atol=1e-7)
                    if isinstance(a["ref"], list):
                        ok &= np.allclose(np.array(a["dev"]), np.array(a["ref"]), rtol=0, atol=1e-7)   # and against the host-driven loop of the same run
                else:
                    ok &= (type(a["dev"]) == type(b["dev"]))
                if isinstance(a["jnt"], list) and isinstance(b["jnt"], list):
                    ok &= np.allclose(np.array(a["jnt"]), np.array(b["jnt"]), rtol=0, atol=1e-7)
                else:
                    ok &= (type(a["jnt"]) == type(b["jnt"]))
                if isinstance(a["jdev"], list) and isinstance(b["jdev"], list):
                    ok &= np.allclose(np.array(a["jdev"]), np.array(b["jdev"]), rtol=0, atol=1e-7)
                    if isinstance(a["jnt"], list):
                        ok &= np.allclose(np.array(a["jdev"]), np.array(a["jnt"]), rtol=0, atol=1e-7)   # and against the host-driven joint loop
                else:
                    ok &= (type(a["jdev"]) == type(b["jdev"]))
                if not ok:
                    bad += 1
                    drec = float(np.max(np.abs(ra - rb) / (1e-300 + np.abs(rb).max())))
                    dmom = float(np.max(np.abs(np.array(a["mom"]) - np.array(b["mom"])) / (1e-300 + np.abs(np.array(b["mom"])).max())))
                    dref = float(np.max(np.abs(np.array(a["ref"]) - np.array(b["ref"])))) if isinstance(a["ref"], list) and isinstance(b["ref"], list) else str(a["ref"])[:40] + "|" + str(b["ref"])[:40]
                    print("MISMATCH", {k: a[k] for k in ("n", "f64", "kind", "flags")}, "rec", drec, "mom", dmom, "votes_eq", a["votes"] == b["votes"], "ref", dref)
        print("cases", total, "bad", bad)
