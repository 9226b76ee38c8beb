"""rpe_gn_refine_device (solve + exp-map on the GPU): the autonomous resident launch against one launch per iteration
(RPE_DEVICE_LOOP_RESIDENT=0), with the host-driven resident loop (rpe_gn_refine) beside them.  Per-iteration time = slope of the call's wall
time over the iteration count; poses compared at a fixed count.  One subprocess per mode (the switch is read once).  Development aid."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    from tail_timeline import scene
    n, kind, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    R, t, arrs = scene(n)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    if mode == "host":
        run = lambda K: ctx.gn_refine([kind], p, max_iter=K, tol=0.0)
    else:
        run = lambda K: ctx.gn_refine_device([(kind, 1.0)], p, 0, K, 0.0)
    run(300)
    rows = []
    for K in (5, 20, 100, 400):
        ts = []
        for _ in range(100):
            t0 = time.perf_counter()
            out = run(K)
            ts.append(time.perf_counter() - t0)
        rows.append((K, float(np.median(ts)) * 1e6))
    Ks = np.array([r[0] for r in rows], float); T = np.array([r[1] for r in rows])
    slope, icpt = np.polyfit(Ks, T, 1)
    pose = run(12)[0]
    print(json.dumps(dict(n=n, kind=kind, mode=mode, us_per_call={str(k): round(v, 2) for k, v in rows}, us_per_iteration=slope, intercept_us=icpt,
                          pose12=[float(x) for x in pose])))
    sys.exit(0)

sizes = [(307200, 0), (1000, 0), (1000000, 1), (10000000, 0)]
for n, kind in sizes:
    res = {}
    for mode, env in (("host", {}), ("device_resident", {"RPE_DEVICE_LOOP_RESIDENT": "1"}), ("device_per_launch", {"RPE_DEVICE_LOOP_RESIDENT": "0"})):
        r = subprocess.run([sys.executable, __file__, "child", str(n), str(kind), mode], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(json.dumps(dict(n=n, kind=kind, mode=mode, error=r.stderr[-400:])), flush=True)
            continue
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    if len(res) == 3:
        import numpy as np
        d = lambda a, b: float(np.max(np.abs(np.array(res[a]["pose12"]) - np.array(res[b]["pose12"]))))
        print(json.dumps(dict(n=n, kind=kind, us_per_iteration={m: round(res[m]["us_per_iteration"], 3) for m in res},
                              intercept_us={m: round(res[m]["intercept_us"], 2) for m in res},
                              max_pose_diff_resident_vs_per_launch=d("device_resident", "device_per_launch"),
                              max_pose_diff_resident_vs_host=d("device_resident", "host"))), flush=True)
