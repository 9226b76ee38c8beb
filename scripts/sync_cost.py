"""Where the driver-sized bench region (20 steps bracketed by torch.cuda.synchronize) spends its time beyond the steps: the refine call
against the closing synchronize, under the runtime's wait policies (default, ROC_ACTIVE_WAIT_TIMEOUT).
One subprocess per policy.  Development aid."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    mode = sys.argv[2]
    import numpy as np, torch
    torch.cuda.init(); torch.zeros(1, device="cuda")
    from rgbd_pose_estimation_amd import _lib as L, api
    from tail_timeline import scene
    R, t, arrs = scene(307200)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    ctx.gn_refine([0], p, max_iter=3000, tol=0.0)
    a, b, c = [], [], []
    for _ in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.gn_refine([0], p, max_iter=20, tol=0.0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        a.append(t1 - t0); b.append(t2 - t1); c.append(t3 - t2)
    med = lambda v: float(np.median(v)) * 1e6
    print(json.dumps(dict(mode=mode, refine20_us=med(a), closing_sync_us=med(b), idle_sync_us=med(c), region_us_per_step=(med(a) + med(b)) / 20)))
    sys.exit(0)

for mode, env in (("default", {}), ("active_wait_1ms", {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"})):
    r = subprocess.run([sys.executable, __file__, "child", mode], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else json.dumps(dict(mode=mode, error=r.stderr[-300:])), flush=True)
