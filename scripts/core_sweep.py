"""Which host core polls fastest: the 20-step refinement + closing synchronize (the driver-sized bench region) and the steady step, with the
calling thread pinned to one core after another (every 8th core of both sockets and a few SMT siblings).  Development aid for bench.py's
candidate list."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
torch.zeros(1, device="cuda")
from rgbd_pose_estimation_amd import _lib as L, api
from tail_timeline import scene
R, t, arrs = scene(307200)
ctx = api.Context(0).load(L.F32, **arrs)
p = api.pose12(R, t)
ncpu = os.cpu_count()
allowed = sorted(os.sched_getaffinity(0))
cands = [c for c in list(range(1, ncpu // 2, 8)) + [ncpu // 2 + 1, ncpu // 2 + 65, 192] if c in allowed]
ctx.gn_refine([0], p, max_iter=3000, tol=0.0)
rows = []
for c in cands:
    os.sched_setaffinity(0, {c})
    ctx.gn_refine([0], p, max_iter=500, tol=0.0)
    reg = []
    for _ in range(60):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.gn_refine([0], p, max_iter=20, tol=0.0)
        torch.cuda.synchronize()
        reg.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    ctx.gn_refine([0], p, max_iter=2000, tol=0.0)
    steady = (time.perf_counter() - t0) / 2000
    rows.append(dict(cpu=c, region20_us_per_step=float(np.median(reg)) / 20 * 1e6, steady_us_per_step=steady * 1e6))
    print(json.dumps(rows[-1]), flush=True)
os.sched_setaffinity(0, set(allowed))
