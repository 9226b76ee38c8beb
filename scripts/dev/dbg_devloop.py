"""Do the autonomous loop's two kernels (workers + solving workgroup) always meet?  Many refinements per size; prints the slow calls."""
import sys, os, time, hashlib
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
from tail_timeline import scene
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for n, kind in [tuple(int(v) for v in c.split(':')) for c in os.environ.get('CASES', '1000000:1,2000000:1,4000000:1').split(',')]:
    R, t, arrs = scene(n)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    ctx.gn_refine([kind], p, max_iter=50, tol=0.0)
    hs, ts = set(), []
    for i in range(calls):
        t0 = time.perf_counter()
        q, it = ctx.gn_refine_device([(kind, 1.0)], p, 0, 200, 0.0)[:2]
        ts.append((time.perf_counter() - t0) / 200 * 1e6)
        if ts[-1] > 500: print("   call", i, L.lib().rpe_last_error().decode()[:200], flush=True)
        hs.add(hashlib.md5(np.asarray(q).tobytes()).hexdigest()[:8])
    st = ctx.resident_state()
    print(n, kind, "poses", sorted(hs), "median us/iter", round(sorted(ts)[len(ts) // 2], 2), "slow calls (>3x median)", [(i, round(x, 1)) for i, x in enumerate(ts) if x > 3 * sorted(ts)[len(ts) // 2]],
          "solver", st.get("solver"), "lost", st["lost"], flush=True)
    ctx.close()
