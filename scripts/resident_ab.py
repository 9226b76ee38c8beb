"""A/B of the host-driven Gauss-Newton loop (rpe_gn_refine): RESIDENT kernel (one launch, poses handed over through device memory) against
one launch per iteration (RPE_RESIDENT=0), and the resident kernel's run length (RPE_RESIDENT_ROWS: workgroups per collecting workgroup;
1 = every workgroup's record goes to the host).  Wall time per step of the library's loop, pose agreement.  Development aid."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(n, kind, steps):
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from tail_timeline import scene
    R, t, arrs = scene(n)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    ctx.gn_refine([kind], p, max_iter=300, tol=0.0)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        q, its, step, cost = ctx.gn_refine([kind], p, max_iter=steps, tol=0.0)
        best = min(best, (time.perf_counter() - t0) / steps)
    p0 = p.copy(); p0[9:] += 0.02
    qc, itc, stepc, _ = ctx.gn_refine([kind], p0, max_iter=30, tol=1e-9)
    print(json.dumps(dict(resident=os.environ.get("RPE_RESIDENT", "1"), rows=os.environ.get("RPE_RESIDENT_ROWS", "auto"), n=n, kind=kind, us_per_step=best * 1e6, iters=its, pose=list(q), conv_iters=itc, conv_pose=list(qc))), flush=True)
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        variants = [dict(RPE_RESIDENT="0")] + [dict(RPE_RESIDENT="1")] + [dict(RPE_RESIDENT="1", RPE_RESIDENT_ROWS=t) for t in (sys.argv[1:] or ["1", "8", "15"])]
        for n, kind in ((307200, 0), (1000000, 1), (1250000, 0), (10000, 0), (10000000, 0)):
            for v in variants:
                e = dict(os.environ, **v)
                subprocess.run(["timeout", "120", sys.executable, os.path.abspath(__file__), "--worker", str(n), str(kind), "2000"], env=e, check=False)
