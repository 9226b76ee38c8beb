"""Round-4 experiment (review item 9): the autonomous resident loop (rpe_gn_refine_device, one launch) with its two hops
(granule -> collecting workgroup -> run record -> every workgroup) against ONE hop (RPE_AUTO_FLAT=1: every workgroup reads every
workgroup's granules itself), and the host-driven resident loop beside them.  us per iteration = slope between 1000 and 3000
iterations (tol = 0), alternating processes.  Stop rule of the review: keep the one-hop form if <= 5.3 us at 307 200."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker():
    import numpy as np
    import util
    from rgbd_pose_estimation_amd import _lib as L, api
    for n in (1000, 307200, 1000000):
        sc = util.scene33(5, n, np.float32, outliers=0.0)
        ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
        p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(1), sc.R, sc.t, 0.01, 0.03))
        ctx.gn_refine([L.RES_P2P], p0, max_iter=50, tol=0.0)            # verifies the arrays (CLEAN flavour for the device loop too)
        res = {}
        for name, f in (("device_loop", lambda k: ctx.gn_refine_device([(L.RES_P2P, 1.0)], p0, 0, k, 0.0)), ("host_driven", lambda k: ctx.gn_refine([L.RES_P2P], p0, max_iter=k, tol=0.0))):
            f(500)
            ts = {}
            for k in (1000, 3000):
                best = 1e9
                for _ in range(5):
                    t0 = time.perf_counter(); out = f(k); best = min(best, time.perf_counter() - t0)
                ts[k] = best
            res[name] = dict(us_per_iteration=(ts[3000] - ts[1000]) / 2000 * 1e6, pose=out[0].tolist(), iters=out[1])
        print(json.dumps(dict(flat=os.environ.get("RPE_AUTO_FLAT", "0"), n=n, device_loop_us=round(res["device_loop"]["us_per_iteration"], 3),
                              host_driven_us=round(res["host_driven"]["us_per_iteration"], 3),
                              pose_diff=float(np.max(np.abs(np.array(res["device_loop"]["pose"]) - np.array(res["host_driven"]["pose"]))))), flush=True)
        ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
    else:
        for rep in range(3):
            for flat in ("0", "1"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=dict(os.environ, RPE_AUTO_FLAT=flat), check=False)
