"""A/B of the cross-workgroup stage of the ordinary (one launch per call) reduction kernels: collecting workgroups + host-side final sum
(default) against the arrival-counter tail (RPE_COLLECT=0).  Per variant: dispatch-timestamp kernel time of the normal-equation kernel,
wall time per call of rpe_normal_eq / rpe_p2p_moments / rpe_inlier_mask / rpe_nl_round, and the launch-per-step Gauss-Newton loop
(RPE_RESIDENT=0).  Development aid; every variant runs in its own process (the knobs are read once per process)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def wall(f, reps):
    for _ in range(50):
        f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


def worker(n, kind):
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
    from tail_timeline import scene
    R, t, arrs = scene(n)
    rng = np.random.default_rng(2)
    arrs["bv"] = (arrs["xc"] / np.linalg.norm(arrs["xc"], axis=1, keepdims=True)).astype(np.float32)
    arrs["nw"] = (arrs["nc"] @ R).astype(np.float32)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    q7 = np.concatenate([api.quat_from_R(R) if hasattr(api, "quat_from_R") else S.quaternion_from_matrix(R), t]) if False else None
    out = dict(collect=os.environ.get("RPE_COLLECT", "1"), n=n, kind=kind)
    for _ in range(200):
        ctx.normal_eq(kind, p)
    ctx.timing_enable(1000, 1)
    for _ in range(1000):
        ctx.normal_eq(kind, p)
    cnt, tot, mn = ctx.timing_collect()
    ctx.timing_enable(0, 1)
    out["normal_eq_kernel_avg_us"] = tot / cnt * 1e3
    out["normal_eq_kernel_min_us"] = mn * 1e3
    out["normal_eq_call_us"] = wall(lambda: ctx.normal_eq(kind, p), 1000)
    out["moments_call_us"] = wall(lambda: ctx.p2p_moments(), 1000)
    if n <= 2000000:
        out["nl_round_call_us"] = wall(lambda: ctx.nl_round(np.zeros(3), np.zeros(3), np.zeros(3), R), 500)
    p0 = np.array(p)
    t0 = time.perf_counter()
    ctx.gn_refine([kind], p0, max_iter=2000, tol=0.0)
    out["gn_loop_us_per_step"] = (time.perf_counter() - t0) / 2000 * 1e6
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]))
    else:
        for n, kind in ((307200, 0), (1000000, 1), (1250000, 0), (10000, 0), (10000000, 0)):
            for v in ("0", "1"):
                subprocess.run(["timeout", "200", sys.executable, os.path.abspath(__file__), "--worker", str(n), str(kind)],
                               env=dict(os.environ, RPE_COLLECT=v, RPE_RESIDENT="0"), check=False)
