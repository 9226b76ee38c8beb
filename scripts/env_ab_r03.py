"""A/B of one environment switch of the library on the resident Gauss-Newton loop (development aid): alternate the values on ONE box,
one process each, and print the slope (2000 vs 4000 steps) of rpe_gn_refine at 307 200 masked point-to-point correspondences with the
host thread pinned.   python3 scripts/env_ab_r03.py RPE_POLL_DEPTH 1 4 [alternations]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(var):
    import bench
    from rgbd_pose_estimation_amd import _lib as L, api
    cpus = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, {cpus[min(16, len(cpus) - 1)]})
    sc = bench.make_shard(0, 307200)
    ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
    R0, t0 = bench.initial_pose(sc)
    ctx.inlier_mask(L.VOTE_33, api.pose7_from_Rt(R0, t0, L.F32), thre_3d=0.2)
    p = api.pose12(R0, t0)

    def f(k):
        ctx.gn_refine([L.RES_P2P], p, None, L.USE_MASK, k, 0.0)
    f(2000)
    best = []
    for _ in range(9):
        a = time.perf_counter(); f(2000); b = time.perf_counter(); f(4000); c = time.perf_counter()
        best.append(((c - b) - (b - a)) / 2000 * 1e6)
    best.sort()
    print(json.dumps({"switch": var, "value": os.environ.get(var), "us_per_step_median": round(best[len(best) // 2], 3), "us_per_step_min": round(best[0], 3)}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2])
    else:
        var, v0, v1 = sys.argv[1:4]
        for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 4):
            for v in (v0, v1):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", var], env=dict(os.environ, **{var: v}), check=False)
