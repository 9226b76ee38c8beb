"""Roofline probes for DESIGN.md / profiles (development aid): HIP-event time of the normal-equation kernels at the
BASELINE.json sizes, from the 307 200-point config (cache resident) to a 480 MB working set (past the 256 MiB
Infinity Cache, true HBM streaming)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S

def run(n, kind, name, bytes_per, steps=60):
    rng = np.random.default_rng(1)
    R, t = S.random_pose(rng)
    # cheap big scenes: tile a 1M-point scene (values do not matter for bandwidth)
    base = S.simulate_2d_3d_nl_correspondences(rng, R, t, min(n, 1_000_000), 1.0, 0.0, 0.02, 0.0, 0.03, 0.0).astype(np.float32)
    reps = (n + len(base.Q) - 1) // len(base.Q)
    tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    ctx = api.Context(0).load(L.F32, xw=tile(base.Q), xc=tile(base.P), bv=tile(base.U), nc=tile(base.N))
    p = api.pose12(R, t)
    for _ in range(5): ctx.normal_eq(kind, p)
    ctx.timing_enable(steps, 1)
    t0 = time.perf_counter()
    for _ in range(steps): ctx.normal_eq(kind, p)
    wall = (time.perf_counter() - t0) / steps
    cnt, tot, mn = ctx.timing_collect()
    avg = tot / cnt * 1e-3
    out = dict(name=name, n=n, kind=kind, bytes_per_corr=bytes_per, working_set_MB=bytes_per * n / 1e6, kernel_avg_us=avg * 1e6, kernel_min_us=mn * 1e3,
               achieved_GBs=bytes_per * n / avg / 1e9, frac_of_8TBs=bytes_per * n / avg / 8e12, wall_us_per_step=wall * 1e6,
               corr_res_per_s_wall=n / wall)
    print(json.dumps(out), flush=True)
    ctx.close()

sizes = [int(x) for x in sys.argv[1:]] or [307200, 1000000, 1250000, 10000000, 20000000]
for n in sizes:
    run(n, L.RES_P2P, "p2p", 24)
for n in sizes:
    if n <= 10000000:
        run(n, L.RES_P2PLANE, "p2plane", 36)
run(1000000, L.RES_BEARING, "bearing", 24)

# ---- the other streaming kernels at 1M / 10M correspondences (wall time per call incl. launch + publish; kernel-only for K1-K3 above)
def other(n):
    rng = np.random.default_rng(2)
    R, t = S.random_pose(rng)
    base = S.simulate_2d_3d_nl_correspondences(rng, R, t, min(n, 1_000_000), 1.0, 0.0, 0.02, 0.0, 0.03, 0.0).astype(np.float32)
    reps = (n + len(base.Q) - 1) // len(base.Q)
    tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    ctx = api.Context(0).load(L.F32, xw=tile(base.Q), xc=tile(base.P), bv=tile(base.U), nw=tile(base.M), nc=tile(base.N))
    q7 = api.pose7_from_Rt(R, t)
    def timeit(f, K=30):
        for _ in range(3): f()
        t0 = time.perf_counter()
        for _ in range(K): f()
        return (time.perf_counter() - t0) / K
    rows = [("moments K1'", 24, lambda: ctx.p2p_moments()),
            ("mask K4b 33", 24 + 2, lambda: ctx.inlier_mask(L.VOTE_33, q7, 0.2)),
            ("mask K4b nn_33_23", 60 + 6, lambda: ctx.inlier_mask(L.VOTE_NN_33_23, q7, 0.2, 0.9999, 0.995)),
            ("nl_round K5", 60 + 6, lambda: ctx.nl_round(np.zeros(3), np.zeros(3), np.zeros(3), np.eye(3)))]
    for name, bpc, f in rows:
        dt = timeit(f)
        print(json.dumps(dict(name=name, n=n, bytes_per_corr=bpc, wall_us=dt * 1e6, wall_GBs=bpc * n / dt / 1e9)), flush=True)
    H = 512
    poses = np.tile(q7, (H, 1)); poses[:, 4:] += 0.01 * rng.standard_normal((H, 3))
    for kind, nm in ((L.VOTE_33, "33"), (L.VOTE_33_23, "33_23"), (L.VOTE_NN_33_23, "nn_33_23")):
        for mode in (L.SCORE_FAST, L.SCORE_EXACT):
            dt = timeit(lambda: ctx.score(kind, poses, 0.2, 0.9999, 0.995, mode), 5)
            print(json.dumps(dict(name=f"score K4 {nm} {'exact' if mode else 'fast'} H={H}", n=n, wall_us=dt * 1e6, corr_hyp_per_s=n * H / dt)), flush=True)
    ctx.close()

for n in (307200, 1000000, 10000000):
    other(n)
