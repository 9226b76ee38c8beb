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
