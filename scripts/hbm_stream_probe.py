"""The bandwidth-bound launches of bench.py's roofline_hbm block by themselves (for rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE /
WRITE_SIZE passes): 40 launches of the point-to-point kernel over 20 M correspondences (480 MB: HBM streaming), 40 over 10 M, 60 of the
point-to-plane kernel over 1 M with normals (configs[3], steady).  Prints the event-timed averages as one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rgbd_pose_estimation_amd import _lib as L, api

R, t, Q, P, _ = bench.cheap_scene(20_000_000)
pose = api.pose12(R, t)
out = {}
for name, n in (("p2p_20M", 20_000_000), ("p2p_10M", 10_000_000)):
    ctx = api.Context(0).load(L.F32, xw=Q[:n], xc=P[:n])
    for _ in range(5):
        ctx.normal_eq(L.RES_P2P, pose)
    ctx.timing_enable(40, 1)
    for _ in range(40):
        ctx.normal_eq(L.RES_P2P, pose)
    cnt, tot, mn = ctx.timing_collect()
    out[name] = {"launches": cnt, "avg_launch_us": tot / cnt * 1e3, "algorithmic_bytes": 24 * n, "achieved_GBs": 24 * n / (tot / cnt * 1e-3) / 1e9}
    ctx.close()
del Q, P
_, _, Q1, P1, N1 = bench.cheap_scene(1_000_000)
ctx = api.Context(0).load(L.F32, xw=Q1, xc=P1, nc=N1)
for _ in range(5):
    ctx.normal_eq(L.RES_P2PLANE, pose)
ctx.timing_enable(60, 1)
for _ in range(60):
    ctx.normal_eq(L.RES_P2PLANE, pose)
cnt, tot, mn = ctx.timing_collect()
out["config3_p2plane_1M_steady"] = {"launches": cnt, "avg_launch_us": tot / cnt * 1e3, "algorithmic_bytes": 36_000_000, "achieved_GBs": 36e6 / (tot / cnt * 1e-3) / 1e9}
ctx.close()
print(json.dumps(out))
