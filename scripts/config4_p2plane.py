"""BASELINE.json configs[3]: NormalAOPoseAdapter-style point-to-plane, 1 M points with normals, one MI355X -- the HBM-bound roofline
run.  Steady state (the 36 MB working set stays in the 256 MiB Infinity Cache between Gauss-Newton iterations) and COLD (a
480 MB stream through another context evicts it before every launch, so the launch reads from HBM).  HIP events around the kernel;
run under rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE for the dispatch-timestamp and traffic views."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
kind = {"p2plane": L.RES_P2PLANE, "p2p": L.RES_P2P}[sys.argv[2] if len(sys.argv) > 2 else "p2plane"]
bpc = 36 if kind == L.RES_P2PLANE else 24
rng = np.random.default_rng(4)
R, t = S.random_pose(rng)
sc = S.simulate_2d_3d_nl_correspondences(rng, R, t, n, 1.0, 0.0, 0.02, 0.0, np.radians(2.0), 0.0).astype(np.float32)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, nc=sc.N)
big_n = 20000000
evict = api.Context(0).load(L.F32, xw=np.zeros((big_n, 3), np.float32), xc=np.zeros((big_n, 3), np.float32))
p = api.pose12(R, t)
for _ in range(20): ctx.normal_eq(kind, p)
K = 100
ctx.timing_enable(K, 1)
for _ in range(K): ctx.normal_eq(kind, p)
cnt, tot, mn = ctx.timing_collect()
steady = tot / cnt * 1e-3
ctx.timing_enable(K, 1)
for _ in range(K):
    evict.p2p_moments()          # 480 MB stream: the 36 MB set is gone from every cache level
    ctx.normal_eq(kind, p)
cnt, tot, mn2 = ctx.timing_collect()
cold = tot / cnt * 1e-3
ev_avg, _ = ctx.timing_calibrate(100)
print(json.dumps(dict(config="configs[3]" if kind == L.RES_P2PLANE else "configs[4] shard", n=n, kind="p2plane" if kind == L.RES_P2PLANE else "p2p",
                      bytes_per_corr=bpc, working_set_MB=bpc * n / 1e6,
                      steady_us=steady * 1e6, steady_GBs=bpc * n / steady / 1e9, steady_frac_of_8TBs=bpc * n / steady / 8e12,
                      cold_us=cold * 1e6, cold_min_us=mn2 * 1e3, cold_GBs=bpc * n / cold / 1e9, cold_frac_of_8TBs=bpc * n / cold / 8e12,
                      empty_event_pair_us=ev_avg * 1e3)), flush=True)
