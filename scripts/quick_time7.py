import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
L.lib(); L.device_count()
import torch
torch.cuda.set_device(0); torch.cuda.synchronize()
mode = sys.argv[1]
sc = S.dense_depth_scene(1, 307200)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
if mode == "alloc": rec = torch.zeros(32, dtype=torch.float64, device="cuda")
p0 = api.pose12(sc.R, sc.t)
pp = p0.copy()
def series(tag, n=12):
    out = []
    for chunk in range(n):
        t0 = time.perf_counter()
        for _ in range(500): ctx.gn_step(L.RES_P2P, pp)
        out.append((time.perf_counter() - t0) / 500 * 1e6)
    print(mode, tag, " ".join("%.1f" % x for x in out), flush=True)
series("a")
torch.cuda.synchronize()
series("after sync")
if mode == "alloc":
    del rec; torch.cuda.empty_cache(); series("after free")
