"""Does the HIP runtime that is loaded first (torch's bundled ROCm 7.0 vs /opt/rocm 7.2) change launch latency?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "torch_first":
    import torch
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
import numpy as np
from rgbd_pose_estimation_amd import _lib as L
if mode == "system_rt":
    L._preload_hip_runtime = lambda: None
from rgbd_pose_estimation_amd import api, simulator as S
def timeit(f, K=1000, W=100):
    for _ in range(W): f()
    t0 = time.perf_counter()
    for _ in range(K): f()
    return (time.perf_counter() - t0) / K * 1e6
sc = S.dense_depth_scene(1, 307200)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
p = api.pose12(sc.R, sc.t)
ctx.inlier_mask(L.VOTE_33, api.pose7_from_Rt(sc.R, sc.t), thre_3d=0.2)
pp = p.copy()
print(mode, "gn_step (mask) %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp, L.USE_MASK)))
import ctypes
print("  loaded:", [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
