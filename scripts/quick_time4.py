import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
def timeit(f, K=1000, W=100):
    for _ in range(W): f()
    t0 = time.perf_counter()
    for _ in range(K): f()
    return (time.perf_counter() - t0) / K * 1e6
sc = S.dense_depth_scene(1, 307200)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
p = api.pose12(sc.R, sc.t)
pp = p.copy()
print(mode, "before torch: gn_step %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp)))
import torch
if mode == "set_device": torch.cuda.set_device(0)
if mode == "sync": torch.cuda.synchronize()
if mode == "tensor": torch.zeros(1, device="cuda")
if mode == "stream": s = torch.cuda.Stream()
print(mode, "after torch : gn_step %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp)))
import ctypes as C
buf = np.zeros(32)
def launch_only():
    ctx.normal_eq_device(L.RES_P2P, p, dptr)
if mode in ("tensor",):
    d = torch.zeros(32, dtype=torch.float64, device="cuda"); dptr = d.data_ptr()
    print(mode, "launch-only (async) %.1f us" % timeit(launch_only)); ctx.synchronize()
print("threads:", len(os.listdir("/proc/self/task")))
ctx2 = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
print(mode, "new ctx after torch: gn_step %.1f us" % timeit(lambda: ctx2.gn_step(L.RES_P2P, pp)))
