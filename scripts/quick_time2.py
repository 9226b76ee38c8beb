"""Where does a GN step's wall time go? (development probe)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S

def timeit(f, K=500, W=50):
    for _ in range(W): f()
    t0 = time.perf_counter()
    for _ in range(K): f()
    return (time.perf_counter() - t0) / K * 1e6

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
sc = S.dense_depth_scene(1, n)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
p = api.pose12(sc.R, sc.t)
print(f"n={n}")
print("normal_eq (no mask)        %.1f us" % timeit(lambda: ctx.normal_eq(L.RES_P2P, p)))
ctx.inlier_mask(L.VOTE_33, api.pose7_from_Rt(sc.R, sc.t), thre_3d=0.2)
print("normal_eq (mask)           %.1f us" % timeit(lambda: ctx.normal_eq(L.RES_P2P, p, L.USE_MASK)))
pp = p.copy()
print("gn_step (mask)             %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp, L.USE_MASK)))
ctx.timing_enable(1000, 1)
print("gn_step + events every     %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp, L.USE_MASK), 500, 0)); print("   events:", ctx.timing_collect())
ctx.timing_enable(1000, 8)
print("gn_step + events 1/8       %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp, L.USE_MASK), 500, 0)); print("   events:", ctx.timing_collect())
ctx.timing_enable(0, 1)
print("moments                    %.1f us" % timeit(lambda: ctx.p2p_moments()))
import torch
st = torch.cuda.Stream()
ctx2 = api.Context(0, st.cuda_stream).load(L.F32, xw=sc.Q, xc=sc.P)
print("normal_eq on torch stream  %.1f us" % timeit(lambda: ctx2.normal_eq(L.RES_P2P, p)))
rec = torch.zeros(32, dtype=torch.float64, device="cuda")
def dev_path():
    ctx2.normal_eq_device(L.RES_P2P, p, rec.data_ptr()); return rec.cpu()
with torch.cuda.stream(st):
    print("normal_eq_device + .cpu()  %.1f us" % timeit(dev_path))
