"""Quick wall-clock probe of the hot kernels (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S

def main():
    for n in (307200, 1000000, 10000000):
        sc = S.dense_depth_scene(1, n)
        ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
        p = api.pose12(sc.R, sc.t)
        for _ in range(5): ctx.normal_eq(L.RES_P2P, p)
        K = 200
        t0 = time.perf_counter()
        for _ in range(K): ctx.normal_eq(L.RES_P2P, p)
        dt = (time.perf_counter() - t0) / K
        print(f"n={n} normal_eq p2p: {dt*1e6:.1f} us/iter  {n/dt:.3e} corr-res/s  {24*n/dt/1e9:.1f} GB/s (wall)")
        t0 = time.perf_counter()
        for _ in range(50): ctx.p2p_moments()
        dt = (time.perf_counter() - t0) / 50
        print(f"n={n} moments: {dt*1e6:.1f} us  {24*n/dt/1e9:.1f} GB/s (wall)")
        rng = np.random.default_rng(0)
        H = 256
        import math
        q = np.tile(np.array([1,0,0,0,0,0,0.0]), (H,1)); q[:,4:] = rng.standard_normal((H,3))
        for mode in (L.SCORE_FAST, L.SCORE_EXACT):
            ctx.score(L.VOTE_33, q, 0.2, mode=mode)
            t0 = time.perf_counter()
            for _ in range(5): ctx.score(L.VOTE_33, q, 0.2, mode=mode)
            dt = (time.perf_counter() - t0) / 5
            print(f"n={n} score33 H={H} mode={mode}: {dt*1e6:.1f} us  {n*H/dt:.3e} corr-hyp/s")
        ctx.close()
main()
