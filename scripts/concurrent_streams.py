"""Aggregate throughput of several independent Gauss-Newton loops running concurrently on one GPU (one context + stream + host
thread each; ctypes releases the GIL inside the library): the single-loop figure of bench.py is latency-bound, the chip is not full."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S

n = 307200
sc = S.dense_depth_scene(1, n)
def run(streams, steps=2000, device_resident=False):
    ctxs = [api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P) for _ in range(streams)]
    poses = [api.pose12(sc.R, sc.t) for _ in range(streams)]
    for c, p in zip(ctxs, poses):
        for _ in range(50): c.gn_step(L.RES_P2P, p)
    go = threading.Barrier(streams + 1)
    done = threading.Barrier(streams + 1)
    def work(c, p):
        go.wait()
        if device_resident:
            c.gn_refine_device([(L.RES_P2P, 1.0)], p, 0, steps, 0.0)
        else:
            for _ in range(steps): c.gn_step(L.RES_P2P, p)
        done.wait()
    th = [threading.Thread(target=work, args=(c, p)) for c, p in zip(ctxs, poses)]
    for t in th: t.start()
    go.wait(); t0 = time.perf_counter(); done.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    for c in ctxs: c.close()
    return dict(streams=streams, mode="device-resident" if device_resident else "host update", us_per_step_per_stream=dt / steps * 1e6,
                aggregate_corr_res_per_s=n * steps * streams / dt)
for dr in (False, True):
    for s in (1, 2, 4, 8, 16):
        print(json.dumps(run(s, device_resident=dr)), flush=True)
