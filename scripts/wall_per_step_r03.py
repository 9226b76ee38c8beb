import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
import bench
for n in (307200, 1000000):
    R, t, Q, P, Nn = bench.cheap_scene(n, seed=9)
    U = Q @ R.T.astype(np.float32) + t.astype(np.float32); U = (U / np.linalg.norm(U, axis=1, keepdims=True)).astype(np.float32)
    pose = api.pose12(R, t)
    for rep in range(2):
        for blk in (256, 512):
            os.environ["RPE_BLOCK"] = str(blk)
            ctx = api.Context(0).load(L.F32, xw=Q, xc=P, bv=U, nc=Nn)
            row = {"n": n, "block": blk}
            for name, kind in (("p2p", L.RES_P2P), ("p2plane", L.RES_P2PLANE), ("bearing", L.RES_BEARING)):
                q = pose.copy()
                for _ in range(300): ctx.gn_step(kind, q)
                t0 = time.perf_counter()
                for _ in range(3000): ctx.gn_step(kind, q)
                row[name + "_wall_us_per_step"] = round((time.perf_counter() - t0) / 3000 * 1e6, 2)
            print(json.dumps(row), flush=True)
            ctx.close()
