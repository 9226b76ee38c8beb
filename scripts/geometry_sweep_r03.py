"""Launch geometry of the one-launch normal-equation kernels at the sizes between a frame and a stream (development aid): HIP-event
time per launch for (workgroup size, workgroup cap) = RPE_BLOCK x RPE_MAX_BLOCKS, per residual kind.  usage: geometry_sweep_r03.py [n ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

sizes = [int(x) for x in sys.argv[1:]] or [307200, 1000000, 2500000]
for n in sizes:
    R, t, Q, P, Nn = bench.cheap_scene(n, seed=9)
    U = Q @ R.T.astype(np.float32) + t.astype(np.float32)
    U = (U / np.linalg.norm(U, axis=1, keepdims=True)).astype(np.float32)
    pose = api.pose12(R, t)
    for blk, cap in ((0, 256), (256, 256), (256, 512), (256, 1024), (512, 256), (512, 512)):
        os.environ["RPE_MAX_BLOCKS"] = str(cap)
        if blk:
            os.environ["RPE_BLOCK"] = str(blk)
        else:
            os.environ.pop("RPE_BLOCK", None)
        ctx = api.Context(0).load(L.F32, xw=Q, xc=P, bv=U, nc=Nn)
        row = {"n": n, "block": blk or "default", "max_blocks": cap}
        for name, kind in (("p2p", L.RES_P2P), ("p2plane", L.RES_P2PLANE), ("bearing", L.RES_BEARING)):
            for _ in range(5):
                ctx.normal_eq(kind, pose)
            ctx.timing_enable(40, 1)
            for _ in range(40):
                ctx.normal_eq(kind, pose)
            cnt, tot, mn = ctx.timing_collect()
            ctx.timing_enable(0, 1)
            row[name + "_us"] = round(tot / cnt * 1e3, 3)
        print(json.dumps(row), flush=True)
        ctx.close()
