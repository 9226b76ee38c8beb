"""One rank of the real multi-GPU checks (tests/test_gpu_multi_gpu.py): ONE PROCESS PER GPU, rank r on device r, the library's own
RCCL communicator (rpe_comm_init) -- no torch, no launcher: the 128-byte unique id travels through a file written by rank 0.
usage: multigpu_worker.py <rank> <world> <dir> <n_total> <seed>
Writes <dir>/rank<r>.json: PCI bus id, ncclCommCount, this shard's local record, the all-reduced record of one sharded step, the pose
after K sharded steps, this shard's local votes and the all-reduced votes."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def scene(seed, n):
    import util
    return util.scene_full(seed, n, np.float32, n2d=2.0, n3d=0.03, outliers=0.0)


def main():
    rank, world, d, n, seed = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    from rgbd_pose_estimation_amd import _lib as L, api
    from rgbd_pose_estimation_amd.distributed import shard_range
    sc = scene(seed, n)                                   # every rank builds the same scene and takes its contiguous range
    lo, hi = shard_range(n, rank, world)
    idf = os.path.join(d, "nccl_id.bin")
    if rank == 0:
        uid = api.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 120:
                sys.exit("rank %d: no unique id from rank 0" % rank)
            time.sleep(0.01)
        uid = open(idf, "rb").read()
    ctx = api.Context(rank).load(L.F32, xw=sc.Q[lo:hi], xc=sc.P[lo:hi])
    local = api.Context(rank).load(L.F32, xw=sc.Q[lo:hi], xc=sc.P[lo:hi])   # the same shard without a communicator: local records / votes
    ctx.comm_init(world, rank, uid)
    out = {"rank": rank, "world": world, "range": [lo, hi], "bus_id": ctx.bus_id(), "comm_count": ctx.comm_count()}
    rng = np.random.default_rng(seed)
    import util
    p0 = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t, 0.01, 0.03))      # the same start pose on every rank (same generator state)
    out["local_record"] = local.normal_eq(L.RES_P2P, p0)[0].tolist()
    ne = np.zeros(32)
    p1 = p0.copy()
    step = C.c_double(0)
    L.check(L.lib().rpe_gn_step_dist(ctx._h, L.RES_P2P, 0, p1.ctypes.data_as(C.c_void_p), ne.ctypes.data_as(C.c_void_p), C.byref(step)))
    out["allreduced_record"] = ne.tolist()
    out["pose_after_one_step"] = p1.tolist()
    p = p0.copy()
    out["last_step"] = ctx.gn_steps_dist(L.RES_P2P, p, 8)
    out["pose_after_8_steps"] = p.tolist()
    poses = np.array([api.pose7_from_Rt(sc.R, sc.t, L.F32), api.pose7_from_Rt(*util.perturbed_pose(rng, sc.R, sc.t, 0.02, 0.05), L.F32)])
    out["local_votes"] = local.score(L.VOTE_33, poses, 0.1).tolist()
    out["allreduced_votes"] = ctx.score(L.VOTE_33, poses, 0.1).tolist()
    ctx.comm_destroy()
    ctx.close(); local.close()
    with open(os.path.join(d, "rank%d.json" % rank), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
