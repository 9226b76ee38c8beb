"""Worker of tests/test_gpu_p2p.py: one rank of a sharded Gauss-Newton run whose per-step all-reduce is the peer-to-peer
exchange inside the kernel (rpe_p2p_*).  All ranks share cuda:0 here (one-GPU box): the IPC mapping, the flag-in-data mailbox
protocol, the rank-ordered sum and the time-out path are the real ones; only the wire (xGMI) is missing."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rgbd_pose_estimation_amd import _lib as L, api  # noqa: E402
from rgbd_pose_estimation_amd.distributed import init_p2p, shard_range  # noqa: E402
from util import scene33  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = sys.argv[1]
    n, steps = int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scene33(11, n, np.float32, noise=0.02, outliers=0.0)
    lo, hi = shard_range(n, rank, world)
    ctx = api.Context(0)
    ctx.load(L.F32, xw=sc.Q[lo:hi], xc=sc.P[lo:hi])
    out = {"rank": rank, "p2p": bool(init_p2p(ctx))}
    pose = api.pose12(np.eye(3), np.zeros(3))
    for _ in range(20):          # first launches of a process can take seconds on a cold box: get them out of the way locally,
        ctx.normal_eq(L.RES_P2P, pose)
    dist.barrier()               # ... then enter the first exchange together (the in-kernel wait is bounded)
    try:
        if out["p2p"]:
            if mode == "straggler" and rank == world - 1:
                import time
                ctx.gn_step_dist(L.RES_P2P, pose)      # step 0 together ...
                dist.barrier()
                time.sleep(14.0)                        # ... then this rank goes missing for longer than the time-out
                out["slept"] = True
            elif mode == "score":
                rng = np.random.default_rng(5)
                import oracle_lib as O
                from util import perturbed_pose
                poses = np.array([O.pose7_from_Rt(*perturbed_pose(rng, sc.R, sc.t, ang=0.003 * (h % 7), dt=0.01 * (h % 5)), False) for h in range(steps)])
                out["votes"] = ctx.score(L.VOTE_33, poses, 0.1, mode=L.SCORE_EXACT).tolist()
                out["votes2"] = ctx.score(L.VOTE_33, poses[::-1].copy(), 0.1, mode=L.SCORE_EXACT).tolist()   # a second exchange (other parity)
            elif mode == "soak":   # the exchanged record must be bitwise the same at every one of `steps` steps from a fixed pose
                import ctypes as C
                rec, first = np.zeros(32), None
                bad = 0
                for k in range(steps):
                    q = pose.copy()
                    L.check(L.lib().rpe_gn_step_dist(ctx._h, L.RES_P2P, 0, q.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p), None))
                    if first is None:
                        first = rec.copy()
                    bad += int(not np.array_equal(rec, first))
                out["bad"] = bad
                out["record"] = first.tolist()
            elif mode == "device":
                pose, its, step, cost = ctx.gn_refine_device([(L.RES_P2P, 1.0)], pose, 0, steps, 1e-10)
                out["iters"] = its
            else:
                for k in range(steps):
                    ctx.gn_step_dist(L.RES_P2P, pose)
                    if mode == "straggler" and k == 0:
                        dist.barrier()
                    if mode == "reinit" and k == 0:
                        # a second rpe_p2p_init after a session of ONE step (tag 1 sits in the parity-0 slots): the step counters restart
                        # at 0, so the own mailbox must come back empty or the new step 0 would accept the stale records
                        dist.barrier()
                        ctx.p2p_init(world, rank, ctx.p2p_handles)
                        dist.barrier()
            out["pose"] = np.asarray(pose).tolist()
    except L.RpeError as e:
        out["error"] = str(e)
        out["code"] = e.code
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        ref = None
        if mode == "score":
            rng = np.random.default_rng(5)
            import oracle_lib as O
            from util import perturbed_pose
            poses = np.array([O.pose7_from_Rt(*perturbed_pose(rng, sc.R, sc.t, ang=0.003 * (h % 7), dt=0.01 * (h % 5)), False) for h in range(steps)])
            full = api.Context(0)
            full.load(L.F32, xw=sc.Q, xc=sc.P)
            ref = full.score(L.VOTE_33, poses, 0.1, mode=L.SCORE_EXACT).tolist()
            full.close()
        if mode in ("steps", "device", "reinit"):
            full = api.Context(0)
            full.load(L.F32, xw=sc.Q, xc=sc.P)
            ref = api.pose12(np.eye(3), np.zeros(3))
            if mode in ("steps", "reinit"):
                for _ in range(steps):
                    full.gn_step(L.RES_P2P, ref)
            else:
                ref, its, _, _ = full.gn_refine_device([(L.RES_P2P, 1.0)], ref, 0, steps, 1e-10)
                gathered[0]["ref_iters"] = its
            ref = np.asarray(ref).tolist()
            full.close()
        print("RESULT " + json.dumps({"ranks": gathered, "reference": ref}), flush=True)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
