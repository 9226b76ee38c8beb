"""Worker of tests/test_gpu_hostex.py: one rank of a sharded run whose per-iteration all-reduce is the HOST-side exchange
(rpe_hostex_init).  All ranks share cuda:0 here (one-GPU box): kernels, run records, shared-memory exchange and rank-ordered sums are
the real ones; with RPE_HOSTEX_ALLOW_SHARED=1 (and a problem small enough for every rank's grid to be resident at once) the resident
kernel runs on every rank as it would with one GPU per rank."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rgbd_pose_estimation_amd import _lib as L, api  # noqa: E402
from rgbd_pose_estimation_amd.distributed import init_host_exchange, shard_range  # noqa: E402
from util import scene33, perturbed_pose  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scene33(11, n, np.float32, noise=0.02, outliers=0.0)
    lo, hi = shard_range(n, rank, world)
    ctx = api.Context(0)
    ctx.load(L.F32, xw=sc.Q[lo:hi], xc=sc.P[lo:hi])
    pose = api.pose12(np.eye(3), np.zeros(3))
    for _ in range(20):   # a process's first launches can take seconds on a cold box
        ctx.normal_eq(L.RES_P2P, pose)
    dist.barrier()
    out = {"rank": rank, "hostex": bool(init_host_exchange(ctx))}
    poses7 = None
    if mode == "score":
        import oracle_lib as O
        rng = np.random.default_rng(5)
        poses7 = np.array([O.pose7_from_Rt(*perturbed_pose(rng, sc.R, sc.t, ang=0.003 * (h % 7), dt=0.01 * (h % 5)), False) for h in range(steps)])
    try:
        if out["hostex"]:
            if mode == "steps":
                ctx.gn_steps_dist(L.RES_P2P, pose, steps)
            elif mode == "refine":
                pose, its, step, cost = ctx.gn_refine([L.RES_P2P], pose, max_iter=steps, tol=1e-10)
                out["iters"] = its
            elif mode == "score":
                out["votes"] = ctx.score(L.VOTE_33, poses7, 0.1, mode=L.SCORE_EXACT).tolist()            # the table kernel (steps > 32) ...
                out["votes_short"] = ctx.score(L.VOTE_33, poses7[:12], 0.1, mode=L.SCORE_EXACT).tolist()   # ... and the single-launch form
            out["pose"] = np.asarray(pose).tolist()
    except L.RpeError as e:
        out["error"] = str(e)
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        full = api.Context(0)
        full.load(L.F32, xw=sc.Q, xc=sc.P)
        ref = api.pose12(np.eye(3), np.zeros(3))
        if mode == "steps":
            for _ in range(steps):
                full.gn_step(L.RES_P2P, ref)
        elif mode == "refine":
            ref, its, _, _ = full.gn_refine([L.RES_P2P], ref, max_iter=steps, tol=1e-10)
            gathered[0]["ref_iters"] = its
        else:
            ref = full.score(L.VOTE_33, poses7, 0.1, mode=L.SCORE_EXACT)
        full.close()
        print("RESULT " + json.dumps({"ranks": gathered, "reference": np.asarray(ref).tolist()}), flush=True)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
