"""Multi-GPU driver of the hot path: one process per GPU, correspondences sharded by contiguous index ranges
(SURVEY.md section 8e), pose replicated, ONE all-reduce(sum) of the 32-double normal-equation record per
Gauss-Newton iteration (216 B of payload: 21 H + 6 g, plus cost / weight), and one all-reduce of the H int32
vote counters per scoring batch.  torch.distributed is only plumbing here: backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests.  The 6x6 solve and the SE(3) exp-map update run redundantly and identically on
every rank (host code of librgbdpose_hip.so), so no broadcast is needed.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L
from . import api


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous index range [lo, hi) of rank `rank`: [r*N/P, (r+1)*N/P)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class ShardedGaussNewton:
    """Gauss-Newton over a correspondence set sharded across ranks.

    local_normal_eq(pose12: np.ndarray) -> torch.Tensor[32] float64 holding THIS rank's record (on the GPU for the
    HIP path; the all-reduce then runs over RCCL).  In production it is HipShard.normal_eq; the CPU tests inject an
    oracle-backed callable to exercise the sharding + collective + update logic without a GPU.
    """

    def __init__(self, local_normal_eq: Callable[[np.ndarray], torch.Tensor], group=None, exchange: "Optional[api.HostExchange]" = None):
        self.local_normal_eq = local_normal_eq
        self.group = group
        self.exchange = exchange   # host-side exchange between the node's rank processes instead of a torch.distributed collective

    def reduce_record(self, pose12: np.ndarray) -> np.ndarray:
        rec = self.local_normal_eq(pose12)
        if self.exchange is not None:
            return self.exchange.allreduce_f64(rec.detach().to("cpu").numpy())
        if dist.is_available() and dist.is_initialized():   # also with one rank: same code path at every N
            dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=self.group)
        return rec.detach().to("cpu").numpy()

    def step(self, pose12: np.ndarray):
        """One iteration.  Returns (new pose12, |delta|, global record)."""
        ne = self.reduce_record(pose12)
        delta = api.gn_solve(ne)
        return api.gn_apply(delta, pose12), float(np.linalg.norm(delta)), ne

    def refine(self, pose12: np.ndarray, max_iter: int = 20, tol: float = 1e-9):
        p = np.array(pose12, np.float64).reshape(12)
        it, step = 0, float("inf")
        for it in range(1, max_iter + 1):
            p, step, _ = self.step(p)
            if step < tol:
                break
        return p, it, step


class ShardedScorer:
    """Batched RANSAC hypothesis scoring over sharded correspondences: local int32 votes, all-reduce(sum)."""

    def __init__(self, local_votes: Callable[[np.ndarray], torch.Tensor], group=None, exchange: "Optional[api.HostExchange]" = None):
        self.local_votes = local_votes
        self.group = group
        self.exchange = exchange

    def score(self, poses7: np.ndarray) -> np.ndarray:
        v = self.local_votes(poses7)
        if self.exchange is not None:
            return self.exchange.allreduce_i32(v.detach().to("cpu").numpy())
        if _world(self.group) > 1:
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        return v.detach().to("cpu").numpy()


def init_native_comm(ctx: "api.Context", group=None) -> bool:
    """Give `ctx` its own RCCL communicator (one per rank, same device as the context): rank 0 draws the 128-byte id,
    torch.distributed only ferries it to the other ranks.  After this, Context.gn_step_dist() runs the whole sharded step
    -- kernel, in-place all-reduce of the launch's run records (8 x 32 doubles) on the context's stream, publish, host solve --
    inside librgbdpose_hip.so, without a Python-side collective; Context.gn_steps_dist(kind, pose, k) runs k of them, and
    Context.gn_steps_dist_device(kind, pose, k) the same k with the solve and the exp-map in the kernels (the host enqueues
    everything and waits once).  Returns False (and leaves ctx untouched) if RCCL cannot be set up."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ok = 1
    try:
        box = [api.comm_unique_id() if rank == 0 else None]
    except Exception as e:  # noqa: BLE001
        box, ok = [None], 0
        print(f"[rgbd_pose_estimation_amd] rank {rank}: no RCCL unique id ({e})", flush=True)
    dist.broadcast_object_list(box, src=0, group=group)
    if box[0] is None:
        return False          # rank 0 could not draw an id: every rank sees None and takes the torch.distributed path
    try:
        if ok:
            ctx.comm_init(world, rank, box[0])
    except Exception as e:  # noqa: BLE001 -- the torch.distributed path stays available
        ok = 0
        print(f"[rgbd_pose_estimation_amd] rank {rank}: native RCCL communicator unavailable ({e})", flush=True)
    # all ranks must take the same path: one rank on the library's communicator and another on torch's would dead-lock
    flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        if ok:
            ctx.comm_destroy()
        return False
    return True


def init_p2p(ctx: "api.Context", group=None) -> bool:
    """Peer-to-peer all-reduce over xGMI for ONE node (<= 8 ranks): every rank exports the HIP IPC handle of its mailbox,
    torch.distributed gathers the handles, every rank maps its peers' mailboxes.  After this Context.gn_step_dist() is ONE
    kernel launch per step (the kernel's last workgroup exchanges and sums the records).  The wait inside the kernel is bounded
    (10 s), so ranks should enter their first exchange together: run a launch locally and barrier first (a process's first
    launches on a cold box can take seconds).  All ranks return the same answer;
    False leaves the context untouched (use init_native_comm / the torch.distributed path instead)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world > 8:
        return False
    ok, handle = 1, None
    try:
        handle = ctx.p2p_export()
    except Exception as e:  # noqa: BLE001
        ok = 0
        print(f"[rgbd_pose_estimation_amd] rank {rank}: no IPC handle for the peer-to-peer mailbox ({e})", flush=True)
    handles = [None] * world
    dist.all_gather_object(handles, handle, group=group)
    if any(h is None for h in handles):
        ok = 0
    if ok:
        try:
            ctx.p2p_init(world, rank, b"".join(handles))
            ctx.p2p_handles = b"".join(handles)   # a later re-initialisation of the same mailboxes (rpe_p2p_init clears the own one)
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"[rgbd_pose_estimation_amd] rank {rank}: peers' mailboxes cannot be mapped ({e})", flush=True)
    flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        ctx.p2p_destroy()
        return False
    return True


def exchange_name(group=None) -> str:
    """A segment name every rank of the group agrees on (rank 0 draws it, torch.distributed ferries it)."""
    import os
    import secrets
    rank = dist.get_rank(group)
    box = [f"/rpe_hx_{os.getpid()}_{secrets.token_hex(4)}" if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


def open_host_exchange(group=None, timeout_s: float = 10.0) -> "api.HostExchange":
    """The exchange by itself (no GPU context): rank 0 creates the segment, everybody opens it, one exchange as the rendezvous, rank 0
    drops the name.  For ShardedGaussNewton / ShardedScorer (`exchange=`)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    name = exchange_name(group)
    hx = api.HostExchange(name, world, rank, create=(rank == 0), timeout_s=timeout_s)
    assert hx.allreduce_f64([1.0])[0] == world
    if rank == 0:
        hx.unlink()
    return hx


def init_host_exchange(ctx: "api.Context", group=None) -> bool:
    """Host-side all-reduce for ONE node: the rank processes exchange their 32-double records (and vote counters) through a POSIX
    shared-memory segment and add them in rank order, so the sharded step needs no collective kernel at all and rpe_gn_refine keeps
    its resident kernel on every rank.  Rank 0 creates the segment, a barrier, the others open it; rpe_hostex_init's first exchange
    is the rendezvous.  All ranks return the same answer; False leaves the context untouched."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world > 8:
        return False
    name = exchange_name(group)
    ok = 1
    # exactly one creator, and it must come first: rank 0 enters rpe_hostex_init (create) while the others wait for the segment
    # to appear inside their own call (bounded); the call's first exchange then holds everyone until all ranks have it mapped
    try:
        ctx.hostex_init(world, rank, name, rank == 0)
    except Exception as e:  # noqa: BLE001
        ok = 0
        print(f"[rgbd_pose_estimation_amd] rank {rank}: host exchange unavailable ({e})", flush=True)
    flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        ctx.hostex_destroy()
        return False
    return True


class HipShard:
    """This rank's shard resident in HBM + the device-side record buffer the collective reduces in place."""

    def __init__(self, device: int, stream: Optional[torch.cuda.Stream] = None):
        self.device = torch.device("cuda", device)
        self.stream = stream
        self.ctx = api.Context(device, stream.cuda_stream if stream is not None else None)
        self.rec = torch.zeros(32, dtype=torch.float64, device=self.device)
        self.kind, self.flags = L.RES_P2P, 0

    def normal_eq(self, pose12: np.ndarray) -> torch.Tensor:
        self.ctx.normal_eq_device(self.kind, pose12, self.rec.data_ptr(), self.flags)
        return self.rec

    def votes(self, kind: int, thre_3d=0.0, cos_thr=2.0, cos_nl=2.0, mode=L.SCORE_FAST):
        def f(poses7):
            v = self.ctx.score(kind, poses7, thre_3d, cos_thr, cos_nl, mode)
            return torch.from_numpy(v).to(self.device)
        return f
