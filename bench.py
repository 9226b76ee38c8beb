#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

A "step" is ONE Gauss-Newton iteration of the point-to-point absolute-orientation refinement over one batch of
synthetic correspondences resident in HBM: stage-1 normal-equation kernel (K1) + stage-2 reduction, [all-reduce
of the 32-double record over RCCL when --gpus > 1], D2H of the record, 6x6 solve and SE(3) exp-map update on the
host.  Workload at N=1 = BASELINE.json configs[1]: 640x480 dense depth = 307 200 3D-3D correspondences, fp32.
Weak scaling: every rank holds its own 307 200-correspondence shard (global problem = N x 307 200), same pose.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Nothing here reads /root/reference.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 307200           # 640 x 480
BYTES_PER_CORR = 24 + 2      # Xw 12 + Xc 12 + short inlier mask 2 (p2p fp32 + mask)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
THRE_3D = 0.2                # Parameters.yml thre_3d


def make_shard(rank: int, n: int):
    """Same ground-truth pose on every rank, different points per rank (Simulator.hpp model, Parameters.yml values)."""
    from rgbd_pose_estimation_amd import simulator as S
    R, t = S.random_pose(np.random.default_rng(20260101))
    rng = np.random.default_rng(1000 + rank)
    return S.simulate_3d_3d_correspondences(rng, R, t, n, 0.05, 0.1).astype(np.float32)


def initial_pose(sc):
    """Start 0.02 rad / 5 cm away from the truth (what a RANSAC winner looks like)."""
    w = np.array([0.012, -0.010, 0.0125]); th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    dR = np.eye(3) + math.sin(th) / th * K + (1 - math.cos(th)) / th ** 2 * K @ K
    return dR @ sc.R, sc.t + np.array([0.03, -0.03, 0.03])


def rot_err(Ra, Rb):
    D = Ra @ Rb.T
    s = np.linalg.norm([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]]) / 2
    return math.atan2(s, (np.trace(D) - 1) / 2)


def cpu_baseline(sc, seconds: float):
    """Oracle (CPU restatement of the reference) timed on this host: shinji_ls2<float> through AOOnlyPoseAdapter's
    virtual getters, exactly what Library.cpp ao() runs; 1 thread because the reference is single-threaded."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import oracle_lib as O
        lib = O.lib()
        xw, xc = np.ascontiguousarray(sc.Q, np.float32), np.ascontiguousarray(sc.P, np.float32)
        R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
        n = len(xw)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        lib.orc_time_ao(p(xw), p(xc), n, 1, p(R), p(t))
        reps, spent = 0, 0.0
        while spent < seconds:
            spent += lib.orc_time_ao(p(xw), p(xc), n, 20, p(R), p(t)); reps += 20
        pose = np.concatenate([np.eye(3).reshape(9), np.zeros(3)])
        out = np.zeros(29)
        g = lib.orc_time_gn_p2p(p(xw), p(xc), C.c_long(n), 20, p(pose), p(out)) / 20
        # all-core variant of the same pass: the reference itself is single-threaded, so this is beside, not instead
        threads = max(1, min(64, (os.cpu_count() or 1)))
        lib.orc_time_gn_p2p_threads.restype = C.c_double
        ga = lib.orc_time_gn_p2p_threads(p(xw), p(xc), C.c_long(n), 200, p(pose), p(out), threads) / 200
        # the reference's own default build has no optimisation flag at all (CMakeLists.txt:13-15): the same port compiled that way
        o0 = None
        try:
            o0_path = os.path.join(ROOT, "oracle", "liboracle_O0.so")
            if os.path.exists(o0_path):
                lib0 = C.CDLL(o0_path)
                lib0.orc_time_ao.restype = C.c_double
                lib0.orc_time_ao(p(xw), p(xc), n, 1, p(R), p(t))
                r0, s0 = 0, 0.0
                while s0 < min(2.0, seconds):
                    s0 += lib0.orc_time_ao(p(xw), p(xc), n, 5, p(R), p(t)); r0 += 5
                o0 = n * r0 / s0
        except Exception:
            o0 = None
        return {"value": n * reps / spent, "unit": "correspondence-residuals/s", "cores": 1, "kind": "port", "no_O_flag_value": o0,
                "sample": f"{reps} calls of the oracle's shinji_ls2<float> (AOOnlyPoseAdapter virtual getters, gather + centroid + covariance "
                          f"passes + 3x3 SVD = Library.cpp ao()) on the same {n}-correspondence scene, g++ -O2, 1 thread, {spent:.1f} s",
                "gn_pass_fp64_value": n / g, "gn_pass_fp64_all_cores": {"value": n / ga, "threads": threads}, "host_cpus": os.cpu_count()}
    except Exception as e:  # the baseline is a reported number, never a dependency of the GPU path
        return {"value": None, "unit": "correspondence-residuals/s", "cores": 1, "kind": "port", "sample": f"unavailable: {e!r}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--n-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs reported beside the headline (profiling runs: only the headline's launches)")
    ap.add_argument("--debug-chunks", action="store_true")
    ap.add_argument("--time-every", type=int, default=16, help="every k-th K1 launch of the timed region carries a HIP event pair (its dispatch begin / end timestamps)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = os.environ.get("RPE_BENCH_FORCE_DIST") == "1"   # exercise the collective path with one rank
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    # Load librgbdpose_hip.so (and let it register its gfx950 code objects) BEFORE torch initialises HIP: measured on
    # MI355X / ROCm 7.x, a library whose fat binary is registered after hipInit() pays 2-3x the launch latency per
    # kernel (62 us vs 20 us per step here).  torch's bundled libamdhip64 is the one runtime of the process (_lib.py).
    from rgbd_pose_estimation_amd import _lib as L, api
    L.lib()
    n_dev = L.device_count()
    import torch
    import torch.distributed as dist
    from rgbd_pose_estimation_amd.distributed import HipShard, ShardedGaussNewton, init_native_comm, init_p2p

    if n_dev < 1 or not torch.cuda.is_available():
        sys.exit("bench.py: no MI355X visible (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("RPE_BENCH_BACKEND", "nccl")
    cdev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"   # where tensors handed to torch.distributed live
    if world > 1 or force_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        # plumbing backend: "nccl" (= RCCL) in production; "gloo" lets the tests run two ranks on ONE GPU (RCCL refuses that)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n = args.n_per_gpu
    sc = make_shard(rank, n)
    dist_path = world > 1 or force_dist
    # the collective needs the kernel on the stream torch orders the all-reduce after; one GPU uses the library's own stream
    stream = torch.cuda.Stream(device=local_rank) if dist_path else None
    import contextlib
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        shard = HipShard(local_rank, stream)
        ctx = shard.ctx
        ctx.load(L.F32, xw=sc.Q, xc=sc.P)
        R0, t0 = initial_pose(sc)
        from rgbd_pose_estimation_amd.api import pose12, pose7_from_Rt
        # untimed prologue: one scoring pass at the initial pose writes the 3D-3D inlier mask (K4b), as RANSAC would
        q0 = pose7_from_Rt(R0, t0, L.F32)
        inl = ctx.inlier_mask(L.VOTE_33, q0, thre_3d=THRE_3D)
        shard.kind, shard.flags = L.RES_P2P, L.USE_MASK
        gn = ShardedGaussNewton(shard.normal_eq)
        pose = pose12(R0, t0)
        # N > 1, in order of preference (every rank takes the same branch; RPE_BENCH_COLLECTIVE=p2p|rccl|torch forces one):
        #   p2p   the kernel's last workgroup exchanges the 32-double record with its peers over xGMI (HIP IPC mailboxes) and sums
        #         in rank order: the sharded step is ONE launch, no collective library on the critical path
        #   rccl  kernel -> in-place ncclAllReduce on the library's own communicator -> publish kernel
        #   torch kernel -> torch.distributed.all_reduce -> .cpu()   (fallback)
        want = os.environ.get("RPE_BENCH_COLLECTIVE", "torch" if os.environ.get("RPE_BENCH_TORCH_ALLREDUCE") == "1" else "auto")
        p2p = native = False
        if dist_path:
            # every rank first runs the kernel on its own (code-object load, clock ramp: the first launches of a process on a cold box
            # can take seconds), so that the ranks enter the first exchange together and not one of them seconds late
            for _ in range(200):
                ctx.normal_eq(L.RES_P2P, pose12(R0, t0), L.USE_MASK)
            dist.barrier()
        if dist_path and want in ("auto", "p2p"):
            p2p = init_p2p(ctx)
            if p2p:   # trust, but verify once against the collective library before timing anything
                chk = pose12(R0, t0)
                rec = np.zeros(32)
                dist.barrier()   # ranks arrive here at different times: the exchange inside the kernel has a bounded wait
                try:
                    L.check(L.lib().rpe_gn_step_dist(ctx._h, L.RES_P2P, L.USE_MASK, chk.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p), None))
                    delivered = 1
                except L.RpeError as e:   # e.g. a peer's record never arrived (bounded wait inside the kernel)
                    delivered = 0
                    print(f"[bench] rank {rank}: peer-to-peer step failed: {e}", file=sys.stderr, flush=True)
                ref = shard.normal_eq(pose12(R0, t0)).clone().to(cdev)
                dist.all_reduce(ref)
                ref = ref.cpu().numpy()
                good = int(delivered and np.all(np.abs(rec[:29] - ref[:29]) <= 1e-9 * (1.0 + np.abs(ref[:29]))))
                flag = torch.tensor([good], dtype=torch.int32, device=cdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    if rank == 0:
                        print("[bench] peer-to-peer record differs from the all-reduced one: falling back to RCCL", file=sys.stderr, flush=True)
                    ctx.p2p_destroy()
                    p2p = False
        if dist_path and not p2p and want == "p2p" and os.environ.get("RPE_BENCH_STRICT_COLLECTIVE") == "1":
            sys.exit("bench.py: the peer-to-peer collective was requested strictly and is not available")   # every rank agrees on p2p (MIN flag): all exit
        rccl_ok = False
        if dist_path and not (want == "p2p" and os.environ.get("RPE_BENCH_STRICT_COLLECTIVE") == "1") and want in ("auto", "rccl", "p2p"):
            # with a peer-to-peer path in place rpe_gn_step_dist prefers it; the communicator is its stand-by
            rccl_ok = init_native_comm(ctx)
        if dist_path and p2p and rccl_ok and want == "auto":
            # both collectives work: keep the faster one (measured here, 400 steps each after a short settling run; every rank reaches
            # the same verdict because the times are max-reduced).  The wire cannot be exercised on a one-GPU box, so the choice is
            # made on the machine the benchmark actually runs on.
            def timed(k):
                q = pose12(R0, t0)
                ctx.gn_steps_dist(L.RES_P2P, q, 200, L.USE_MASK)
                dist.barrier()
                t0_ = time.perf_counter()
                ctx.gn_steps_dist(L.RES_P2P, q, k, L.USE_MASK)
                tt = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=cdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt.item()) / k
            t_p2p = timed(400)
            L.check(L.lib().rpe_p2p_pause(ctx._h, 1))      # stand-by: the same call now takes the RCCL path
            t_rccl = timed(400)
            L.check(L.lib().rpe_p2p_pause(ctx._h, 0))
            if rank == 0:
                print(f"[bench] sharded step: peer-to-peer {t_p2p * 1e6:.1f} us, RCCL {t_rccl * 1e6:.1f} us", file=sys.stderr, flush=True)
            if t_rccl < t_p2p:
                dist.barrier()
                ctx.p2p_destroy()
                p2p = False
        native = rccl_ok or p2p
        collective = "none" if not dist_path else ("peer-to-peer exchange of 32 fp64 per step inside the kernel (xGMI, HIP IPC mailboxes)" if p2p else
                                                   "all-reduce(sum) of 32 fp64 per step over RCCL, " + ("library-owned communicator" if native else "torch.distributed"))

        # The host side of the loop (launch, wait for the published record, 6x6 solve, SE(3) exp-map update) is C++ inside the
        # library -- rpe_gn_refine / rpe_gn_steps_dist with tol = 0 run exactly k iterations -- so that the timed region contains
        # no Python per step (a ctypes call per step costs 2-3 us of the ~18).  The torch fallback keeps its Python loop.
        def run_steps(p, k):
            if k <= 0:
                return p
            if not dist_path:
                q, its, _, _ = ctx.gn_refine([L.RES_P2P], p, None, L.USE_MASK, k, 0.0)   # k x (kernel + publish + solve + exp-map)
                assert its == k
                return q
            if native:
                ctx.gn_steps_dist(L.RES_P2P, p, k, L.USE_MASK)   # k x (kernel [+ peer exchange | RCCL all-reduce + publish] + solve + exp-map)
                return p
            for _ in range(k):
                p = gn.step(p)[0]
            return p

        # untimed pre-warm, independent of --warmup: the first ~25 ms of launches after torch has initialised HIP contain a
        # one-off ~35 ms stall (measured; runtime lazy initialisation), and the FIRST process on a freshly booted box runs its
        # kernels ~25 % slower for about a second (measured: 16.4 us vs 12.7 us per launch; 1 s of pre-warm removes it) --
        # neither may land in the timed region
        if dist_path:
            pose = run_steps(pose, int(os.environ.get("RPE_BENCH_PREWARM_STEPS", "60000")))   # a FIXED count: every rank must issue the same number of collective steps
        else:
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < float(os.environ.get("RPE_BENCH_PREWARM_S", "1.5")):
                pose = run_steps(pose, 500)
        pose = run_steps(pose12(R0, t0), args.warmup)
        if args.time_every > 0:
            ctx.timing_enable(args.steps // args.time_every + 1, args.time_every)   # HIP events around every time_every-th K1 launch
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        chunk_t = []
        if args.debug_chunks:
            for k in range(0, args.steps, 250):
                pose = run_steps(pose, min(250, args.steps - k))
                chunk_t.append(time.perf_counter())
        else:
            pose = run_steps(pose, args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
        cnt, k_total_ms, k_min_ms = ctx.timing_collect()
        ctx.timing_enable(0, 1)
        ev_avg_ms, ev_min_ms = ctx.timing_calibrate(200)   # what an empty event pair reads: the marker latency inside every interval

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # extra on the collective path (every rank takes part): the sharded device-resident loop -- exchange + sum + 6x6 solve + exp-map
    # in the kernel's last workgroup, one launch per iteration on every GPU, one host wait per refinement
    sharded_loop = None
    if world > 1 and p2p and not args.no_extras:
        try:
            K = 500
            ctx.gn_refine_device([(L.RES_P2P, 1.0)], pose12(R0, t0), L.USE_MASK, 50, 0.0)
            dist.barrier()
            t0d = time.perf_counter()
            pd, itd, _, _ = ctx.gn_refine_device([(L.RES_P2P, 1.0)], pose12(R0, t0), L.USE_MASK, K, 0.0)
            td = torch.tensor([time.perf_counter() - t0d], dtype=torch.float64, device=cdev)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            dtd = float(td.item())
            sharded_loop = {"value": float(n) * world * K / dtd, "unit": "correspondence-residuals/s", "us_per_iteration": dtd / K * 1e6, "iterations": itd,
                            "rot_rad_vs_host_loop": rot_err(pd[:9].reshape(3, 3), pose[:9].reshape(3, 3)),
                            "note": "rpe_gn_refine_device after rpe_p2p_init: one launch per iteration on every GPU; beside, not instead of, the headline"}
        except Exception as e:
            sharded_loop = {"error": repr(e)}

    if rank == 0:
        total = float(n) * world * args.steps
        k_avg_s = (k_total_ms / max(cnt, 1)) * 1e-3
        achieved = BYTES_PER_CORR * n / k_avg_s / 1e9 if k_avg_s > 0 else None
        traffic = None
        rocprof_avg_us = None
        try:   # the committed rocprofv3 --kernel-trace --stats summary of this command (profiles/), for side-by-side reading
            import csv
            with open(os.path.join(ROOT, "profiles", "r01_bench_rocprofv3_kernel_stats.csv")) as fh:
                for row in csv.DictReader(fh):
                    if "normal_eq_kernel<float, 0" in row["Name"]:
                        rocprof_avg_us = float(row["AverageNs"]) * 1e-3
        except Exception:
            rocprof_avg_us = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("normal_eq_p2p_f32_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "correspondence-residuals/sec", "value": total / elapsed, "unit": "correspondence-residuals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: 640x480 dense depth, {n} 3D-3D correspondences per GPU, point-to-point absolute "
                                   "orientation, Gauss-Newton step (K1 normal equations + host SE3 exp-map update) over the RANSAC inlier mask",
                       "corr_per_gpu": n, "global_corr": n * world, "inliers_rank0": int(inl), "accumulate": "fp64",
                       "collective": collective},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "rpe::normal_eq_kernel<float, 0>", "algorithmic_bytes_per_launch": BYTES_PER_CORR * n,
                         "avg_launch_us": k_avg_s * 1e6, "min_launch_us": k_min_ms * 1e3, "launches_timed": cnt,
                         "empty_event_pair_us": ev_avg_ms * 1e3, "rocprofv3_avg_launch_us": rocprof_avg_us,
                         "note": "HIP events on the kernel's own stream: every time_every-th stage-1 launch of the timed region is issued with "
                                 "hipExtLaunchKernelGGL(start, stop), so the pair holds that dispatch's begin / end timestamps (the quantity rocprofv3 "
                                 "reports; rocprofv3_avg_launch_us is its average for the same command from the committed profiles/ summary; two "
                                 "marker packets around the launch would add their own latency, empty_event_pair_us); 7.99 MB working set is "
                                 "L2/Infinity-Cache resident after the first step, so this is not an HBM-streaming figure (see DESIGN.md)"},
        }
        # pose parity: converged GN pose vs the CPU oracle's closed form (shinji, fp64, same fp32 inputs, same inlier set)
        if world == 1:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib as O
                m = ctx.download_mask(L.MOD_33)
                Ro, to, _ = O.shinji_f32in_f64(sc.Q[m == 1], sc.P[m == 1])
                out["pose_error_vs_cpu"] = {"rot_rad": rot_err(pose[:9].reshape(3, 3), Ro),
                                            "trans_rel": float(np.linalg.norm(pose[9:] - to) / np.linalg.norm(to)),
                                            "tolerance": {"rot_rad": 1e-5, "trans_rel": 1e-4},
                                            "reference": "oracle shinji() fp64 on the same fp32 inputs and inlier set"}
            except Exception as e:
                out["pose_error_vs_cpu"] = {"error": repr(e)}
            out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(sc, args.cpu_seconds)
            # extra (not the headline: the north star keeps the exp-map on the host): the same iterations with the 6x6 solve and
            # the SE(3) update done by the kernel's last workgroup, one launch per iteration, one host wait at the end
            try:
                if args.no_extras:
                    raise StopIteration
                K = 500
                p0d = pose12(R0, t0)
                ctx.gn_refine_device([(L.RES_P2P, 1.0)], p0d, L.USE_MASK, 50, 0.0)
                t0d = time.perf_counter()
                pd, itd, _, _ = ctx.gn_refine_device([(L.RES_P2P, 1.0)], p0d, L.USE_MASK, K, 0.0)
                dtd = time.perf_counter() - t0d
                out["device_resident_loop"] = {"value": n * K / dtd, "unit": "correspondence-residuals/s", "us_per_iteration": dtd / K * 1e6,
                                               "iterations": itd, "rot_rad_vs_host_loop": rot_err(pd[:9].reshape(3, 3), pose[:9].reshape(3, 3)),
                                               "note": "rpe_gn_refine_device: solve + exp-map on the GPU; reported beside, not instead of, the host-update headline"}
            except StopIteration:
                pass
            except Exception as e:
                out["device_resident_loop"] = {"error": repr(e)}
            # extra: the step BEFORE the path -- two rendered 640x480 depth frames of the reference camera,
            # dense projective ICP with association and normal equations fused in one kernel per round, pose kept on the GPU
            try:
                if args.no_extras:
                    raise StopIteration
                from rgbd_pose_estimation_amd import simulator as S
                Ra, ta = S._rot_zyx(0.05, -0.1, 0.02), np.array([0.1, -0.05, 0.2])
                dR = S._rot_zyx(0.02, -0.015, 0.01)
                Rb, tb = dR @ Ra, dR @ ta + np.array([0.03, -0.02, 0.025])
                fctx = api.Context(local_rank)
                fctx.frame_set_depth(S.render_depth(Ra, ta, as_u16=True), S.DEFAULT_CAMERA, 0.001, 0.1, 10.0, 0.1)
                fctx.model_from_frame(pose12(Ra, ta))
                fctx.frame_set_depth(S.render_depth(Rb, tb, as_u16=True), S.DEFAULT_CAMERA, 0.001, 0.1, 10.0, 0.1)
                K, reps = 20, 25
                fctx.icp(pose12(Ra, ta), L.RES_P2PLANE, K, 0.0, 0.15, 0.8, device_resident=True, fused=True)
                t0i = time.perf_counter()
                for _ in range(reps):
                    pi, iti, _, _, pairs = fctx.icp(pose12(Ra, ta), L.RES_P2PLANE, K, 0.0, 0.15, 0.8, device_resident=True, fused=True)
                dti = (time.perf_counter() - t0i) / reps
                out["icp_frame_loop"] = {"value": pairs * K / dti, "unit": "pixel-residuals/s", "us_per_round": dti / K * 1e6, "rounds": K,
                                         "pairs": pairs, "pixels": 307200, "rot_err_rad_vs_truth": rot_err(pi[:9].reshape(3, 3), Rb),
                                         "trans_err_m_vs_truth": float(np.linalg.norm(pi[9:] - tb)),
                                         "note": "rpe_icp fused + device-resident: projective association + point-to-plane normal equations in one kernel per round"}
                fctx.close()
            except StopIteration:
                pass
            except Exception as e:
                out["icp_frame_loop"] = {"error": repr(e)}
            # extra: the other half of the hot path -- batched RANSAC hypothesis scoring (vote loops V1/V2, kernel K4) on the same frame:
            # 512 hypotheses per pass; the CPU figure is the oracle's vote loop (AOOnlyPoseAdapter virtual getters, 1 thread)
            try:
                if args.no_extras:
                    raise StopIteration
                rng_s = np.random.default_rng(3)
                H = 512
                dq = 0.002 * rng_s.standard_normal((H, 7))
                base7 = pose7_from_Rt(sc.R, sc.t, L.F32)
                poses = base7[None, :] + dq
                poses[:, :4] /= np.linalg.norm(poses[:, :4], axis=1, keepdims=True)
                poses = np.ascontiguousarray(poses.astype(np.float32).astype(np.float64))
                res = {}
                for mode, name in ((L.SCORE_FAST, "fast"), (L.SCORE_EXACT, "exact")):
                    t_w = time.perf_counter()
                    while time.perf_counter() - t_w < 0.2:   # settle: freeing the ICP context above stalls the next launches for ~0.1 s once
                        ctx.score(L.VOTE_33, poses, THRE_3D, mode=mode)
                    t0s = time.perf_counter()
                    reps = 20
                    for _ in range(reps):
                        v = ctx.score(L.VOTE_33, poses, THRE_3D, mode=mode)
                    dts = (time.perf_counter() - t0s) / reps
                    res[name] = {"corr_hyp_per_s": n * H / dts, "us_per_pass": dts * 1e6}
                    if mode == L.SCORE_EXACT:
                        v_exact = v
                cpu = None
                if not args.no_cpu_baseline:
                    import oracle_lib as O
                    votes_cpu = np.zeros(8, np.int32)
                    xw32, xc32 = np.ascontiguousarray(sc.Q, np.float32), np.ascontiguousarray(sc.P, np.float32)
                    p8 = np.ascontiguousarray(poses[:8])
                    O.lib().orc_time_votes33.restype = C.c_double
                    dtc = O.lib().orc_time_votes33(xw32.ctypes.data_as(C.c_void_p), xc32.ctypes.data_as(C.c_void_p), n, p8.ctypes.data_as(C.c_void_p), 8,
                                                    C.c_float(THRE_3D), votes_cpu.ctypes.data_as(C.c_void_p))
                    cpu = {"corr_hyp_per_s": n * 8 / dtc, "cores": 1, "sample": "8 hypotheses through the oracle's vote loop",
                           "votes_equal_exact_mode": bool(np.array_equal(votes_cpu, v_exact[:8]))}
                out["ransac_scoring"] = {"hypotheses": H, "kind": "3D-3D (V1/V2)", "fast": res["fast"], "exact": res["exact"], "cpu_port": cpu,
                                         "note": "wall time per rpe_score call incl. pose upload and vote read-out; beside, not instead of, the headline"}
            except StopIteration:
                pass
            except Exception as e:
                out["ransac_scoring"] = {"error": repr(e)}
        else:
            if sharded_loop is not None:
                out["device_resident_loop"] = sharded_loop
            out["pose_error_vs_truth"] = {"rot_rad": rot_err(pose[:9].reshape(3, 3), sc.R), "trans_abs_m": float(np.linalg.norm(pose[9:] - sc.t))}
            out["cpu_baseline"] = None
        if args.debug_chunks:
            ts = [t_start] + chunk_t
            out["chunk_us_per_step"] = [round((b - a) / 250 * 1e6, 1) for a, b in zip(ts[:-1], ts[1:])]
    # tear everything down first: RCCL prints its version banner on stdout around communicator life-cycle events, and
    # the JSON line must be the LAST line rank 0 prints
    if world > 1:
        dist.barrier()   # no rank may unmap its mailbox / destroy its communicator while a peer can still reach it
    ctx.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    try:  # RCCL's banner sits in the C stdio buffer until exit: push it out before the JSON line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
