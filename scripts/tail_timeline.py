"""Per-phase timeline of the normal-equation kernel and A/B timing of its tail variants (development aid, run on the GPU box).

  python scripts/tail_timeline.py            -> JSON lines on stdout:
     * "stamps": the DIAGNOSTIC build (-DRPE_STAMPS, rgbd_pose_estimation_amd/build.py:build_stamps) stamps the 100 MHz clock in thread 0
       of every workgroup at the phase boundaries; reported per launch relative to the earliest workgroup start, medians over launches
     * "timing": the PRODUCT build with its default stage for host-consumed results (collecting workgroups + host-side final sum) and,
       under RPE_COLLECT=0, the arrival-counter tails RPE_TAIL = 0 (all records summed by the last workgroup), 1 (per-shard sums first),
       2 (0 with one batch of loads), +8 (pipelined loop even for one group per thread): dispatch-timestamp kernel time and the wall
       time per Gauss-Newton step of the library's launch-per-step host loop (RPE_RESIDENT=0)
Every variant runs in its own process (the knobs are read once per process)."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = ["start", "body_done", "wave_reduced", "wg_barrier1", "record_stored", "wg_barrier2", "arrived", "last_known", "records_summed",
          "published", "flag_stored", "loads_landed", "top_last_known", "shards_summed"]


def scene(n):
    import numpy as np
    from rgbd_pose_estimation_amd import simulator as S
    rng = np.random.default_rng(1)
    R, t = S.random_pose(rng)
    base = S.simulate_2d_3d_nl_correspondences(rng, R, t, min(n, 1_000_000), 1.0, 0.0, 0.02, 0.0, 0.03, 0.0).astype(np.float32)
    reps = (n + len(base.Q) - 1) // len(base.Q)
    tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    return R, t, dict(xw=tile(base.Q), xc=tile(base.P), nc=tile(base.N))


def worker_timing(n, kind, steps):
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    R, t, arrs = scene(n)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    for _ in range(200):
        ctx.normal_eq(kind, p)
    ctx.timing_enable(steps, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.normal_eq(kind, p)
    wall_py = (time.perf_counter() - t0) / steps
    cnt, tot, mn = ctx.timing_collect()
    ctx.timing_enable(0, 1)
    # the library's own host loop (no Python per step): tol = 0 -> exactly `steps` steps
    p0 = np.array(p)
    t0 = time.perf_counter()
    ctx.gn_refine([kind], p0, max_iter=2000, tol=0.0)
    wall_loop = (time.perf_counter() - t0) / 2000
    bpc = 36 if kind == L.RES_P2PLANE else 24
    print(json.dumps(dict(what="timing", tail=os.environ.get("RPE_TAIL", "0"), collect=os.environ.get("RPE_COLLECT", "1"), block=os.environ.get("RPE_BLOCK", ""), groups=os.environ.get("RPE_REDUCE_GROUPS", ""),
                          n=n, kind=kind, kernel_avg_us=tot / cnt * 1e3, kernel_min_us=mn * 1e3, wall_us_python_step=wall_py * 1e6, wall_us_library_step=wall_loop * 1e6,
                          achieved_GBs=bpc * n / (tot / cnt * 1e-3) / 1e9)), flush=True)
    ctx.close()


def worker_stamps(n, kind, launches):
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    lib = L.lib()
    lib.rpe_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    R, t, arrs = scene(n)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    for _ in range(50):
        ctx.normal_eq(kind, p)
    buf = np.zeros(4096 * 16, np.uint64)
    lib.rpe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
    rows = []
    steady = int(os.environ.get("RPE_STEADY", "1"))   # launches issued back to back before the stamps of the LAST one are read (no idle gap before it)
    for _ in range(launches):
        for _ in range(steady):
            ctx.normal_eq(kind, p)
        lib.rpe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
        s = buf.reshape(4096, 16).astype(np.int64)
        live = s[:, 0] > 0
        G = int(live.sum())
        s = s[live]
        t0 = s[:, 0].min()
        rel = lambda col, rows_=slice(None): (s[rows_, col] - t0) * 0.01  # 100 MHz ticks -> us
        row = dict(G=G, start_last=float(rel(0).max()))
        for k, name in ((1, "body_done"), (2, "wave_reduced"), (3, "wg_barrier1"), (4, "record_stored"), (5, "wg_barrier2"), (6, "arrived")):
            row[name + "_med"] = float(np.median(rel(k)))
            row[name + "_max"] = float(rel(k).max())
        if (s[:, 11] > 0).any():
            row["loads_landed_med"] = float(np.median(rel(11)))
            row["loads_landed_max"] = float(rel(11).max())
        lead = s[:, 9] > 0   # collecting workgroups (collect_and_send): 4 = own sums stored / kept, 7 = its granules read, 8 = all read, 9 = run record sent
        if lead.any() and not (s[:, 10] > 0).any():
            row["collecting_workgroups"] = int(lead.sum())
            row["granules_stored_med"] = float(np.median(rel(4)))
            row["granules_stored_max"] = float(rel(4).max())
            row["run_read_max"] = float(((s[lead, 8] - t0) * 0.01).max())
            row["run_record_sent_max"] = float(((s[lead, 9] - t0) * 0.01).max())
            for k in ("record_stored_med", "record_stored_max", "wg_barrier2_med", "wg_barrier2_max", "arrived_med", "arrived_max"):
                row.pop(k, None)
        fin = s[:, 10] > 0   # the workgroup that published
        if fin.any():
            i = int(np.argmax(fin))
            for k, name in ((0, "last_wg_start"), (1, "last_wg_body_done"), (6, "last_wg_arrived"), (7, "last_known"), (8, "records_summed"), (12, "top_last_known"),
                            (13, "shards_summed"), (9, "published"), (10, "flag_stored")):
                if s[i, k] > 0:
                    row[name] = float((s[i, k] - t0) * 0.01)
        rows.append(row)
    keys = sorted({k for r in rows for k in r})
    med = {k: float(np.median([r[k] for r in rows if k in r])) for k in keys}
    print(json.dumps(dict(what="stamps", level=os.environ.get("RPE_STAMP_LEVEL", "1"), tail=os.environ.get("RPE_TAIL", "0"), collect=os.environ.get("RPE_COLLECT", "1"), n=n, kind=kind, launches=launches,
                          unit="us after the first workgroup's start; medians over launches", **med)), flush=True)
    ctx.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        what, n, kind = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        (worker_timing if what == "timing" else worker_stamps)(n, kind, int(sys.argv[5]))
        return
    from rgbd_pose_estimation_amd import build as B
    cases = [(307200, 0), (1000000, 1), (1250000, 0)]
    def run(env, what, n, kind, count):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", what, str(n), str(kind), str(count)], env=e, check=False)
    for level in (1, 2):
        so = os.path.join(ROOT, 'rgbd_pose_estimation_amd', 'lib', f'librgbdpose_hip_stamps{level}.so')
        if not os.path.exists(so):
            so = B.build_stamps(level)
        for n, kind in cases[:2]:   # the product's stage for host-consumed results: collecting workgroups + host-side final sum
            run({"RPE_LIBRARY": so, "RPE_STAMP_LEVEL": str(level)}, "stamps", n, kind, 200)
        for tail in ("0", "1", "2"):   # the arrival-counter tails (device / collective targets; RPE_COLLECT=0 forces them for host results too)
            for n, kind in cases[:2]:
                run({"RPE_LIBRARY": so, "RPE_TAIL": tail, "RPE_COLLECT": "0", "RPE_STAMP_LEVEL": str(level)}, "stamps", n, kind, 200)
    for rep in range(2):
        for n, kind in cases:
            run({"RPE_RESIDENT": "0"}, "timing", n, kind, 2000)
        for tail in ("0", "1", "2", "8", "10"):
            for n, kind in cases:
                run({"RPE_TAIL": tail, "RPE_COLLECT": "0", "RPE_RESIDENT": "0"}, "timing", n, kind, 2000)
    for blk, groups in (("256", "1"), ("1024", "1"), ("512", "2"), ("256", "2")):
        for n, kind in cases:
            run({"RPE_BLOCK": blk, "RPE_REDUCE_GROUPS": groups, "RPE_RESIDENT": "0"}, "timing", n, kind, 2000)

if __name__ == "__main__":
    main()
