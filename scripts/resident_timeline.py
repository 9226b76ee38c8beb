"""Where one iteration of the RESIDENT Gauss-Newton kernel spends its time (development aid; diagnostic build -DRPE_STAMPS): thread 0 of
every workgroup stamps the 100 MHz clock in iteration 1000 of a 2000-iteration refinement -- pose seen, slice done, granules stored,
(collecting workgroup 0) its granule read, all read, run record stored towards the host.  Reported relative to the first workgroup's "pose seen"."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(n, kind):
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from tail_timeline import scene
    lib = L.lib()
    lib.rpe_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    R, t, arrs = scene(n)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    ctx.gn_refine([kind], p, max_iter=300, tol=0.0)
    buf = np.zeros(4096 * 16, np.uint64)
    lib.rpe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
    rows = []
    for _ in range(30):
        ctx.gn_refine([kind], p, max_iter=2000, tol=0.0)
        lib.rpe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
        s = buf.reshape(4096, 16).astype(np.int64)
        live = s[:, 0] > 0
        s = s[live]
        t0 = s[:, 0].min()
        rel = lambda c: (s[:, c] - t0) * 0.01
        row = dict(G=int(live.sum()), pose_seen_med=float(np.median(rel(0))), pose_seen_last=float(rel(0).max()), body_done_med=float(np.median(rel(1))),
                   body_done_last=float(rel(1).max()), stored_med=float(np.median(rel(2))), stored_last=float(rel(2).max()))
        w0 = s[s[:, 5] > 0]
        if len(w0):
            for c, name in ((0, "wg0_pose_seen"), (1, "wg0_body_done"), (2, "wg0_reduced"), (3, "wg0_granules_read"), (4, "wg0_barrier"), (5, "wg0_record_stored")):
                row[name] = float((w0[0, c] - t0) * 0.01)
        rows.append(row)
    keys = sorted({k for r in rows for k in r})
    print(json.dumps(dict(what="resident_stamps", rows=os.environ.get("RPE_RESIDENT_ROWS", "auto"), n=n, kind=kind, unit="us after the first workgroup saw the pose; medians over 30 refinements",
                          **{k: float(np.median([r[k] for r in rows if k in r])) for k in keys})), flush=True)
    ctx.close()


def worker_auto(n, kind):
    """the AUTONOMOUS loop (rpe_gn_refine_device, one launch): stamps 0 iteration starts | 1 slice done | 2 granules stored | 3 run collected and
    run record stored (collecting workgroups) | 4 every run record read | 5 record expanded | 6 solve + update done"""
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from tail_timeline import scene
    lib = L.lib()
    lib.rpe_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    R, t, arrs = scene(n)
    ctx = api.Context(0).load(L.F32, **arrs)
    p = api.pose12(R, t)
    ctx.gn_refine_device([(kind, 1.0)], p, 0, 300, 0.0)
    buf = np.zeros(4096 * 16, np.uint64)
    lib.rpe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
    rows = []
    for _ in range(30):
        ctx.gn_refine_device([(kind, 1.0)], p, 0, 2000, 0.0)
        lib.rpe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
        s = buf.reshape(4096, 16).astype(np.int64)
        s = s[s[:, 0] > 0]
        t0 = s[:, 0].min()
        rel = lambda c: (s[:, c] - t0) * 0.01
        names = ("iteration_starts", "slice_done", "granules_stored", "collected", "run_records_read", "record_expanded", "solved")
        row = dict(G=len(s))
        for c, nm in enumerate(names):
            # with a solving workgroup beside the grid (frame-sized problems) the workers run only the first stages: a stage nobody stamped
            # (all zeros) is left out instead of being reported relative to t0
            if not (s[:, c] > 0).all():
                continue
            row[nm + "_med"] = float(np.median(rel(c))); row[nm + "_last"] = float(rel(c).max())
        rows.append(row)
    keys = sorted({k for r in rows for k in r})
    print(json.dumps(dict(what="autonomous_resident_stamps", n=n, kind=kind, unit="us after the first workgroup began iteration 1000; medians over 30 loops",
                          **{k: float(np.median([r[k] for r in rows if k in r])) for k in keys})), flush=True)
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker-auto":
        worker_auto(int(sys.argv[2]), int(sys.argv[3]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]))
    else:
        from rgbd_pose_estimation_amd import build as B
        so = os.path.join(ROOT, "rgbd_pose_estimation_amd", "lib", "librgbdpose_hip_stamps1.so")
        if not os.path.exists(so):
            so = B.build_stamps(1)
        for n, kind in ((307200, 0), (1000000, 1)):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(n), str(kind)], env=dict(os.environ, RPE_LIBRARY=so), check=False)
        for n, kind in ((307200, 0), (1000, 0)):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--worker-auto", str(n), str(kind)], env=dict(os.environ, RPE_LIBRARY=so), check=False)
