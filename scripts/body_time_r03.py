"""How long the streaming loop of the one-launch normal-equation kernels takes INSIDE a workgroup (development aid; diagnostic build
-DRPE_STAMPS): per workgroup, body_done - start of thread 0 on the 100 MHz clock, per residual kind and launch geometry.  Separates
the loop (issue- or bandwidth-bound) from the dispatch ramp and the cross-workgroup tail."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(n):
    import numpy as np
    from rgbd_pose_estimation_amd import _lib as L, api
    import bench
    lib = L.lib()
    lib.rpe_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    R, t, Q, P, Nn = bench.cheap_scene(n, seed=9)
    U = Q @ R.T.astype(np.float32) + t.astype(np.float32)
    U = (U / np.linalg.norm(U, axis=1, keepdims=True)).astype(np.float32)
    ctx = api.Context(0).load(L.F32, xw=Q, xc=P, bv=U, nc=Nn)
    p = api.pose12(R, t)
    buf = np.zeros(4096 * 16, np.uint64)
    for name, kind in (("p2p", L.RES_P2P), ("p2plane", L.RES_P2PLANE), ("bearing", L.RES_BEARING)):
        for _ in range(20):
            ctx.normal_eq(kind, p)
        lib.rpe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
        body, ramp, total = [], [], []
        for _ in range(40):
            for _ in range(12):   # back-to-back launches: the stamps read below are the LAST one's (steady state, no idle gap before it)
                ctx.normal_eq(kind, p)
            lib.rpe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
            s = buf.reshape(4096, 16).astype(np.int64)
            s = s[s[:, 0] > 0]
            body.append(np.median(s[:, 1] - s[:, 0]) * 0.01)
            ramp.append((s[:, 0].max() - s[:, 0].min()) * 0.01)
            total.append((s[:, 1].max() - s[:, 0].min()) * 0.01)
        print(json.dumps(dict(n=n, kind=name, block=os.environ.get("RPE_BLOCK", "default"), max_blocks=os.environ.get("RPE_MAX_BLOCKS", "256"), workgroups=len(s),
                              loop_us_per_workgroup_median=float(np.median(body)), last_start_us=float(np.median(ramp)),
                              first_start_to_last_body_done_us=float(np.median(total)))), flush=True)
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]))
    else:
        from rgbd_pose_estimation_amd import build as B
        so = B.build_stamps(1)
        for n in (1000000, 2500000):
            for blk, cap in (("256", "256"), ("256", "512"), ("512", "256")):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(n)], env=dict(os.environ, RPE_LIBRARY=so, RPE_BLOCK=blk, RPE_MAX_BLOCKS=cap), check=False)
