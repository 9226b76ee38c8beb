"""rpe_prosac_order by itself (the PROSAC weight order of a dense frame): wall time per call against the host's partial sort, and -- under
rocprofv3 --kernel-trace --stats -- the four kernels behind it.   usage: prosac_order_probe.py [n] [top_k]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
top_k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(3)
w = rng.random(n).astype(np.float32)
ctx = api.Context(0)
lib = L.lib()
order = np.zeros(top_k, np.int32)
_p = lambda a: a.ctypes.data_as(C.c_void_p)
def dev():
    L.check(lib.rpe_prosac_order(ctx._h, _p(w), n, top_k, _p(order)))
def host():
    idx = np.argpartition(-w, top_k)[:top_k]
    return idx[np.lexsort((idx, -w[idx]))]
def best(f, reps=20):
    f(); f()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return round(b * 1e6, 1)
out = {"n": n, "top_k": top_k, "device_us": best(dev), "numpy_partition_sort_us": best(host)}
ref = host()
out["equal"] = bool(np.array_equal(order, ref.astype(np.int32)))
print(json.dumps(out))
