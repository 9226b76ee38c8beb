"""Where ao() (Library.cpp:17, host arrays in) spends its wall time at 640 x 480: the whole call against its parts through the context
API -- set_problem + two uploads (+ synchronise), the moments kernel + wait, the host SVD.   usage: ao_breakdown.py [n]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RPE_QUIET"] = "1"
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
rng = np.random.default_rng(1)
R, t = S.random_pose(rng)
sc = S.simulate_3d_3d_correspondences(rng, R, t, n, 0.02, 0.0).astype(np.float32)
Q, P = np.ascontiguousarray(sc.Q), np.ascontiguousarray(sc.P)

def best(f, reps=10):
    f(); f()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return round(b * 1e6, 1)

out = {"n": n, "MB": 2 * Q.nbytes / 1e6}
out["ao_total_us"] = best(lambda: api.ao(Q, P))
ctx = api.Context(0)
lib = L.lib()
def upload_only():
    ctx.load(L.F32, xw=Q, xc=P); L.check(lib.rpe_synchronize(ctx._h))
out["set_problem_two_uploads_sync_us"] = best(upload_only)
out["moments_kernel_and_wait_us"] = best(lambda: ctx.p2p_moments(0))
Q2, P2 = Q.copy(), P.copy()
out["host_memcpy_of_both_arrays_us"] = best(lambda: (np.copyto(Q2, Q), np.copyto(P2, P)))
out["ao_ransac_total_us"] = best(lambda: api.ao_ransac(Q, P), 5)
print(json.dumps(out))
