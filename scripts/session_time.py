"""Per-call cost inside a resident scoring session (K4r): begin, then batches of 8 / 32 hypotheses and the masks, after a pause (the
kernel is resident and warm) -- against the same calls outside a session."""
import ctypes as C
import os
import json
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd_pose_estimation_amd import _lib as L, api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
rng = np.random.default_rng(0)
xw = rng.standard_normal((n, 3)).astype(np.float32) + np.float32([0, 0, 4])
xc = xw + 0.01 * rng.standard_normal((n, 3)).astype(np.float32)
bv = xc / np.linalg.norm(xc, axis=1, keepdims=True)
ctx = api.Context(0).load(L.F32, xw=xw, xc=xc, bv=bv.astype(np.float32))
lib = L.lib()
q = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (32, 1))
q[:, 4:] += 0.001 * rng.standard_normal((32, 3))
v = np.zeros(32, np.int32)
tot = C.c_int(0)
qp, vp = q.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)


def t(f, reps=200):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter_ns(); f(); ts.append(time.perf_counter_ns() - t0)
    ts.sort()
    return round(ts[len(ts) // 2] / 1e3, 2)


out = {"n": n}
for kind, name in ((L.VOTE_33, "33"), (L.VOTE_33_23, "33_23")):
    thr = (0.05, 0.9999, 2.0)
    for mode, mname in ((L.SCORE_EXACT, "exact"), (L.SCORE_FAST, "fast")):
        r = {}
        r["launch_score8"] = t(lambda: lib.rpe_score(ctx._h, kind, mode, qp, 8, *thr, vp))
        r["launch_score32"] = t(lambda: lib.rpe_score(ctx._h, kind, mode, qp, 32, *thr, vp))
        r["launch_mask"] = t(lambda: lib.rpe_inlier_mask(ctx._h, kind, mode, qp, *thr, C.byref(tot)))
        assert lib.rpe_score_session_begin(ctx._h, kind, mode, *thr) == 0
        time.sleep(0.001)
        r["session_score8"] = t(lambda: lib.rpe_score(ctx._h, kind, mode, qp, 8, *thr, vp))
        r["session_score32"] = t(lambda: lib.rpe_score(ctx._h, kind, mode, qp, 32, *thr, vp))
        r["session_mask"] = t(lambda: lib.rpe_inlier_mask(ctx._h, kind, mode, qp, *thr, C.byref(tot)))
        lib.rpe_score_session_end(ctx._h)

        def whole():
            lib.rpe_score_session_begin(ctx._h, kind, mode, *thr)
            lib.rpe_score(ctx._h, kind, mode, qp, 8, *thr, vp)
            lib.rpe_inlier_mask(ctx._h, kind, mode, qp, *thr, C.byref(tot))
            lib.rpe_score_session_end(ctx._h)
        r["session_begin_score8_mask_end"] = t(whole)

        def begin_end():
            lib.rpe_score_session_begin(ctx._h, kind, mode, *thr)
            lib.rpe_score_session_end(ctx._h)
        r["session_begin_end"] = t(begin_end)
        r["python_call_overhead"] = t(lambda: lib.rpe_score_session_end(ctx._h))
        out[name + "_" + mname] = r
print(json.dumps(out))
