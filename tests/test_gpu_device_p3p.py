"""-m gpu: the FAST-mode device generator of the P3P solvers (rpe_ransac_p3p_batch, csrc/rpe_hypotheses.hip): tolerance parity with the
host's hypotheses on the same sample stream (the vote-exact default never takes this path), and whole kneip_ransac /
shinji_kneip_ransac runs in FAST mode with and without it."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import util
from rgbd_pose_estimation_amd import _lib as L, api

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = (1 << 64) - 1


def rand31_state(seed, stream=54):
    """(state, inc) of rpe::Rand31 after reseed(seed) (pose/Utility.hpp): PCG32 seeding."""
    inc = ((stream << 1) | 1) & M64
    state = (0 * 6364136223846793005 + inc) & M64
    state = (state + seed) & M64
    state = (state * 6364136223846793005 + inc) & M64
    return state, inc


def rot_from_q(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("solver,method", [(0, api.M_KNEIP_RANSAC), (1, api.M_SK_RANSAC), (2, api.M_NL_KNEIP_RANSAC), (3, api.M_NL_SHINJI_RANSAC),
                                           (4, api.M_NL_SK_RANSAC)])
def test_device_p3p_hypotheses_follow_the_host_stream(gpu_ctx_factory, solver, method, f64):
    n, iters, seed = 5000, 400, 7
    dt = np.float64 if f64 else np.float32
    # (no invalid camera points here: on a sample that hits one the reference's nl_2p reuses whatever the previous iteration left in its
    # sample matrices, AbsoluteOrientationNormal.hpp:315,389 -- a junk hypothesis either way, but not the same junk)
    sc = util.scene_full(31, n, dt, n2d=2.0, n3d=0.01, nnl_deg=0.5, outliers=0.2, nan_frac=0.05 if solver < 2 else 0.0)
    full = dict(xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    keys = {0: dict(xw=sc.Q, bv=sc.U), 1: dict(xw=sc.Q, xc=sc.P, bv=sc.U), 2: full, 3: full, 4: full}[solver]
    hq7, hfirst = api.host_hypotheses(method, L.F64 if f64 else L.F32, iters=iters, seed=seed, **keys)
    ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, **keys)
    per = (1, 2, 1, 2, 3)[solver]
    votes = np.zeros(iters * per, np.int32); q7 = np.zeros((iters * per, 7)); valid = np.zeros(iters * per, np.uint8)
    state, inc = rand31_state(seed)
    cos_thr = float(np.cos(np.arctan(8.0 / 585.0)))
    cos_nl = float(np.cos(0.1))
    L.check(L.lib().rpe_ransac_p3p_batch(ctx._h, solver, C.c_uint64(state), C.c_uint64(inc), iters, 0.2, cos_thr, cos_nl, votes.ctypes.data_as(C.c_void_p),
                                         q7.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p)))
    # the same iterations yield the same NUMBER of hypotheses (validity decisions agree away from degenerate samples) ...
    dev_counts = valid.reshape(iters, per).sum(axis=1)
    host_counts = np.diff(hfirst)
    agree = dev_counts == host_counts
    assert agree.mean() > 0.97, (agree.mean(), dev_counts[:20], host_counts[:20])
    # ... and where they do, the poses agree to rounding (fp32 P3P on the host against fp64 on the device: 2e-2 rad worst case on
    # ill-conditioned samples, typically 1e-5; fp64 against fp64: 1e-6)
    rot_tol, med_tol = (5e-2, 2e-4) if not f64 else (1e-5, 1e-9)
    errs = []
    for i in np.nonzero(agree)[0]:
        hs = hq7[hfirst[i]:hfirst[i + 1]]
        ds = q7[i * per:(i + 1) * per][valid[i * per:(i + 1) * per] == 1]
        for h, d in zip(hs, ds):
            Rh, Rd = rot_from_q(h[:4] / np.linalg.norm(h[:4])), rot_from_q(d[:4] / np.linalg.norm(d[:4]))
            errs.append(util.rot_err(Rh, Rd))
    errs = np.array(errs)
    assert len(errs) > 0.9 * iters and np.median(errs) < med_tol and np.mean(errs < rot_tol) > 0.97, (np.median(errs), np.sort(errs)[-10:])
    # the votes are those rpe_score gives the same device poses in FAST mode
    sel = np.nonzero(valid)[0][:64]
    kind = (L.VOTE_23, L.VOTE_33_23, L.VOTE_NN_23, L.VOTE_NN_33, L.VOTE_NN_33_23)[solver]
    again = ctx.score(kind, q7[sel], 0.2, cos_thr, cos_nl, mode=L.SCORE_FAST)
    assert np.mean(np.abs(again - votes[sel]) <= 2) > 0.95   # the pose went through quaternion form and back: a vote may flip at a threshold


def test_fast_mode_pipelines_with_device_p3p(tmp_path):
    """kneip_ransac and shinji_kneip_ransac in FAST mode on a scene with 55 % outliers (hundreds of iterations: the later batches are
    generated on the device) against the same runs with host generation: both find the pose, consensus sizes within 2 %."""
    script = tmp_path / "p3p_pipelines.py"
    script.write_text(f"""
import sys, json
sys.path.insert(0, {repr(ROOT)}); sys.path.insert(0, {repr(os.path.join(ROOT, "tests"))})
import numpy as np
import util
from rgbd_pose_estimation_amd import _lib as L, api
out = {{}}
for f64 in (False, True):
    sc = util.scene_full(5, 40000, np.float64 if f64 else np.float32, n2d=2.0, n3d=0.02, outliers=0.55, nan_frac=0.0)
    all5 = ("xw", "xc", "bv", "nw", "nc")
    for name, m, keys in (("kneip", api.M_KNEIP_RANSAC, ("xw", "bv")), ("sk", api.M_SK_RANSAC, ("xw", "xc", "bv")), ("nlk", api.M_NL_KNEIP_RANSAC, all5),
                          ("nls", api.M_NL_SHINJI_RANSAC, all5), ("nlsk", api.M_NL_SK_RANSAC, all5)):
        data = dict(xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
        r = api.run(m, L.F64 if f64 else L.F32, thre_3d=0.1, thre_2d=6.0, thre_nl=0.1, iters=2000, confidence=0.9999, seed=3, score_mode=L.SCORE_FAST, **{{k: data[k] for k in keys}})
        out[name + ("64" if f64 else "32")] = dict(votes=int(r["max_votes"]), iters=int(r["iters"]), rot=float(util.rot_err(r["R"], sc.R)),
                                                    trans=float(np.linalg.norm(r["t"] - sc.t)))
print("RESULT " + json.dumps(out))
""")
    import json
    res = {}
    for tag, env in (("device", {}), ("host", {"RPE_HOST_HYPOTHESES": "1"})):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, RPE_QUIET="1", **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        res[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    for k in res["device"]:
        d, h = res["device"][k], res["host"][k]
        assert d["rot"] < 0.05 and d["trans"] < 0.25 and h["rot"] < 0.05 and h["trans"] < 0.25, (k, d, h)
        assert abs(d["votes"] - h["votes"]) <= 0.02 * h["votes"], (k, d, h)
        assert d["iters"] > 56 and h["iters"] > 56, (k, d, h)   # 8 + 16 + 32 host-generated iterations, then device batches


def test_p3p_batch_argument_and_state_errors(gpu_ctx_factory):
    sc = util.scene_full(3, 500, np.float32)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, bv=sc.U)
    votes = np.zeros(64, np.int32); q7 = np.zeros((64, 7)); valid = np.zeros(64, np.uint8)
    args = lambda solver, iters: (ctx._h, solver, C.c_uint64(1), C.c_uint64(109), iters, 0.2, 0.9999, 0.99, votes.ctypes.data_as(C.c_void_p),
                                  q7.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
    lib = L.lib()
    assert lib.rpe_ransac_p3p_batch(*args(5, 8)) == L.RPE_ERR_ARG            # unknown solver
    assert lib.rpe_ransac_p3p_batch(*args(0, 0)) == L.RPE_ERR_ARG            # no iterations
    assert lib.rpe_ransac_p3p_batch(*args(0, 8193)) == L.RPE_ERR_ARG         # more slots than one scoring launch takes
    assert lib.rpe_ransac_p3p_batch(*args(1, 8)) == L.RPE_ERR_STATE          # shinji + kneip needs the camera points
    assert lib.rpe_ransac_p3p_batch(*args(4, 8)) == L.RPE_ERR_STATE          # the normal-aware solvers need all five arrays
    assert lib.rpe_ransac_p3p_batch(*args(0, 8)) == L.RPE_OK
    tiny = gpu_ctx_factory().load(L.F32, xw=sc.Q[:3], bv=sc.U[:3])
    a = list(args(0, 8)); a[0] = tiny._h
    assert lib.rpe_ransac_p3p_batch(*a) == L.RPE_ERR_ARG                     # fewer than 4 correspondences
