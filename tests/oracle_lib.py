"""ctypes binding of oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

# method / ls / vote-kind ids: keep in sync with oracle/oracle_capi.cpp
M_SHINJI_RANSAC, M_SHINJI_RANSAC2, M_SHINJI_PROSAC, M_KNEIP_RANSAC, M_KNEIP_PROSAC = 0, 1, 2, 3, 4
M_SK_RANSAC, M_SK_PROSAC, M_NL_KNEIP_RANSAC, M_NL_SHINJI_RANSAC, M_NL_SK_RANSAC, M_NONE = 5, 6, 7, 8, 9, 10
LS_NONE, LS_SHINJI_INLIERS, LS_NL_BUGCOMPAT, LS_NL_FIXED, LS_SHINJI_ALL = 0, 1, 2, 3, 4
V_33, V_23, V_33_23, V_NN_23, V_NN_33, V_NN_33_23, V_23_MATRIX = 0, 1, 2, 3, 4, 5, 6
GN_P2P, GN_P2PLANE, GN_BEARING, GN_NORMAL, GN_REPROJ = 0, 1, 2, 3, 4
ROBUST_NONE, ROBUST_HUBER, ROBUST_CAUCHY = 0, 1, 2


class OrcProblem(C.Structure):
    _fields_ = [("n", C.c_int), ("bv", C.c_void_p), ("xc", C.c_void_p), ("nc", C.c_void_p), ("xw", C.c_void_p),
                ("nw", C.c_void_p), ("weights", C.c_void_p), ("wcols", C.c_int), ("fx", C.c_double), ("fy", C.c_double)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle_capi.cpp", "orc_linalg.hpp", "orc_pose.hpp", "orc_gn.hpp")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_cos_thr.restype = C.c_double
        _lib.orc_cos_nl.restype = C.c_double
        _lib.orc_time_ao.restype = C.c_double
        _lib.orc_time_gn_p2p.restype = C.c_double
        _lib.orc_time_votes33.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dt(is_f64):
    return np.float64 if is_f64 else np.float32


def _arr(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


class Problem:
    """Keeps the numpy buffers alive for the lifetime of the C struct."""

    def __init__(self, is_f64, xw=None, xc=None, bv=None, nw=None, nc=None, weights=None, f=585.0):
        dt = _dt(is_f64)
        self.is_f64 = int(bool(is_f64))
        self.xw, self.xc, self.bv, self.nw, self.nc = (_arr(a, dt) for a in (xw, xc, bv, nw, nc))
        self.w = None if weights is None else np.asfortranarray(weights, dtype=dt)
        n = next(len(a) for a in (self.xw, self.xc, self.bv) if a is not None)
        self.n = n
        self.c = OrcProblem(n, _p(self.bv), _p(self.xc), _p(self.nc), _p(self.xw), _p(self.nw), _p(self.w),
                            0 if self.w is None else self.w.shape[1], f, f)


def shinji(xw, xc, K=None, is_f64=True):
    dt = _dt(is_f64)
    xw, xc = _arr(xw, dt), _arr(xc, dt)
    n = len(xw)
    R, t = np.zeros(9), np.zeros(3)
    rc = lib().orc_shinji(int(is_f64), _p(xw), _p(xc), n, n if K is None else K, _p(R), _p(t))
    return R.reshape(3, 3), t, rc


def shinji_f32in_f64(xw, xc):
    xw, xc = _arr(xw, np.float32), _arr(xc, np.float32)
    R, t = np.zeros(9), np.zeros(3)
    rc = lib().orc_shinji_f32in_f64(_p(xw), _p(xc), len(xw), _p(R), _p(t))
    return R.reshape(3, 3), t, rc


def ao(xw, xc):
    xw, xc = _arr(xw, np.float32), _arr(xc, np.float32)
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    lib().orc_ao(_p(xw), _p(xc), len(xw), _p(R), _p(t))
    return R.reshape(3, 3), t


def ao_ransac(xw, xc, seed=1):
    xw, xc = _arr(xw, np.float32), _arr(xc, np.float32)
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    it, votes = C.c_int(0), C.c_int(0)
    lib().orc_ao_ransac(_p(xw), _p(xc), len(xw), _p(R), _p(t), C.c_uint64(seed), C.byref(it), C.byref(votes))
    return R.reshape(3, 3), t, it.value, votes.value


def run(prob: Problem, method, thre_3d=0.0, thre_2d=0.0, thre_nl=0.0, iters=0, confidence=0.99, seed=1, ls=LS_NONE,
        adapter_kind_for_none=None, mask_in=None, pose_in=None, max_votes_in=1):
    if adapter_kind_for_none is None:  # the adapter the reference's demos would build for these arrays
        adapter_kind_for_none = 2 if prob.nc is not None else (1 if prob.bv is not None else 0)
    R, t = np.zeros(9), np.zeros(3)
    if pose_in is not None:
        R[:] = np.asarray(pose_in[0], float).reshape(9)
        t[:] = np.asarray(pose_in[1], float)
    it = C.c_int(iters)
    mv = C.c_int(max_votes_in)
    mask_out = np.zeros((3, prob.n), np.int16)
    mi = None if mask_in is None else np.ascontiguousarray(mask_in, dtype=np.int16)
    lib().orc_run(prob.is_f64, method, C.byref(prob.c), C.c_double(thre_3d), C.c_double(thre_2d), C.c_double(thre_nl), C.byref(it),
                  C.c_double(confidence), C.c_uint64(seed), ls, adapter_kind_for_none, _p(mi), _p(R), _p(t), C.byref(mv), _p(mask_out))
    return dict(R=R.reshape(3, 3), t=t, iters=it.value, max_votes=mv.value, masks=mask_out)


def hypotheses(prob: Problem, method, iters, seed=1):
    """orc_hypotheses: the hypothesis stream of `method` over `iters` iterations -> (q7[H, 7], first[iters + 1])."""
    cap = 3 * iters + 1
    q7, first = np.zeros((cap, 7)), np.zeros(iters + 1, np.int32)
    H = lib().orc_hypotheses(prob.is_f64, method, C.byref(prob.c), iters, C.c_uint64(seed), _p(q7), cap, _p(first))
    assert H >= 0
    return q7[:H].copy(), first


def run_replay(prob: Problem, method, poses7, first, thre_3d=0.0, thre_2d=0.0, thre_nl=0.0, iters=0, confidence=0.99, ls=LS_NONE):
    poses7 = np.ascontiguousarray(poses7, np.float64).reshape(-1, 7)
    first = np.ascontiguousarray(first, np.int32)
    R, t = np.zeros(9), np.zeros(3)
    it, mv = C.c_int(iters), C.c_int(0)
    mask_out = np.zeros((3, prob.n), np.int16)
    lib().orc_run_replay(prob.is_f64, method, C.byref(prob.c), _p(poses7), _p(first), len(first) - 1, C.c_double(thre_3d), C.c_double(thre_2d),
                         C.c_double(thre_nl), C.byref(it), C.c_double(confidence), ls, _p(R), _p(t), C.byref(mv), _p(mask_out))
    return dict(R=R.reshape(3, 3), t=t, iters=it.value, max_votes=mv.value, masks=mask_out)


def votes(prob: Problem, kind, poses7, thre_3d=0.0, cos_thr=2.0, cos_nl=2.0, mask_for=-1):
    poses7 = np.ascontiguousarray(poses7, dtype=np.float64).reshape(-1, 7)
    H = len(poses7)
    v = np.zeros(H, np.int32)
    mask = np.zeros((3, prob.n), np.int16) if mask_for >= 0 else None
    lib().orc_votes(prob.is_f64, kind, C.byref(prob.c), _p(poses7), H, C.c_double(thre_3d), C.c_double(cos_thr), C.c_double(cos_nl),
                    _p(v), _p(mask), mask_for)
    return (v, mask) if mask_for >= 0 else v


def cos_thr(is_f64, thre_2d, f):
    return lib().orc_cos_thr(int(is_f64), C.c_double(thre_2d), C.c_double(f), C.c_double(f))


def cos_nl(is_f64, nl_thre):
    return lib().orc_cos_nl(int(is_f64), C.c_double(nl_thre))


def residual_33(xw, xc, pose7, is_f64):
    dt = _dt(is_f64)
    xw, xc = _arr(xw, dt), _arr(xc, dt)
    q = np.ascontiguousarray(pose7, np.float64)
    out = np.zeros(len(xw))
    lib().orc_residual_33(int(is_f64), _p(xw), _p(xc), len(xw), _p(q), _p(out))
    return out


def cos_23(xw, bv, pose7, is_f64):
    dt = _dt(is_f64)
    xw, bv = _arr(xw, dt), _arr(bv, dt)
    q = np.ascontiguousarray(pose7, np.float64)
    out = np.zeros(len(xw))
    lib().orc_cos_23(int(is_f64), _p(xw), _p(bv), len(xw), _p(q), _p(out))
    return out


def lsq_pnp(xw, bv, pose7, is_f64, with_terms=False):
    """(the reference's sequential Tp total, the same Tp terms added in double[, the terms])"""
    dt = _dt(is_f64)
    xw, bv = _arr(xw, dt), _arr(bv, dt)
    q = np.ascontiguousarray(pose7, np.float64)
    out, terms = np.zeros(2), np.zeros(len(xw))
    lib().orc_lsq_pnp(int(is_f64), _p(xw), _p(bv), len(xw), _p(q), _p(out), _p(terms))
    return (out[0], out[1], terms) if with_terms else (out[0], out[1])


def cos_nn(nw, nc, pose7, is_f64):
    dt = _dt(is_f64)
    nw, nc = _arr(nw, dt), _arr(nc, dt)
    q = np.ascontiguousarray(pose7, np.float64)
    out = np.zeros(len(nw))
    lib().orc_cos_nn(int(is_f64), _p(nw), _p(nc), len(nw), _p(q), _p(out))
    return out


def ransac_update_num_iters(is_f64, p, ep, model_points, max_iters):
    return lib().orc_ransac_update_num_iters(int(is_f64), C.c_double(p), C.c_double(ep), model_points, max_iters)


def rand31(seed, count):
    out = np.zeros(count, np.int32)
    lib().orc_rand31(C.c_uint64(seed), count, _p(out))
    return out


def random_elements(n, m, seed, draws):
    out = np.zeros((draws, m), np.int32)
    lib().orc_random_elements(n, m, C.c_uint64(seed), draws, _p(out))
    return out


def prosac_samples(is_f64, m, n, seed, draws):
    out = np.zeros((draws, m), np.int32)
    lib().orc_prosac_samples(int(is_f64), m, n, C.c_uint64(seed), draws, _p(out))
    return out


def sort_indexes(w, is_f64=True):
    w = _arr(w, _dt(is_f64))
    out = np.zeros(len(w), np.int32)
    lib().orc_sort_indexes(int(is_f64), _p(w), len(w), _p(out))
    return out


def kneip_main(xw4, bv4, is_f64=True):
    dt = _dt(is_f64)
    xw4, bv4 = _arr(xw4, dt), _arr(bv4, dt)
    sols = np.zeros((4, 12))
    cnt = lib().orc_kneip_main(int(is_f64), _p(xw4), _p(bv4), _p(sols))
    return [(sols[i, :9].reshape(3, 3).copy(), sols[i, 9:].copy()) for i in range(cnt)]


def kneip(xw4, bv4, is_f64=True):
    dt = _dt(is_f64)
    xw4, bv4 = _arr(xw4, dt), _arr(bv4, dt)
    R, t = np.zeros(9), np.zeros(3)
    ok = lib().orc_kneip(int(is_f64), _p(xw4), _p(bv4), _p(R), _p(t))
    return (R.reshape(3, 3), t) if ok else None


def o4_roots(p5):
    p5 = np.ascontiguousarray(p5, np.float64)
    r = np.zeros(4)
    lib().orc_o4_roots(_p(p5), _p(r))
    return r


def nl_2p(pt1_c, nl1_c, pt2_c, pt1_w, nl1_w, pt2_w, is_f64=True):
    v = np.ascontiguousarray(np.stack([pt1_c, nl1_c, pt2_c, pt1_w, nl1_w, pt2_w]), np.float64)
    R, t = np.zeros(9), np.zeros(3)
    lib().orc_nl_2p(int(is_f64), _p(v), _p(R), _p(t))
    return R.reshape(3, 3), t


def find_opt_cc(prob: Problem, R, mask23):
    R = np.ascontiguousarray(R, np.float64)
    m = np.ascontiguousarray(mask23, np.int16)
    c = np.zeros(3)
    lib().orc_find_opt_cc(prob.is_f64, C.byref(prob.c), _p(R), _p(m), _p(c))
    return c


def calc_err(Rgt, tgt, Rse, tse):
    a = [np.ascontiguousarray(x, np.float64) for x in (Rgt, tgt, Rse, tse)]
    out = np.zeros(2)
    lib().orc_calc_err(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out))
    return out  # (t_e, r_e)


def calc_percentage_err(Rgt, tgt, Rse, tse):
    a = [np.ascontiguousarray(x, np.float64) for x in (Rgt, tgt, Rse, tse)]
    out = np.zeros(2)
    lib().orc_calc_percentage_err(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out))
    return out


def se3_exp(a6):
    a6 = np.ascontiguousarray(a6, np.float64)
    R, t = np.zeros(9), np.zeros(3)
    lib().orc_se3_exp(_p(a6), _p(R), _p(t))
    return R.reshape(3, 3), t


def se3_log(R, t):
    R, t = np.ascontiguousarray(R, np.float64), np.ascontiguousarray(t, np.float64)
    a = np.zeros(6)
    lib().orc_se3_log(_p(R), _p(t), _p(a))
    return a


def svd3(A):
    A = np.ascontiguousarray(A, np.float64)
    U, s, V = np.zeros(9), np.zeros(3), np.zeros(9)
    lib().orc_svd3(_p(A), _p(U), _p(s), _p(V))
    return U.reshape(3, 3), s, V.reshape(3, 3)


def quat_from_R(R, is_f64=True):
    R = np.ascontiguousarray(R, np.float64)
    q = np.zeros(4)
    lib().orc_quat_from_R(int(is_f64), _p(R), _p(q))
    return q


def pose12(R, t):
    return np.concatenate([np.asarray(R, float).reshape(9), np.asarray(t, float).reshape(3)])


def pose7_from_Rt(R, t, is_f64):
    """(qw qx qy qz tx ty tz) in Tp precision, the way a Sophus::SE3<Tp> built from R would hold it."""
    dt = _dt(is_f64)
    Rr = np.asarray(R, dt).astype(np.float64)
    q = quat_from_R(Rr, is_f64)
    return np.concatenate([q, np.asarray(t, dt).astype(np.float64)])


def gn_normal_eq(kind, a, b, c=None, mask=None, weight=None, pose=None, in_f64=False, robust=0, robust_k=1.0):
    dt = _dt(in_f64)
    a, b, c, weight = _arr(a, dt), _arr(b, dt), _arr(c, dt), _arr(weight, dt)
    mask = None if mask is None else np.ascontiguousarray(mask, np.int16)
    p = np.ascontiguousarray(pose, np.float64)
    out = np.zeros(29)
    lib().orc_gn_normal_eq_robust(int(in_f64), kind, _p(a), _p(b), _p(c), _p(mask), _p(weight), C.c_long(len(a)), _p(p), robust,
                                  C.c_double(robust_k), _p(out))
    return out


def gn_solve(packed29):
    p = np.ascontiguousarray(packed29, np.float64)
    d = np.zeros(6)
    rc = lib().orc_gn_solve(_p(p), _p(d))
    return d, rc


def gn_apply(delta6, pose):
    d = np.ascontiguousarray(delta6, np.float64)
    p = np.array(pose, np.float64).copy()
    lib().orc_gn_apply(_p(d), _p(p))
    return p


def gn_refine(terms, n, pose, max_iter=20, tol=1e-9, in_f64=False):
    """terms: list of dicts kind,a,b[,c,mask,weight,scale]."""
    dt = _dt(in_f64)
    k = len(terms)
    keep = []
    def vp(key, conv):
        arr = (C.c_void_p * k)()
        for i, tm in enumerate(terms):
            v = tm.get(key)
            if v is not None:
                v = conv(v); keep.append(v); arr[i] = v.ctypes.data
            else:
                arr[i] = None
        return arr
    kinds = (C.c_int * k)(*[tm["kind"] for tm in terms])
    as_, bs, cs = vp("a", lambda v: _arr(v, dt)), vp("b", lambda v: _arr(v, dt)), vp("c", lambda v: _arr(v, dt))
    masks = vp("mask", lambda v: np.ascontiguousarray(v, np.int16))
    ws = vp("weight", lambda v: _arr(v, dt))
    scales = (C.c_double * k)(*[float(tm.get("scale", 1.0)) for tm in terms])
    robusts = (C.c_int * k)(*[int(tm.get("robust", 0)) for tm in terms])
    rks = (C.c_double * k)(*[float(tm.get("robust_k", 1.0)) for tm in terms])
    p = np.array(pose, np.float64).copy()
    step, cost = C.c_double(0), C.c_double(0)
    its = lib().orc_gn_refine(int(in_f64), k, kinds, as_, bs, cs, masks, ws, scales, robusts, rks, C.c_long(n), _p(p), max_iter,
                              C.c_double(tol), C.byref(step), C.byref(cost))
    return p, its, step.value, cost.value
