"""Thin numpy-facing wrapper of the C ABI (include/rgbd_pose_hip.h) used by tests, bench.py and the
multi-GPU driver.  All computation happens in librgbdpose_hip.so on the GPU; nothing here computes."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _np_dtype(dtype):
    return np.float64 if dtype == L.F64 else np.float32


def pose12(R, t):
    return np.concatenate([np.asarray(R, np.float64).reshape(9), np.asarray(t, np.float64).reshape(3)])


class Context:
    """One GPU context holding a correspondence set resident in HBM (rpe_context)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        L.check(L.lib().rpe_create(C.byref(self._h), device, C.c_void_p(stream) if stream else None))
        self.n = 0
        self.dtype = L.F32
        self._keep = {}

    def close(self):
        if self._h:
            L.lib().rpe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- data
    def set_problem(self, n: int, dtype=L.F32):
        L.check(L.lib().rpe_set_problem(self._h, n, dtype))
        self.n, self.dtype = n, dtype

    def upload(self, slot: int, host: np.ndarray):
        a = np.ascontiguousarray(host, dtype=_np_dtype(self.dtype))
        assert a.size == 3 * self.n, (a.shape, self.n)
        L.check(L.lib().rpe_upload(self._h, slot, _p(a)))
        L.check(L.lib().rpe_synchronize(self._h))

    def bind(self, slot: int, device_ptr: int):
        L.check(L.lib().rpe_bind(self._h, slot, C.c_void_p(device_ptr)))

    def load(self, dtype=L.F32, xw=None, xc=None, bv=None, nw=None, nc=None):
        n = next(len(a) for a in (xw, xc, bv) if a is not None)
        self.set_problem(n, dtype)
        for slot, a in ((L.XW, xw), (L.XC, xc), (L.BV, bv), (L.NW, nw), (L.NC, nc)):
            if a is not None:
                self.upload(slot, a)
        return self

    def upload_mask(self, modality: int, mask):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.int16)
        L.check(L.lib().rpe_upload_mask(self._h, modality, _p(m)))

    def upload_weight(self, modality: int, w):
        a = None if w is None else np.ascontiguousarray(w, dtype=_np_dtype(self.dtype))
        L.check(L.lib().rpe_upload_weight(self._h, modality, _p(a)))

    def download_mask(self, modality: int) -> np.ndarray:
        m = np.zeros(self.n, np.int16)
        L.check(L.lib().rpe_download_mask(self._h, modality, _p(m)))
        return m

    def synchronize(self):
        L.check(L.lib().rpe_synchronize(self._h))

    # ---- kernels
    def p2p_moments(self, flags: int = 0) -> np.ndarray:
        out = np.zeros(17)
        L.check(L.lib().rpe_p2p_moments(self._h, flags, _p(out)))
        return out

    def normal_eq(self, kind: int, pose, flags: int = 0):
        """Returns (record32, pose_used12)."""
        p = np.array(pose, np.float64).reshape(12).copy()
        out = np.zeros(32)
        L.check(L.lib().rpe_normal_eq(self._h, kind, flags, _p(p), _p(out)))
        return out, p

    def normal_eq_device(self, kind: int, pose, d_out_ptr: int, flags: int = 0):
        p = np.array(pose, np.float64).reshape(12).copy()
        L.check(L.lib().rpe_normal_eq_device(self._h, kind, flags, _p(p), C.c_void_p(d_out_ptr)))
        return p

    def gn_refine(self, kinds, pose, scales=None, flags: int = 0, max_iter: int = 20, tol: float = 1e-9):
        kinds = np.ascontiguousarray(kinds, np.int32)
        sc = None if scales is None else np.ascontiguousarray(scales, np.float64)
        p = np.array(pose, np.float64).reshape(12).copy()
        it, step, cost = C.c_int(0), C.c_double(0), C.c_double(0)
        L.check(L.lib().rpe_gn_refine(self._h, len(kinds), _p(kinds), _p(sc), flags, _p(p), max_iter, tol, C.byref(it), C.byref(step),
                                      C.byref(cost)))
        return p, it.value, step.value, cost.value

    def score(self, kind: int, poses7, thre_3d=0.0, cos_thr=2.0, cos_nl=2.0, mode=L.SCORE_FAST) -> np.ndarray:
        q = np.ascontiguousarray(poses7, np.float64).reshape(-1, 7)
        v = np.zeros(len(q), np.int32)
        L.check(L.lib().rpe_score(self._h, kind, mode, _p(q), len(q), thre_3d, cos_thr, cos_nl, _p(v)))
        return v

    def inlier_mask(self, kind: int, pose7, thre_3d=0.0, cos_thr=2.0, cos_nl=2.0, mode=L.SCORE_FAST) -> int:
        q = np.ascontiguousarray(pose7, np.float64).reshape(7)
        v = C.c_int(0)
        L.check(L.lib().rpe_inlier_mask(self._h, kind, mode, _p(q), thre_3d, cos_thr, cos_nl, C.byref(v)))
        return v.value

    def nl_round(self, c_opt, Cw, Cc, Rwc) -> np.ndarray:
        a = [np.ascontiguousarray(x, np.float64) for x in (c_opt, Cw, Cc, Rwc)]
        out = np.zeros(44)
        L.check(L.lib().rpe_nl_round(self._h, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out)))
        return out


def pose_from_moments(m17):
    m = np.ascontiguousarray(m17, np.float64)
    R, t = np.zeros(9), np.zeros(3)
    L.check(L.lib().rpe_pose_from_moments(_p(m), _p(R), _p(t)))
    return R.reshape(3, 3), t


def gn_solve(ne32):
    a = np.ascontiguousarray(ne32, np.float64)
    d = np.zeros(6)
    L.check(L.lib().rpe_gn_solve(_p(a), _p(d)))
    return d


def gn_apply(delta6, pose):
    d = np.ascontiguousarray(delta6, np.float64)
    p = np.array(pose, np.float64).reshape(12).copy()
    L.check(L.lib().rpe_gn_apply(_p(d), _p(p)))
    return p


def ao(xw, xc):
    """Reference FFI: ao(x_w, x_c, n, R_cw, t) (Library.cpp:17)."""
    xw = np.ascontiguousarray(xw, np.float32)
    xc = np.ascontiguousarray(xc, np.float32)
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    L.lib().ao(_p(xw), _p(xc), len(xw), _p(R), _p(t))
    return R.reshape(3, 3), t


def ao_ransac(xw, xc):
    """Reference FFI: ao_ransac(x_w, x_c, n, R_cw, t) (Library.cpp:47)."""
    xw = np.ascontiguousarray(xw, np.float32)
    xc = np.ascontiguousarray(xc, np.float32)
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    L.lib().ao_ransac(_p(xw), _p(xc), len(xw), _p(R), _p(t))
    return R.reshape(3, 3), t
