"""Thin numpy-facing wrapper of the C ABI (include/rgbd_pose_hip.h) used by tests, bench.py and the
multi-GPU driver.  All computation happens in librgbdpose_hip.so on the GPU; nothing here computes."""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np

from . import _lib as L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _np_dtype(dtype):
    return np.float64 if dtype == L.F64 else np.float32


def pose12(R, t):
    return np.concatenate([np.asarray(R, np.float64).reshape(9), np.asarray(t, np.float64).reshape(3)])


def pose7_from_Rt(R, t, dtype=L.F32) -> np.ndarray:
    """(qw qx qy qz tx ty tz) of the Sophus::SE3<Tp> a caller would build from R, t: the quaternion is extracted
    from the Tp-rounded matrix in Tp arithmetic (Eigen's Quaternion(Matrix3) branches, sophus/so3.hpp:561).
    Host-side input preparation only."""
    dt = _np_dtype(dtype)
    m = np.asarray(R, dt).reshape(3, 3)
    one, half = dt(1), dt(0.5)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4, dt)
    if tr > 0:
        s = np.sqrt(tr + one); q[0] = half * s; s = half / s
        q[1] = (m[2, 1] - m[1, 2]) * s; q[2] = (m[0, 2] - m[2, 0]) * s; q[3] = (m[1, 0] - m[0, 1]) * s
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + one)
        q[1 + i] = half * s; s = half / s
        q[0] = (m[k, j] - m[j, k]) * s; q[1 + j] = (m[j, i] + m[i, j]) * s; q[1 + k] = (m[k, i] + m[i, k]) * s
    return np.concatenate([q.astype(np.float64), np.asarray(t, dt).astype(np.float64)])


class Context:
    """One GPU context holding a correspondence set resident in HBM (rpe_context)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        L.check(L.lib().rpe_create(C.byref(self._h), device, C.c_void_p(stream) if stream else None))
        self.n = 0
        self.dtype = L.F32
        self._keep = {}
        self._pixels = self._model_pixels = 0   # front end: pixels of the current frame / of the model view

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

    def download(self, slot: int) -> np.ndarray:
        out = np.empty((self.n, 3), _np_dtype(self.dtype))
        L.check(L.lib().rpe_download(self._h, slot, _p(out)))
        return out

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
        out = np.zeros(18)
        L.check(L.lib().rpe_p2p_moments(self._h, flags, _p(out)))
        return out

    def sine_error_sum(self, pose7):
        """lsq_pnp (reference P3P.hpp:472-502): sum over all correspondences of |normalize(R Xw + t) x bv| at pose7 = (qw, qx, qy, qz, t);
        returns (sum, number of terms)."""
        q = np.ascontiguousarray(pose7, np.float64).reshape(7)
        out, cnt = np.zeros(1), np.zeros(1, np.int64)
        L.check(L.lib().rpe_sine_error_sum(self._h, _p(q), _p(out), _p(cnt)))
        return float(out[0]), int(cnt[0])

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

    @staticmethod
    def _terms(terms):
        """terms: list of (kind, scale[, robust, robust_k]) tuples or dicts."""
        arr = (L.RpeTerm * len(terms))()
        for i, t in enumerate(terms):
            if isinstance(t, dict):
                t = (t["kind"], t.get("scale", 1.0), t.get("robust", 0), t.get("robust_k", 1.0))
            t = tuple(t) + (1.0, 0, 1.0)[len(t) - 1:]
            arr[i].kind, arr[i].scale, arr[i].robust, arr[i].robust_k = int(t[0]), float(t[1]), int(t[2]), float(t[3])
        return arr

    def normal_eq_joint(self, terms, pose, flags: int = 0) -> np.ndarray:
        arr = self._terms(terms)
        p = np.array(pose, np.float64).reshape(12).copy()
        out = np.zeros(32)
        L.check(L.lib().rpe_normal_eq_joint(self._h, len(arr), arr, flags, _p(p), _p(out)))
        return out

    def gn_refine_joint(self, terms, pose, flags: int = 0, max_iter: int = 20, tol: float = 1e-9):
        arr = self._terms(terms)
        p = np.array(pose, np.float64).reshape(12).copy()
        it, step, cost = C.c_int(0), C.c_double(0), C.c_double(0)
        L.check(L.lib().rpe_gn_refine_joint(self._h, len(arr), arr, flags, _p(p), max_iter, tol, C.byref(it), C.byref(step), C.byref(cost)))
        return p, it.value, step.value, cost.value

    def gn_refine_device(self, terms, pose, flags: int = 0, max_iter: int = 20, tol: float = 1e-9):
        """Device-resident loop: one launch per iteration, solve + exp-map on the GPU, one host wait at the end."""
        arr = self._terms(terms)
        p = np.array(pose, np.float64).reshape(12).copy()
        it, step, cost = C.c_int(0), C.c_double(0), C.c_double(0)
        L.check(L.lib().rpe_gn_refine_device(self._h, len(arr), arr, flags, _p(p), max_iter, tol, C.byref(it), C.byref(step), C.byref(cost)))
        return p, it.value, step.value, cost.value

    def gn_step(self, kind: int, pose12_inout: np.ndarray, flags: int = 0) -> float:
        """One GN step in place on a float64[12] array; returns |delta|."""
        step = C.c_double(0)
        L.check(L.lib().rpe_gn_step(self._h, kind, flags, _p(pose12_inout), None, C.byref(step)))
        return step.value

    def gn_step_dist(self, kind: int, pose12_inout: np.ndarray, flags: int = 0) -> float:
        """One sharded GN step in place (kernel -> RCCL all-reduce -> host solve); needs comm_init."""
        step = C.c_double(0)
        L.check(L.lib().rpe_gn_step_dist(self._h, kind, flags, _p(pose12_inout), None, C.byref(step)))
        return step.value

    # ---- front end (Part 3 of the C ABI): depth frame -> maps -> projective association -> ICP
    @staticmethod
    def _camera(cam) -> "L.RpeCamera":
        """cam: (fx, fy, cx, cy, width, height); defaults of the reference simulator: (585, 585, 320, 240, 640, 480)."""
        fx, fy, cx, cy, w, h = cam
        return L.RpeCamera(float(fx), float(fy), float(cx), float(cy), int(w), int(h))

    def frame_set_depth(self, depth: np.ndarray, cam=(585.0, 585.0, 320.0, 240.0, 640, 480), depth_scale: float | None = None,
                        dmin: float = 0.0, dmax: float = 1e30, max_jump: float = 0.1):
        """depth: (height, width) uint16 (default scale 0.001: millimetres) or float32 (default scale 1: metres)."""
        k = self._camera(cam)
        d = np.ascontiguousarray(depth)
        if d.shape != (k.height, k.width):
            raise ValueError(f"depth shape {d.shape} does not match the camera ({k.height}, {k.width})")
        if d.dtype == np.uint16:
            kind, scale = L.DEPTH_U16, 0.001 if depth_scale is None else depth_scale
        elif d.dtype == np.float32:
            kind, scale = L.DEPTH_F32, 1.0 if depth_scale is None else depth_scale
        else:
            raise TypeError("depth must be uint16 or float32")
        L.check(L.lib().rpe_frame_set_depth(self._h, _p(d), kind, C.byref(k), scale, dmin, dmax, max_jump))
        self._pixels = k.width * k.height
        return self

    def frame_download(self, which: int) -> np.ndarray:
        """One map as (pixels, 3) float32; pixels of the model's view for the MAP_MODEL_* maps."""
        n = self._model_pixels if which >= L.MAP_MODEL_VERTEX else self._pixels
        out = np.empty((n, 3), np.float32)
        L.check(L.lib().rpe_frame_download(self._h, which, _p(out)))
        return out

    def model_from_frame(self, pose12):
        p = np.array(pose12, np.float64).reshape(12)
        L.check(L.lib().rpe_model_from_frame(self._h, _p(p)))
        self._model_pixels = self._pixels
        return self

    def model_upload(self, vertex_w: np.ndarray, normal_w: np.ndarray, cam, pose12):
        k = self._camera(cam)
        v = np.ascontiguousarray(vertex_w, np.float32).reshape(-1, 3)
        nw = np.ascontiguousarray(normal_w, np.float32).reshape(-1, 3)
        if len(v) != k.width * k.height or len(nw) != len(v):
            raise ValueError("model maps must hold width*height xyz triples")
        p = np.array(pose12, np.float64).reshape(12)
        L.check(L.lib().rpe_model_upload(self._h, _p(v), _p(nw), C.byref(k), _p(p)))
        self._model_pixels = len(v)
        return self

    def associate(self, pose12, dist_thr: float = 0.1, cos_thr: float = 0.9, use_normals: bool = True, count: bool = True):
        """Fill XW XC BV NW NC of this context from the frame and the model; returns the number of pairs (or None)."""
        p = np.array(pose12, np.float64).reshape(12)
        m = C.c_int64(0)
        L.check(L.lib().rpe_associate(self._h, _p(p), dist_thr, cos_thr, int(use_normals), C.byref(m) if count else None))
        self.n, self.dtype = self._pixels, L.F32
        return m.value if count else None

    def icp(self, pose12, kind: int = L.RES_P2PLANE, max_iter: int = 10, tol: float = 1e-6, dist_thr: float = 0.1, cos_thr: float = 0.9,
            use_normals: bool = True, device_resident: bool = False, fused: bool = False):
        """Projective-association ICP; returns (pose12, iterations, last |delta|, cost, pairs of the last round)."""
        p = np.array(pose12, np.float64).reshape(12).copy()
        o = L.RpeIcpOptions(kind, max_iter, tol, dist_thr, cos_thr, int(use_normals), int(device_resident), int(fused))
        it, step, cost, m = C.c_int(0), C.c_double(0), C.c_double(0), C.c_int64(0)
        L.check(L.lib().rpe_icp(self._h, C.byref(o), _p(p), C.byref(it), C.byref(step), C.byref(cost), C.byref(m)))
        self.n, self.dtype = self._pixels, L.F32
        return p, it.value, step.value, cost.value, m.value


    def gn_steps_dist(self, kind: int, pose12_inout: np.ndarray, steps: int, flags: int = 0) -> float:
        """`steps` sharded GN steps in place, the loop inside the library; returns the last |delta|."""
        step = C.c_double(0)
        L.check(L.lib().rpe_gn_steps_dist(self._h, kind, flags, _p(pose12_inout), steps, C.byref(step)))
        return step.value

    def gn_steps_dist_device(self, kind: int, pose12_inout: np.ndarray, steps: int, flags: int = 0) -> float:
        """`steps` sharded GN steps over the RCCL communicator chained on the device (rpe_gn_steps_dist_device: solve + exp-map in the
        kernels, the host enqueues everything and waits once); returns the last |delta|."""
        step = C.c_double(0)
        L.check(L.lib().rpe_gn_steps_dist_device(self._h, kind, flags, _p(pose12_inout), steps, C.byref(step)))
        return step.value

    def comm_init(self, world: int, rank: int, id128: bytes):
        buf = (C.c_char * 128).from_buffer_copy(id128)
        L.check(L.lib().rpe_comm_init(self._h, world, rank, buf))

    def p2p_export(self) -> bytes:
        """64-byte HIP IPC handle of this context's mailbox (peer-to-peer all-reduce over xGMI)."""
        buf = (C.c_char * 64)()
        L.check(L.lib().rpe_p2p_export(self._h, buf))
        return bytes(buf)

    def p2p_init(self, world: int, rank: int, handles: bytes):
        assert len(handles) == 64 * world
        buf = (C.c_char * len(handles)).from_buffer_copy(handles)
        L.check(L.lib().rpe_p2p_init(self._h, world, rank, buf))

    def p2p_destroy(self):
        L.check(L.lib().rpe_p2p_destroy(self._h))

    def comm_count(self) -> int:
        """ranks of this context's RCCL communicator, from the communicator (ncclCommCount); 0 = none"""
        n = C.c_int(0)
        L.check(L.lib().rpe_comm_count(self._h, C.byref(n)))
        return n.value

    def bus_id(self) -> str:
        buf = C.create_string_buffer(64)
        L.check(L.lib().rpe_device_bus_id(self._h, buf, 64))
        return buf.value.decode()

    def comm_destroy(self):
        L.check(L.lib().rpe_comm_destroy(self._h))

    def hostex_init(self, world: int, rank: int, name: str, create: bool):
        """Host-side all-reduce between the node's rank processes (POSIX shared memory, sums in rank order); after this
        gn_step_dist / gn_steps_dist / gn_refine / score are sharded calls."""
        L.check(L.lib().rpe_hostex_init(self._h, world, rank, name.encode(), 1 if create else 0))

    def hostex_destroy(self):
        L.check(L.lib().rpe_hostex_destroy(self._h))

    def tune_host_thread(self, kind: int, pose, flags: int = 0, steps: int = 200, reps: int = 5):
        """rpe_tune_host_thread: measure candidate CPUs for the thread that drives the resident loop, leave THIS thread pinned to the
        fastest.  Returns dict(cpu, us_per_step, trials={cpu: us})."""
        p = np.ascontiguousarray(pose, np.float64).reshape(12)
        best, us, n = C.c_int(-1), C.c_double(0), C.c_int(0)
        cpus, tus = np.zeros(16, np.int32), np.zeros(16, np.float64)
        L.check(L.lib().rpe_tune_host_thread(self._h, kind, flags, _p(p), steps, reps, C.byref(best), C.byref(us), _p(cpus), _p(tus), 16, C.byref(n)))
        return {"cpu": best.value, "us_per_step": us.value, "trials": {int(cpus[i]): float(tus[i]) for i in range(n.value)}}

    def timing_enable(self, max_records: int, stride: int = 1):
        L.check(L.lib().rpe_timing_enable(self._h, max_records, stride))

    def timing_calibrate(self, pairs: int = 200):
        """(average, minimum) milliseconds an empty HIP event pair reports on this context's stream."""
        a, m = C.c_double(0), C.c_double(0)
        L.check(L.lib().rpe_timing_calibrate(self._h, pairs, C.byref(a), C.byref(m)))
        return a.value, m.value

    def timing_collect(self):
        cnt, tot, mn = C.c_int(0), C.c_double(0), C.c_double(0)
        L.check(L.lib().rpe_timing_collect(self._h, C.byref(cnt), C.byref(tot), C.byref(mn)))
        return cnt.value, tot.value, mn.value

    def resident_state(self) -> dict:
        """enabled / lost grids / co-residency cap of this context's resident loops (rpe_debug_resident_state)."""
        en, lost, cap = C.c_int(0), C.c_int(0), C.c_int(0)
        L.check(L.lib().rpe_debug_resident_state(self._h, C.byref(en), C.byref(lost), C.byref(cap)))
        return {"enabled": bool(en.value), "host_driven": bool(en.value & 2), "solver": bool(en.value & 4), "lost": lost.value, "cap": cap.value}

    def inject_resident_fault(self, iteration: int = 0, pose_wait_s: float = 0.0):
        """Test hook (rpe_debug_inject_resident_fault): the last workgroup of the next host-driven resident loops withholds its sums of
        `iteration`; (0, 0) switches it off."""
        L.check(L.lib().rpe_debug_inject_resident_fault(self._h, iteration, pose_wait_s))

    def gn_refine(self, kinds, pose, scales=None, flags: int = 0, max_iter: int = 20, tol: float = 1e-9):
        kinds = np.ascontiguousarray(kinds, np.int32)
        sc = None if scales is None else np.ascontiguousarray(scales, np.float64)
        p = np.array(pose, np.float64).reshape(12).copy()
        it, step, cost = C.c_int(0), C.c_double(0), C.c_double(0)
        L.check(L.lib().rpe_gn_refine(self._h, len(kinds), _p(kinds), _p(sc), flags, _p(p), max_iter, tol, C.byref(it), C.byref(step),
                                      C.byref(cost)))
        return p, it.value, step.value, cost.value

    def score(self, kind: int, poses7, thre_3d=0.0, cos_thr=2.0, cos_nl=2.0, mode=L.SCORE_EXACT) -> np.ndarray:
        q = np.ascontiguousarray(poses7, np.float64).reshape(-1, 7)
        v = np.zeros(len(q), np.int32)
        L.check(L.lib().rpe_score(self._h, kind, mode, _p(q), len(q), thre_3d, cos_thr, cos_nl, _p(v)))
        return v

    def inlier_mask(self, kind: int, pose7, thre_3d=0.0, cos_thr=2.0, cos_nl=2.0, mode=L.SCORE_EXACT) -> int:
        q = np.ascontiguousarray(pose7, np.float64).reshape(7)
        v = C.c_int(0)
        L.check(L.lib().rpe_inlier_mask(self._h, kind, mode, _p(q), thre_3d, cos_thr, cos_nl, C.byref(v)))
        return v.value

    def score_session_begin(self, kind: int, thre_3d=0.0, cos_thr=2.0, cos_nl=2.0, mode=L.SCORE_EXACT) -> bool:
        """Open a resident scoring session (rpe_score_session_begin): score() calls of <= 128 hypotheses and inlier_mask() calls with
        these parameters are then served by one resident launch.  False if the context cannot hold one (RPE_ERR_STATE)."""
        rc = L.lib().rpe_score_session_begin(self._h, kind, mode, thre_3d, cos_thr, cos_nl)
        if rc == L.RPE_ERR_STATE:
            return False
        L.check(rc)
        return True

    def score_session_end(self):
        L.check(L.lib().rpe_score_session_end(self._h))

    @contextlib.contextmanager
    def score_session(self, kind: int, thre_3d=0.0, cos_thr=2.0, cos_nl=2.0, mode=L.SCORE_EXACT):
        """`with ctx.score_session(kind, ...) as resident:` -- the session is ended on the way out whatever happens inside (an exception
        between _begin and _end would otherwise leave the device's resident slot to the session until its grid's bounded wait has run
        out: other contexts then run without resident grids for up to 2 s).  `resident` = whether a session could be opened."""
        opened = self.score_session_begin(kind, thre_3d, cos_thr, cos_nl, mode)
        try:
            yield opened
        finally:
            if opened and self._h:
                self.score_session_end()

    def nl_round(self, c_opt, Cw, Cc, Rwc) -> np.ndarray:
        a = [np.ascontiguousarray(x, np.float64) for x in (c_opt, Cw, Cc, Rwc)]
        out = np.zeros(44)
        L.check(L.lib().rpe_nl_round(self._h, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out)))
        return out


def comm_unique_id() -> bytes:
    buf = (C.c_char * 128)()
    L.check(L.lib().rpe_comm_unique_id(buf))
    return bytes(buf)


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


# ---- adapter-level pipelines (rpe_run): method / ls ids shared with the oracle's C API
M_SHINJI_RANSAC, M_SHINJI_RANSAC2, M_SHINJI_PROSAC, M_KNEIP_RANSAC, M_KNEIP_PROSAC = 0, 1, 2, 3, 4
M_SK_RANSAC, M_SK_PROSAC, M_NL_KNEIP_RANSAC, M_NL_SHINJI_RANSAC, M_NL_SK_RANSAC, M_NONE = 5, 6, 7, 8, 9, 10
LS_NONE, LS_SHINJI_INLIERS, LS_NL_BUGCOMPAT, LS_NL_FIXED, LS_SHINJI_ALL, LS_GN_P2P, LS_GN_JOINT, LS_GN_P2PLANE, LS_GN_BEARING, LS_GN_REPROJ = range(10)


def run(method, dtype=L.F32, xw=None, xc=None, bv=None, nw=None, nc=None, weights=None, f=585.0, thre_3d=0.0, thre_2d=0.0, thre_nl=0.0,
        iters=0, confidence=0.99, seed=1, ls=LS_NONE, score_mode=L.SCORE_EXACT, mask_in=None, pose_in=None, max_votes_in=1, want_masks=True):
    """Run one solver of pose/*.hpp on a freshly built adapter (AOOnly / PnP / AO / NormalAO chosen like the
    reference's demos do).  Returns dict(R, t, iters, max_votes, masks[3, n]); want_masks=False leaves the masks on the device
    (mask_out = NULL: no read-back), masks is then None."""
    dt = _np_dtype(dtype)
    arrs = {k: (None if a is None else np.ascontiguousarray(a, dtype=dt)) for k, a in dict(xw=xw, xc=xc, bv=bv, nw=nw, nc=nc).items()}
    n = len(arrs["xw"])
    w = None if weights is None else np.asfortranarray(weights, dtype=dt)
    prob = L.RpeProblem(n, dtype, _p(arrs["bv"]), _p(arrs["xc"]), _p(arrs["nc"]), _p(arrs["xw"]), _p(arrs["nw"]), _p(w),
                        0 if w is None else w.shape[1], f, f)
    R, t = np.zeros(9), np.zeros(3)
    if pose_in is not None:
        R[:] = np.asarray(pose_in[0], float).reshape(9)
        t[:] = np.asarray(pose_in[1], float)
    else:
        R[:] = np.eye(3).reshape(9)
    it, mv = C.c_int(iters), C.c_int(max_votes_in)
    mi = None if mask_in is None else np.ascontiguousarray(mask_in, dtype=np.int16)
    mo = np.zeros((3, n), np.int16) if want_masks else None
    L.check(L.lib().rpe_run(method, C.byref(prob), thre_3d, thre_2d, thre_nl, C.byref(it), confidence, seed, ls, score_mode, _p(mi), _p(R),
                            _p(t), C.byref(mv), _p(mo)))
    return dict(R=R.reshape(3, 3), t=t, iters=it.value, max_votes=mv.value, masks=mo)


def _problem(dtype, xw, xc, bv, nw, nc, weights, f):
    dt = _np_dtype(dtype)
    arrs = {k: (None if a is None else np.ascontiguousarray(a, dtype=dt)) for k, a in dict(xw=xw, xc=xc, bv=bv, nw=nw, nc=nc).items()}
    n = len(arrs["xw"])
    w = None if weights is None else np.asfortranarray(weights, dtype=dt)
    prob = L.RpeProblem(n, dtype, _p(arrs["bv"]), _p(arrs["xc"]), _p(arrs["nc"]), _p(arrs["xw"]), _p(arrs["nw"]), _p(w),
                        0 if w is None else w.shape[1], f, f)
    return prob, n, (arrs, w)   # the last item keeps the buffers alive


def host_hypotheses(method, dtype=L.F32, xw=None, xc=None, bv=None, nw=None, nc=None, weights=None, f=585.0, iters=0, seed=1):
    """rpe_host_hypotheses: the hypothesis stream `method` generates in `iters` iterations (no GPU).  Returns (q7[H, 7], first[iters + 1])."""
    prob, n, keep = _problem(dtype, xw, xc, bv, nw, nc, weights, f)
    cap = 3 * iters + 1
    q7, first = np.zeros((cap, 7)), np.zeros(iters + 1, np.int32)
    H = L.lib().rpe_host_hypotheses(method, C.byref(prob), iters, seed, _p(q7), cap, _p(first))
    if H < 0:
        L.check(H)
    return q7[:H].copy(), first


def run_replay(method, poses7, first, dtype=L.F32, xw=None, xc=None, bv=None, nw=None, nc=None, weights=None, f=585.0, thre_3d=0.0, thre_2d=0.0,
               thre_nl=0.0, iters=0, confidence=0.99, ls=LS_NONE, score_mode=L.SCORE_EXACT):
    """rpe_run_replay: rpe_run with the hypotheses of iteration i taken from poses7[first[i]:first[i+1]]."""
    prob, n, keep = _problem(dtype, xw, xc, bv, nw, nc, weights, f)
    poses7 = np.ascontiguousarray(poses7, np.float64).reshape(-1, 7)
    first = np.ascontiguousarray(first, np.int32)
    R, t = np.eye(3).reshape(9).copy(), np.zeros(3)
    it, mv = C.c_int(iters), C.c_int(0)
    mo = np.zeros((3, n), np.int16)
    L.check(L.lib().rpe_run_replay(method, C.byref(prob), _p(poses7), _p(first), len(first) - 1, thre_3d, thre_2d, thre_nl, C.byref(it), confidence, ls,
                                   score_mode, _p(R), _p(t), C.byref(mv), _p(mo)))
    return dict(R=R.reshape(3, 3), t=t, iters=it.value, max_votes=mv.value, masks=mo)


class HostExchange:
    """The host-side exchange by itself (no GPU): all-reduce of up to 64 doubles / 8192 int32 between the rank processes of one node
    through a POSIX shared-memory segment; every rank gets bitwise the same sums (rank order)."""

    def __init__(self, name: str, world: int, rank: int, create: bool, timeout_s: float = 10.0):
        self._h = C.c_void_p()
        L.check(L.lib().rpe_host_exchange_open(name.encode(), world, rank, 1 if create else 0, float(timeout_s), C.byref(self._h)))
        self.world, self.rank = world, rank

    def allreduce_f64(self, v) -> np.ndarray:
        a = np.ascontiguousarray(v, np.float64).copy()
        L.check(L.lib().rpe_host_exchange_allreduce_f64(self._h, _p(a), a.size))
        return a

    def allreduce_i32(self, v) -> np.ndarray:
        a = np.ascontiguousarray(v, np.int32).copy()
        L.check(L.lib().rpe_host_exchange_allreduce_i32(self._h, _p(a), a.size))
        return a

    def set_label(self, label: str):
        L.check(L.lib().rpe_host_exchange_set_label(self._h, label.encode()))

    def labels_collide(self) -> bool:
        return bool(L.lib().rpe_host_exchange_labels_collide(self._h))

    def unlink(self):
        L.check(L.lib().rpe_host_exchange_unlink(self._h))

    def close(self):
        if self._h:
            L.lib().rpe_host_exchange_close(self._h)
            self._h = C.c_void_p()
