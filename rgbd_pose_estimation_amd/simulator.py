"""Seeded numpy restatement of the reference's synthetic scene generator (S1).

Follows the generative model of /root/reference/pose/Simulator.hpp (function names kept), vectorised and
driven by an explicit ``numpy.random.Generator`` instead of the reference's unseeded global
``std::default_random_engine`` + ``Eigen::Random`` (Simulator.hpp:13-14,19,33,140), so exact random
streams differ by construction; distributions, parameters and outlier placement are the reference's.

Arrays are returned the way the adapters take them: ``3 x N`` column-major, i.e. numpy shape ``(N, 3)``
C-contiguous (xyz interleaved), dtype float32 or float64.  Convention: ``Xc = R_cw @ Xw + t_w``
(AbsoluteOrientation.hpp:51).  Used by bench.py, tests and smoke(); inputs only -- no solver code here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


def generate_random_translation_uniform(rng: np.random.Generator, size: float) -> np.ndarray:
    """Simulator.hpp:16-21 -- size * U[-1,1]^3."""
    return size * rng.uniform(-1.0, 1.0, 3)


def _rot_zyx(rx, ry, rz):
    """R = Rz(rz) @ Ry(ry) @ Rx(rx), batched (Simulator.hpp:48-81)."""
    rx, ry, rz = np.broadcast_arrays(np.asarray(rx, float), np.asarray(ry, float), np.asarray(rz, float))
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    R = np.empty(rx.shape + (3, 3))
    R[..., 0, 0] = cz * cy
    R[..., 0, 1] = cz * sy * sx - sz * cx
    R[..., 0, 2] = cz * sy * cx + sz * sx
    R[..., 1, 0] = sz * cy
    R[..., 1, 1] = sz * sy * sx + cz * cx
    R[..., 1, 2] = sz * sy * cx - cz * sx
    R[..., 2, 0] = -sy
    R[..., 2, 1] = cy * sx
    R[..., 2, 2] = cy * cx
    return R


def generate_random_rotation(rng: np.random.Generator, max_angle_radian: float, use_gaussian: bool = True, size=None) -> np.ndarray:
    """Simulator.hpp:23-83: rv ~ N(0,1)^3 or U[-1,1]^3; (rx, ry, rz) = a*(rv0, rv1/2, rv2), clamped."""
    shape = (3,) if size is None else (size, 3)
    rv = rng.standard_normal(shape) if use_gaussian else rng.uniform(-1.0, 1.0, shape)
    rx = np.clip(max_angle_radian * rv[..., 0], -math.pi, math.pi)
    ry = np.clip(max_angle_radian * rv[..., 1] * 0.5, -math.pi / 2, math.pi / 2)
    rz = np.clip(max_angle_radian * rv[..., 2], -math.pi, math.pi)
    return _rot_zyx(rx, ry, rz)


def simulate_rand_point_cloud_in_frustum(rng, number: int, f: float, min_depth: float, max_depth: float) -> np.ndarray:
    """Simulator.hpp:135-173: rejection-sample points in the 640x480 frustum; returns (number, 3) float64."""
    tx, ty = 320.0 / f, 240.0 / f
    out = np.empty((number, 3))
    filled = 0
    while filled < number:
        m = max(1024, int((number - filled) * 2.2))
        u = rng.uniform(-1.0, 1.0, (m, 3))
        P = np.empty_like(u)
        P[:, 0] = u[:, 0] * tx * max_depth
        P[:, 1] = u[:, 1] * ty * max_depth
        P[:, 2] = (u[:, 2] + 1.0) / 2.0 * (max_depth - min_depth) + min_depth
        ok = (np.abs(P[:, 0] / P[:, 2]) < tx) & (np.abs(P[:, 1] / P[:, 2]) < ty)
        P = P[ok][: number - filled]
        out[filled : filled + len(P)] = P
        filled += len(P)
    return out


def _noise(rng, n, dim, use_gaussian):
    return rng.standard_normal((n, dim)) if use_gaussian else rng.uniform(-1.0, 1.0, (n, dim))


def _outlier_indices(rng, number, ratio):
    out = int(ratio * number + 0.5)
    return rng.permutation(number)[:out], out


@dataclass
class Scene:
    R: np.ndarray           # 3x3 ground truth R_cw
    t: np.ndarray           # 3   ground truth t_w
    Q: np.ndarray = None    # world points  (N,3)  "points_g"
    P: np.ndarray = None    # camera points (N,3)  "points_c"
    U: np.ndarray = None    # bearing vectors (N,3)
    M: np.ndarray = None    # world normals (N,3)   "normal_g"
    N: np.ndarray = None    # camera normals (N,3)  "normal_c"
    weights: np.ndarray = None  # (N,3): columns 2D-3D, 3D-3D, N-N (Fortran order = N x 3 column-major)
    f: float = 585.0

    def astype(self, dt):
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
        w = None if self.weights is None else np.asfortranarray(self.weights, dtype=dt)
        return Scene(self.R, self.t, c(self.Q), c(self.P), c(self.U), c(self.M), c(self.N), w, self.f)


def simulate_3d_3d_correspondences(rng, R, t, number, noise, outlier_ratio, min_depth=0.4, max_depth=8.0, f=585.0, use_gaussian=True) -> Scene:
    """Simulator.hpp:268-314.  Q = R^-1 (P_gt - t) + noise*rv ; outliers overwrite random Q columns with
    fresh frustum points ("outliers remain in CRS"); P_gt stays clean.  Used as AOOnlyPoseAdapter(P, Q)."""
    P_gt = simulate_rand_point_cloud_in_frustum(rng, number, f, min_depth, max_depth)
    Q = (P_gt - t) @ R  # row form of R^T (P - t)
    rv = _noise(rng, number, 3, use_gaussian)
    w = np.ones((number, 3))
    w[:, 1] = 1.0 / np.linalg.norm(rv, axis=1)
    Q = Q + noise * rv
    idx, out = _outlier_indices(rng, number, outlier_ratio)
    Q[idx] = simulate_rand_point_cloud_in_frustum(rng, out, f, min_depth, max_depth)
    return Scene(R, t, Q=Q, P=P_gt, weights=w, f=f)


def simulate_2d_3d_correspondences(rng, R, t, number, noise, outlier_ratio, min_depth=0.4, max_depth=8.0, f=585.0, use_gaussian=True) -> Scene:
    """Simulator.hpp:175-233: pixel noise, outlier pixels from fresh points, bearing = normalize(kp_x, kp_y, f)."""
    P_gt = simulate_rand_point_cloud_in_frustum(rng, number, f, min_depth, max_depth)
    kp = f * P_gt[:, :2] / P_gt[:, 2:3]
    Q = (P_gt - t) @ R
    rv = _noise(rng, number, 2, use_gaussian)
    w = np.ones((number, 3))
    w[:, 0] = 1.0 / np.linalg.norm(rv, axis=1)
    kp = kp + noise * rv
    idx, out = _outlier_indices(rng, number, outlier_ratio)
    op = simulate_rand_point_cloud_in_frustum(rng, out, f, min_depth, max_depth)
    kp[idx] = f * op[:, :2] / op[:, 2:3]
    U = np.concatenate([kp, np.full((number, 1), f)], axis=1)
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    return Scene(R, t, Q=Q, P=P_gt, U=U, weights=w, f=f)


def simulate_2d_3d_3d_correspondences(rng, R, t, number, noise_2d, noise_3d, outlier_ratio, min_depth=0.4, max_depth=8.0, f=585.0, use_gaussian=True) -> Scene:
    """Simulator.hpp:235-265: the 2D-3D scene, then Q += noise_3d * rv.  Used as AOPoseAdapter(U, P, Q)."""
    s = simulate_2d_3d_correspondences(rng, R, t, number, noise_2d, outlier_ratio, min_depth, max_depth, f, use_gaussian)
    rv = _noise(rng, number, 3, use_gaussian)
    s.weights[:, 1] = 1.0 / np.linalg.norm(rv, axis=1)
    s.Q = s.Q + noise_3d * rv
    return s


def simulate_nl_nl_correspondences(rng, R, number, noise_nl, outlier_ratio_nl, use_gaussian=True):
    """Simulator.hpp:85-130.  Returns (M world normals, N camera normals, N_gt, w).  Outliers overwrite the
    FIRST `out` columns (index i, not vIdx[i], :114-119) -- reproduced."""
    Ngt = np.empty((number, 3)); Mw = np.empty((number, 3)); Nc = np.empty((number, 3))
    todo = np.arange(number)
    down = np.array([0.0, 0.0, -1.0])
    while len(todo):
        k = len(todo)
        g = generate_random_rotation(rng, math.pi / 2, False, size=k) @ down
        g /= np.linalg.norm(g, axis=1, keepdims=True)
        m = g @ R  # R^-1 g
        m /= np.linalg.norm(m, axis=1, keepdims=True)
        nz = np.einsum("kij,kj->ki", generate_random_rotation(rng, noise_nl, use_gaussian, size=k), g)
        nz /= np.linalg.norm(nz, axis=1, keepdims=True)
        ok = ~(np.arccos(np.clip(nz[:, 2], -1, 1)) < math.pi / 2)
        Ngt[todo[ok]] = g[ok]; Mw[todo[ok]] = m[ok]; Nc[todo[ok]] = nz[ok]
        todo = todo[~ok]
    w = np.einsum("ij,ij->i", Nc, Ngt)
    out = int(outlier_ratio_nl * number + 0.5)
    rng.permutation(number)  # the reference draws (and ignores) vIdx
    todo = np.arange(out)
    while len(todo):
        k = len(todo)
        g = generate_random_rotation(rng, math.pi / 2, False, size=k) @ down
        g /= np.linalg.norm(g, axis=1, keepdims=True)
        ok = ~(np.arccos(np.clip(g[:, 2], -1, 1)) < math.pi / 2)
        Nc[todo[ok]] = g[ok]
        todo = todo[~ok]
    return Mw, Nc, Ngt, w


def simulate_2d_3d_nl_correspondences(rng, R, t, number, n2D, or_2D, n3D, or_3D, nNl, or_Nl, min_depth=0.4, max_depth=8.0, f=585.0, use_gaussian=True) -> Scene:
    """Simulator.hpp:316-367: 2D-3D scene (clean Q), normals, then P = P_gt + n3D*rv with 3-D outliers on P.
    Used as NormalAOPoseAdapter(U, P, N, Q, M)."""
    s = simulate_2d_3d_correspondences(rng, R, t, number, n2D, or_2D, min_depth, max_depth, f, use_gaussian)
    Mw, Nc, _, wn = simulate_nl_nl_correspondences(rng, R, number, nNl, or_Nl, True)
    s.weights[:, 2] = wn
    rv = _noise(rng, number, 3, use_gaussian)
    s.weights[:, 1] = 1.0 / np.linalg.norm(rv, axis=1)
    P = s.P + n3D * rv
    idx, out = _outlier_indices(rng, number, or_3D)
    P[idx] = simulate_rand_point_cloud_in_frustum(rng, out, f, min_depth, max_depth)
    s.P, s.M, s.N = P, Mw, Nc
    return s


def lateral_noise_kinect(theta, z, f):
    """Simulator.hpp:369-377 (Nguyen et al. 2012)."""
    return (0.8 + 0.035 * theta / (math.pi / 2 - theta)) * z / f


def axial_noise_kinect(theta, z):
    """Simulator.hpp:379-387."""
    base = 0.0012 + 0.0019 * (z - 0.4) ** 2
    extra = 0.0001 * theta * theta / np.sqrt(z) / (math.pi / 2 - theta) ** 2
    return np.where(np.abs(theta) <= math.pi / 3, base, base + extra)


def simulate_kinect_2d_3d_nl_correspondences(rng, R, t, number, noise_2d, or_2d, or_3d, noise_nl, or_nl, min_depth=0.4, max_depth=3.0, f=585.0) -> Scene:
    """Simulator.hpp:389-436: Kinect lateral/axial noise on P, weight = sigma_min / sigma_a."""
    s = simulate_2d_3d_correspondences(rng, R, t, number, noise_2d, or_2d, min_depth, max_depth, f, True)
    Mw, Nc, Ngt, wn = simulate_nl_nl_correspondences(rng, R, number, noise_nl, or_nl, True)
    s.weights[:, 2] = wn
    sigma_min = float(axial_noise_kinect(np.array(0.0), np.array(min_depth)))
    theta = np.arccos(np.clip(Ngt @ np.array([0.0, 0.0, -1.0]), -1, 1))
    z = s.P[:, 2]
    sl = lateral_noise_kinect(theta, z, f)
    sa = axial_noise_kinect(theta, z)
    g = rng.standard_normal((number, 3))
    P = s.P + np.stack([sl * g[:, 0], sl * g[:, 1], sa * g[:, 2]], axis=1)
    s.weights[:, 1] = sigma_min / sa
    idx, out = _outlier_indices(rng, number, or_3d)
    P[idx] = simulate_rand_point_cloud_in_frustum(rng, out, f, min_depth, max_depth)
    s.P, s.M, s.N = P, Mw, Nc
    return s


def random_pose(rng):
    """The pose every demo draws: t = 5*U[-1,1]^3, R = generate_random_rotation(pi/2, uniform) (SimpleMain.cpp:23-24)."""
    t = generate_random_translation_uniform(rng, 5.0)
    R = generate_random_rotation(rng, math.pi / 2, False)
    return R, t


def dense_depth_scene(seed: int, number: int = 307200, noise_3d: float = 0.05, outlier_ratio: float = 0.1, dtype=np.float32) -> Scene:
    """BASELINE config 2 shape: `number` 3D-3D correspondences (640x480 = 307200), sigma 0.05 m, 10 % outliers."""
    rng = np.random.default_rng(seed)
    R, t = random_pose(rng)
    return simulate_3d_3d_correspondences(rng, R, t, number, noise_3d, outlier_ratio).astype(dtype)


# ---------------------------------------------------------------------------------------------------------------------
# Depth frames for the front end (rpe_frame_set_depth / rpe_associate / rpe_icp).  The reference simulator only draws
# sparse points inside the frustum of its 640 x 480, f = 585 camera (Simulator.hpp:157-173); a dense frame of the same
# camera is rendered here by casting one ray per pixel into a small analytic scene (a room with spheres in it), so that
# every pixel has an exact depth and the scene constrains all six degrees of freedom.
# ---------------------------------------------------------------------------------------------------------------------
DEFAULT_CAMERA = (585.0, 585.0, 320.0, 240.0, 640, 480)


def default_room():
    """(box_min, box_max, spheres[k, 4] = centre xyz + radius): the camera sits inside the box."""
    spheres = np.array([[-0.9, 0.3, 2.6, 0.55], [0.8, -0.4, 3.4, 0.7], [0.1, 0.9, 2.0, 0.35], [1.6, 0.8, 4.2, 0.5], [-1.7, -0.7, 3.9, 0.6]])
    return np.array([-2.5, -1.6, -1.0]), np.array([2.7, 1.5, 5.0]), spheres


def render_depth(R, t, cam=DEFAULT_CAMERA, room=None, noise_sigma: float = 0.0, rng=None, as_u16: bool = False):
    """Depth image (height, width) of the room seen by the camera Xc = R Xw + t: float32 metres, or uint16 millimetres.
    noise_sigma: Gaussian depth noise in metres (rng required)."""
    fx, fy, cx, cy, w, h = cam
    lo, hi, spheres = room if room is not None else default_room()
    R = np.asarray(R, np.float64)
    t = np.asarray(t, np.float64)
    C0 = -R.T @ t                                    # camera centre in the world
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1).reshape(-1, 3)
    d = d_cam @ R                                    # world direction of each pixel's ray (rows: R^T d_cam); z_cam = lambda
    lam = np.full(len(d), np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        for ax in range(3):                          # the six walls, seen from inside
            for bound in (lo[ax], hi[ax]):
                l = (bound - C0[ax]) / d[:, ax]
                p = C0 + l[:, None] * d
                ok = l > 1e-9
                for o in range(3):
                    if o != ax:
                        ok &= (p[:, o] >= lo[o] - 1e-9) & (p[:, o] <= hi[o] + 1e-9)
                lam = np.where(ok & (l < lam), l, lam)
        for sx, sy, sz, r in spheres:
            oc = C0 - np.array([sx, sy, sz])
            a = np.sum(d * d, 1)
            b = 2.0 * d @ oc
            c = oc @ oc - r * r
            disc = b * b - 4 * a * c
            l = (-b - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2 * a)
            ok = (disc > 0) & (l > 1e-9)
            lam = np.where(ok & (l < lam), l, lam)
    z = np.where(np.isfinite(lam), lam, 0.0)
    if noise_sigma > 0:
        z = np.where(z > 0, z + noise_sigma * rng.standard_normal(z.shape), 0.0)
    z = z.reshape(h, w)
    if as_u16:
        return np.clip(np.rint(z * 1000.0), 0, 65535).astype(np.uint16)
    return z.astype(np.float32)
