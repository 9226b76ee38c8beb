"""Shared helpers for the parity tests (numpy only)."""
import math

import numpy as np

from rgbd_pose_estimation_amd import simulator as S

# tolerances stated by BASELINE.json north_star: pose vs the CPU reference
ROT_TOL_RAD = 1e-5
TRANS_REL_TOL = 1e-4


def rot_err(Ra, Rb):
    """rotation angle of Ra Rb^T [rad], accurate for tiny angles (calc_err semantics, AbsoluteOrientation.hpp:29-43)."""
    D = np.asarray(Ra, float) @ np.asarray(Rb, float).T
    s = np.linalg.norm([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]]) / 2
    c = (np.trace(D) - 1) / 2
    return math.atan2(s, c)


def trans_rel_err(ta, tb):
    return float(np.linalg.norm(np.asarray(ta, float) - np.asarray(tb, float)) / np.linalg.norm(tb))


def scene33(seed, n, dtype=np.float32, noise=0.05, outliers=0.1):
    rng = np.random.default_rng(seed)
    R, t = S.random_pose(rng)
    return S.simulate_3d_3d_correspondences(rng, R, t, n, noise, outliers).astype(dtype)


def scene_full(seed, n, dtype=np.float32, n2d=15.0, n3d=0.05, nnl_deg=2.0, outliers=0.1, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    R, t = S.random_pose(rng)
    sc = S.simulate_2d_3d_nl_correspondences(rng, R, t, n, n2d, outliers, n3d, outliers, math.radians(nnl_deg), outliers)
    if nan_frac > 0:
        idx = rng.permutation(n)[: max(1, int(nan_frac * n))]
        sc.P[idx] = np.nan  # "no 3-D measurement": AOPoseAdapter::isValid (AOPoseAdapter.hpp:147-152)
    return sc.astype(dtype)


def perturbed_pose(rng, R, t, ang=0.02, dt=0.05):
    if ang == 0:
        return R, t + dt * rng.standard_normal(3)
    w = rng.standard_normal(3); w *= ang / np.linalg.norm(w)
    th = np.linalg.norm(w); K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    dR = np.eye(3) + math.sin(th) / th * K + (1 - math.cos(th)) / th**2 * K @ K
    return dR @ R, t + dt * rng.standard_normal(3)


def unpack_ne(rec):
    H = np.zeros((6, 6)); k = 0
    for a in range(6):
        for b in range(a, 6):
            H[a, b] = H[b, a] = rec[k]; k += 1
    return H, np.array(rec[21:27]), rec[27], rec[28]
