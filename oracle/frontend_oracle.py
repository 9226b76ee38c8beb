"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

numpy statement of the FRONT END (back-projection, normals, projective data association),
the contract the gfx950 kernels of csrc/rpe_frontend.hip are held to BIT-EXACTLY.

PARITY UNPINNED: the reference has no such stage -- its arrays come from pose/Simulator.hpp or from the caller -- so
there is no reference text, golden vector or runnable reference to pin this file to.  What the reference does fix is
the camera model, and that is followed: pinhole with u - cx = f X / Z (project_point_cloud, Simulator.hpp:150-155),
640 x 480 with the principal point at the centre and f = 585 (Simulator.hpp:160-162, SimpleMain.cpp:42-44), bearing
vector = normalize((u - cx)/f, (v - cy)/f, 1) (Simulator.hpp:215-222), pose convention Xc = R Xw + t
(AbsoluteOrientation.hpp:51), invalid camera point = NaN column (AOPoseAdapter.hpp:147-152).  Beyond that the file is
pinned by analytic cases in tests/test_frontend_oracle.py (planes, spheres: exact vertices and normals; identity
association; ICP recovering a known motion).

Every expression is evaluated in IEEE fp32 in a fixed order (numpy evaluates a*b + c*d + e*f as ((a*b)+(c*d))+(e*f), the
same left-to-right order the kernels are written in, with FMA contraction disabled there)."""
import numpy as np

F = np.float32


def _cam(cam):
    fx, fy, cx, cy, w, h = cam
    return F(fx), F(fy), F(cx), F(cy), int(w), int(h)


def _vertex(depth_m, cam):
    """vertex map (h, w, 3) and validity from a metric depth image (fp32)."""
    fx, fy, cx, cy, w, h = _cam(cam)
    u = np.arange(w, dtype=F)[None, :]
    v = np.arange(h, dtype=F)[:, None]
    xn = (u - cx) / fx
    yn = (v - cy) / fy
    return xn, yn, np.stack([np.broadcast_to(xn, (h, w)) * depth_m, np.broadcast_to(yn, (h, w)) * depth_m, depth_m], -1)


def frame_maps(depth, cam, depth_scale, dmin, dmax, max_jump):
    """F1.  depth (h, w) uint16 or float32 -> vertex, normal, bearing maps, each (h*w, 3) float32."""
    fx, fy, cx, cy, w, h = _cam(cam)
    with np.errstate(invalid="ignore", divide="ignore"):
        z = depth.astype(F) * F(depth_scale)
        ok = (z > F(dmin)) & (z < F(dmax))
        xn, yn, V = _vertex(z, cam)
        s = np.sqrt(xn * xn + yn * yn + F(1.0))
        B = np.stack([np.broadcast_to(xn / s, (h, w)), np.broadcast_to(yn / s, (h, w)), np.broadcast_to(F(1.0) / s, (h, w))], -1).astype(F)
        Vout = np.where(ok[..., None], V, F(np.nan)).astype(F)
        N = np.full((h, w, 3), np.nan, F)
        if w > 2 and h > 2:
            c = V[1:-1, 1:-1]
            l, r, t, b = V[1:-1, :-2], V[1:-1, 2:], V[:-2, 1:-1], V[2:, 1:-1]
            okc = ok[1:-1, 1:-1] & ok[1:-1, :-2] & ok[1:-1, 2:] & ok[:-2, 1:-1] & ok[2:, 1:-1]
            mj = F(max_jump)
            for nb in (l, r, t, b):
                okc = okc & (np.abs(nb[..., 2] - c[..., 2]) <= mj)
            a = r - l      # d/du
            e = b - t      # d/dv
            nx = e[..., 1] * a[..., 2] - e[..., 2] * a[..., 1]
            ny = e[..., 2] * a[..., 0] - e[..., 0] * a[..., 2]
            nz = e[..., 0] * a[..., 1] - e[..., 1] * a[..., 0]
            ln = np.sqrt(nx * nx + ny * ny + nz * nz)
            okc = okc & (ln > F(0))
            nx, ny, nz = nx / ln, ny / ln, nz / ln
            facing = nx * c[..., 0] + ny * c[..., 1] + nz * c[..., 2]
            flip = facing > F(0)
            nx, ny, nz = np.where(flip, -nx, nx), np.where(flip, -ny, ny), np.where(flip, -nz, nz)
            N[1:-1, 1:-1] = np.where(okc[..., None], np.stack([nx, ny, nz], -1), F(np.nan))
    return Vout.reshape(-1, 3), N.reshape(-1, 3), B.reshape(-1, 3)


def _pose_f(pose12):
    p = np.asarray(pose12, np.float64).reshape(12).astype(F)
    return p[:9], p[9:]


def _to_world(R, t, X):
    dx, dy, dz = X[:, 0] - t[0], X[:, 1] - t[1], X[:, 2] - t[2]
    return np.stack([R[0] * dx + R[3] * dy + R[6] * dz, R[1] * dx + R[4] * dy + R[7] * dz, R[2] * dx + R[5] * dy + R[8] * dz], -1)


def _rot_to_world(R, X):
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([R[0] * x + R[3] * y + R[6] * z, R[1] * x + R[4] * y + R[7] * z, R[2] * x + R[5] * y + R[8] * z], -1)


def to_world(V, N, pose12):
    """F2.  frame maps -> world maps: Xw = R^T (Xc - t), Nw = R^T Nc."""
    R, t = _pose_f(pose12)
    with np.errstate(invalid="ignore"):
        return _to_world(R, t, V).astype(F), _rot_to_world(R, N).astype(F)


def associate(V, N, B, MV, MN, mcam, pose12, mpose12, dist_thr, cos_thr, use_normals=True):
    """F3.  Returns XW, XC, BV, NW, NC (each (pixels, 3) float32) and the number of pairs."""
    fx, fy, cx, cy, w, h = _cam(mcam)
    R, t = _pose_f(pose12)
    Rm, tm = _pose_f(mpose12)
    d = F(dist_thr)
    dist_sq = d * d
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        ok = ~np.isnan(V).any(1)
        W = _to_world(R, t, V)
        wx, wy, wz = W[:, 0], W[:, 1], W[:, 2]
        px = Rm[0] * wx + Rm[1] * wy + Rm[2] * wz + tm[0]
        py = Rm[3] * wx + Rm[4] * wy + Rm[5] * wz + tm[1]
        pz = Rm[6] * wx + Rm[7] * wy + Rm[8] * wz + tm[2]
        ok &= pz > F(0)
        uf = np.floor(fx * (px / pz) + cx + F(0.5))
        vf = np.floor(fy * (py / pz) + cy + F(0.5))
        ok &= (uf >= F(0)) & (uf <= F(w - 1)) & (vf >= F(0)) & (vf <= F(h - 1))
        j = np.where(ok, vf, 0).astype(np.int64) * w + np.where(ok, uf, 0).astype(np.int64)
        M, G = MV[j], MN[j]
        ex, ey, ez = M[:, 0] - wx, M[:, 1] - wy, M[:, 2] - wz
        ok &= (ex * ex + ey * ey + ez * ez) <= dist_sq
        if use_normals:
            Q = _rot_to_world(R, N)
            ok &= (Q[:, 0] * G[:, 0] + Q[:, 1] * G[:, 1] + Q[:, 2] * G[:, 2]) >= F(cos_thr)
    o = ok[:, None]
    nan = F(np.nan)
    return (np.where(o, M, F(0)).astype(F), np.where(o, V, nan).astype(F), np.where(o, B, nan).astype(F), np.where(o, G, F(0)).astype(F),
            np.where(o, N, nan).astype(F), int(ok.sum()))
