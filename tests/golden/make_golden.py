#!/usr/bin/env python3
"""Generate tests/golden/golden_inputs.npz + golden.json: expected outputs for the hot path computed by an
INDEPENDENT numpy/scipy implementation (no code shared with oracle/ or the product).  Run in the build container:

    python tests/golden/make_golden.py

Why this exists: the reference (/root/reference) ships no golden vectors and cannot be compiled here (Eigen3 and
OpenCV are absent), so the CPU oracle cannot be pinned against reference outputs ("parity unpinned").  It is pinned
instead against (1) this independent implementation of the same published algorithms and (2) analytic known-answer
cases.  Nothing in this script reads /root/reference; the algorithms are restated from the papers / SURVEY.md.
"""
import json
import math
import os
import sys

import numpy as np
from scipy.linalg import expm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rgbd_pose_estimation_amd import simulator as S  # noqa: E402  (input generation only)

M64 = (1 << 64) - 1


class PCG32:
    """O'Neill's PCG32 XSH-RR 64/32, pure Python."""

    def __init__(self, seed, seq=54):
        self.state, self.inc = 0, ((seq << 1) | 1) & M64
        self.next32()
        self.state = (self.state + seed) & M64
        self.next32()

    def next32(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & M64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def rand31(self):
        return self.next32() >> 1


def random_elements(n, m, rng):
    idx = list(range(n))
    out = []
    for j in range(n - 1, n - m - 1, -1):
        r = rng.rand31() % (j + 1)
        idx[r], idx[j] = idx[j], idx[r]
        out.append(idx[j])
    return out


class Prosac:
    def __init__(self, m, N, real=float):
        self.m, self.N, self.t, self.real = m, N, 1, real

    def sample(self, rng):
        real = self.real
        t_n = real(20000)
        n = self.m
        for i in range(self.m):
            t_n = real(t_n * (real(n - i) / real(self.N - i)))
        t_n_prime = real(1.0)
        for t in range(1, self.t + 1):
            if t > t_n_prime and n < self.N:
                nxt = real((t_n * real(n + 1.0)) / real(n + 1.0 - self.m))
                t_n_prime = real(t_n_prime + math.ceil(nxt - t_n))
                t_n = nxt
                n += 1
        out = []
        if t_n_prime < self.t:
            for _ in range(self.m):
                while True:
                    r = rng.rand31() % n
                    if r not in out:
                        break
                out.append(r)
        else:
            for _ in range(self.m - 1):
                while True:
                    r = rng.rand31() % (n - 1)
                    if r not in out:
                        break
                out.append(r)
            out.append(n if n < self.N else self.N - 1)
        self.t += 1
        return out


def update_num_iters(p, ep, model_points, max_iters):
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).eps)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).eps:
        return 0
    num, denom = math.log(num), math.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(num / denom + 0.5)


def kabsch(xw, xc):
    cw, cc = xw.mean(0), xc.mean(0)
    M = (xc - cc).T @ (xw - cw)
    U, _, Vt = np.linalg.svd(M)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    return R, cc - R @ cw


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def se3_exp(a):
    H = np.zeros((4, 4))
    H[:3, :3] = hat(a[3:])
    H[:3, 3] = a[:3]
    E = expm(H)
    return E[:3, :3], E[:3, 3]


def residuals(kind, R, t, a, b, c):
    p = a @ R.T + t
    if kind == 0:
        return (p - b).reshape(-1)
    if kind == 1:
        return np.einsum("ij,ij->i", c, p - b)
    if kind == 4:   # pixel reprojection in normalised image coordinates (SURVEY.md Appendix B row 4 with f = 1)
        return (p[:, :2] / p[:, 2:3] - b[:, :2] / b[:, 2:3]).reshape(-1)
    ph = p / np.linalg.norm(p, axis=1, keepdims=True)
    return np.cross(ph, b).reshape(-1)


def normal_eq_numeric(kind, R, t, a, b, c, h=1e-6):
    """H = J^T J, g = J^T r with J from central differences of r(exp(delta) T)."""
    r0 = residuals(kind, R, t, a, b, c)
    J = np.zeros((len(r0), 6))
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        Rp, tp = se3_exp(d)
        Rm, tm = se3_exp(-d)
        J[:, k] = (residuals(kind, Rp @ R, Rp @ t + tp, a, b, c) - residuals(kind, Rm @ R, Rm @ t + tm, a, b, c)) / (2 * h)
    return J.T @ J, J.T @ r0, float(r0 @ r0)


def votes_numpy(kind, R, t, sc, thre_3d, cos_thr, cos_nl):
    """kind bits: 1 = 3D-3D, 2 = 2D-3D, 4 = N-N; returns (votes, masks[3, n])."""
    n = len(sc["Q"])
    valid = ~np.isnan(sc["P"]).all(1)
    p = sc["Q"] @ R.T + t
    m = np.zeros((3, n), np.int16)
    if kind & 1:
        m[1] = valid & (np.linalg.norm(np.nan_to_num(sc["P"]) - p, axis=1) < thre_3d)
    if kind & 2:
        ph = p / np.linalg.norm(p, axis=1, keepdims=True)
        m[0] = np.einsum("ij,ij->i", ph, sc["U"]) > cos_thr
    if kind & 4:
        m[2] = valid & (np.einsum("ij,ij->i", sc["N"], sc["M"] @ R.T) > cos_nl)
    return int(m.sum()), m


def find_opt_cc(Rcw, U, Q, m23):
    Rwc = Rcw.T
    AA, bb = np.zeros((3, 3)), np.zeros(3)
    for i in np.nonzero(m23)[0]:
        v = Rwc @ U[i]
        A = np.eye(3) - np.outer(v, v)
        AA += A
        bb += A @ Q[i]
    if abs(np.linalg.det(AA)) < 1e-4:
        return np.full(3, np.nan)
    return np.linalg.lstsq(AA, bb, rcond=None)[0]


def polar_rotation(M):
    U, _, Vt = np.linalg.svd(M)
    if np.linalg.det(U @ Vt) < 0:
        U = U.copy()
        U[:, 2] *= -1
    return U @ Vt


def nl_shinji_kneip_ls(sc, masks, R0, t0, w, bug):
    """Restated from the algorithm description in SURVEY.md 8a (L1) / Appendix A, incl. the accumulate-across-rounds quirk."""
    Q, P, U, Mw, Nc = sc["Q"], sc["P"], sc["U"], sc["M"], sc["N"]
    m23, m33, mnn = (masks[k] == 1 for k in range(3))
    SH = 32767.0
    w23 = np.ones(len(Q)) if w is None else w[:, 0]
    w33 = np.ones(len(Q)) if w is None else w[:, 1] / SH
    wnn = np.ones(len(Q)) if w is None else w[:, 2] / SH
    N = int(m33.sum())
    TV = float(w33[m33].sum())
    Cw = (w33[m33, None] * Q[m33]).sum(0)
    Cc = (w33[m33, None] * P[m33]).sum(0)
    if N > 2:
        Cw, Cc = Cw / TV, Cc / TV
    M33, MNN, M23 = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 3))
    TL = TW = 0.0
    Mc = Kc = 0
    c_opt = R0.T @ (-t0)
    R_opt = np.eye(3)
    for _ in range(3):
        if not bug:
            M33, MNN, M23 = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 3))
            TL = TW = 0.0
            Mc = Kc = 0
        Aw = Q[m23] - c_opt
        Aw /= np.linalg.norm(Aw, axis=1, keepdims=True)
        M23 = M23 + (w23[m23, None] * U[m23]).T @ Aw
        TW += float(w23[m23].sum())
        Kc += int(m23.sum())
        Ac = P[m33] - Cc
        sigma = float((w33[m33] * (Ac * Ac).sum(1)).sum())
        M33 = M33 + (w33[m33, None] * Ac).T @ (Q[m33] - Cw)
        MNN = MNN + (wnn[mnn, None] * Nc[mnn]).T @ Mw[mnn]
        TL += float(wnn[mnn].sum())
        Mc += int(mnn.sum())
        if N > 2:
            M33, sigma = M33 / TV, sigma / TV
        else:
            M33, sigma = np.zeros((3, 3)), 1.0
        MNN = MNN / TL if Mc > 0 else np.zeros((3, 3))
        M23 = M23 / TW if Kc > 0 else np.zeros((3, 3))
        M33 = M33 + sigma * (M23 + MNN)
        R_opt = polar_rotation(M33)
        c = Cw - R_opt.T @ Cc
        cp = find_opt_cc(R0, U, Q, m23)
        if N > 2:
            c_opt = (Kc / (Kc + N)) * cp + (N / (Kc + N)) * c if not np.isnan(cp[0]) else c
        else:
            if np.isnan(cp[0]):
                break
            c_opt = cp
    return R_opt, R_opt @ (-c_opt)


def calc_err(Rgt, tgt, Rse, tse):
    Rd = Rse @ Rgt.T
    td = tse - Rd @ tgt
    ang = math.atan2(np.linalg.norm([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]]) / 2, (np.trace(Rd) - 1) / 2)
    return float(np.linalg.norm(td)), float(ang)


def quat_wxyz(R):
    """Unit quaternion with w >= 0 via the eigenvector of the K matrix (Bar-Itzhack) -- independent of branchy formulas."""
    K = np.array([[R[0, 0] - R[1, 1] - R[2, 2], R[1, 0] + R[0, 1], R[2, 0] + R[0, 2], R[2, 1] - R[1, 2]],
                  [R[1, 0] + R[0, 1], R[1, 1] - R[0, 0] - R[2, 2], R[2, 1] + R[1, 2], R[0, 2] - R[2, 0]],
                  [R[2, 0] + R[0, 2], R[2, 1] + R[1, 2], R[2, 2] - R[0, 0] - R[1, 1], R[1, 0] - R[0, 1]],
                  [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
    w, v = np.linalg.eigh(K)
    q = v[:, -1]
    q = np.array([q[3], q[0], q[1], q[2]])
    return q if q[0] >= 0 else -q


def main():
    out = {}
    arrays = {}
    # ---- 1. PCG32 reference vector (pcg-random.org check output for seed 42, stream 54)
    g = PCG32(42, 54)
    got = [g.next32() for _ in range(6)]
    assert got == [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293, 0xBFA4784B, 0xCBED606E], [hex(x) for x in got]
    out["pcg32_seed42_seq54"] = got
    g = PCG32(7)
    out["rand31_seed7"] = [g.rand31() for _ in range(16)]
    # ---- 2. samplers
    g = PCG32(3)
    out["random_elements_n100_m4_seed3"] = [random_elements(100, 4, g) for _ in range(20)]
    g = PCG32(3)
    out["random_elements_n5_m3_seed3"] = [random_elements(5, 3, g) for _ in range(10)]
    g = PCG32(9)
    ps = Prosac(4, 100, float)
    out["prosac_f64_m4_n100_seed9"] = [ps.sample(g) for _ in range(300)]
    g = PCG32(9)
    ps = Prosac(3, 50, np.float32)
    out["prosac_f32_m3_n50_seed9"] = [ps.sample(g) for _ in range(300)]
    # ---- 3. adaptive iteration bound
    tbl = []
    for p in (0.99, 0.9999, 0.99999, 0.5):
        for ep in (0.0, 0.01, 0.1, 0.5, 0.9, 0.999, 1.0):
            for mp in (3, 4):
                for mx in (10, 300, 1000, 100000):
                    tbl.append([p, ep, mp, mx, update_num_iters(p, ep, mp, mx)])
    out["update_num_iters_f64"] = tbl
    # ---- 4. scenes
    rng = np.random.default_rng(2024)
    R, t = S.random_pose(rng)
    sc = S.simulate_2d_3d_nl_correspondences(rng, R, t, 1000, 2.0, 0.1, 0.05, 0.1, math.radians(2), 0.1)
    P = sc.P.copy()
    P[rng.permutation(1000)[:50]] = np.nan
    scene = dict(Q=sc.Q, P=P, U=sc.U, M=sc.M, N=sc.N, W=np.asarray(sc.weights))
    arrays.update({"full_" + k: v for k, v in scene.items()})
    out["full_R"], out["full_t"] = R.tolist(), t.tolist()
    # ---- 5. closed form (Kabsch) incl. special cases
    kab = {}
    valid = ~np.isnan(P).all(1)
    for name, (xw, xc) in {
        "noisy_valid": (sc.Q[valid], P[valid]),
        "first3": (sc.Q[valid][:3], P[valid][:3]),
    }.items():
        Rk, tk = kabsch(xw, xc)
        kab[name] = dict(R=Rk.tolist(), t=tk.tolist())
    xw0 = rng.uniform(-3, 3, (200, 3))
    for name, (Rs, ts) in {"pure_translation": (np.eye(3), np.array([0.3, -1.2, 2.0])),
                           "rot180_z": (np.diag([-1.0, -1.0, 1.0]), np.array([1.0, 2.0, 3.0])),
                           "rot180_axis": (2 * np.outer([1, 2, 2], [1, 2, 2]) / 9.0 - np.eye(3), np.array([-1.0, 0.5, 0.25]))}.items():
        xc0 = xw0 @ Rs.T + ts
        arrays["kab_" + name + "_xw"], arrays["kab_" + name + "_xc"] = xw0, xc0
        Rk, tk = kabsch(xw0, xc0)
        assert np.allclose(Rk, Rs, atol=1e-12) and np.allclose(tk, ts, atol=1e-12)
        kab[name] = dict(R=Rs.tolist(), t=ts.tolist())
    # planar world points (rank-2 covariance) and a mirrored cloud (det(UV^T) < 0 branch)
    xwp = np.c_[rng.uniform(-2, 2, (100, 2)), np.zeros(100)]
    Rp, tp = S.random_pose(rng)
    xcp = xwp @ Rp.T + tp
    arrays["kab_planar_xw"], arrays["kab_planar_xc"] = xwp, xcp
    kab["planar"] = dict(R=Rp.tolist(), t=tp.tolist())
    xwm = rng.uniform(-2, 2, (100, 3))
    xcm = (xwm * np.array([1, 1, -1.0])) @ Rp.T + tp + 0.01 * rng.standard_normal((100, 3))
    arrays["kab_mirror_xw"], arrays["kab_mirror_xc"] = xwm, xcm
    Rk, tk = kabsch(xwm, xcm)
    assert np.linalg.det(Rk) > 0
    kab["mirror"] = dict(R=Rk.tolist(), t=tk.tolist())
    out["kabsch"] = kab
    # ---- 6. votes on an explicit hypothesis list (float64; the quaternion is what both sides consume)
    hyps = []
    for h in range(12):
        ang = [0.0, 0.003, 0.02, 0.3][h % 4]
        w = rng.standard_normal(3)
        w *= ang / np.linalg.norm(w)
        dR, _ = se3_exp(np.r_[0, 0, 0, w])
        Rh, th = dR @ R, t + ang * rng.standard_normal(3)
        q = quat_wxyz(Rh)
        hyps.append(np.r_[q, th])
    hyps = np.array(hyps)
    arrays["hyp_q7"] = hyps
    thre_3d, cos_thr, cos_nl = 0.2, math.cos(math.atan(8.0 / 585.0)), math.cos(0.1)
    out["vote_thresholds"] = [thre_3d, cos_thr, cos_nl]
    vt = {}
    for name, bits in {"33": 1, "23": 2, "33_23": 3, "nn_23": 6, "nn_33": 5, "nn_33_23": 7}.items():
        res = []
        for q7 in hyps:
            qw, qx, qy, qz = q7[:4]
            Rq = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                           [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                           [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
            v, m = votes_numpy(bits, Rq, q7[4:], scene, thre_3d, cos_thr, cos_nl)
            res.append(v)
        vt[name] = res
    _, m0 = votes_numpy(7, R, t, scene, thre_3d, cos_thr, cos_nl)
    arrays["mask_truth"] = m0
    out["votes"] = vt
    # ---- 7. Gauss-Newton normal equations by numerical differentiation on SE(3)
    w = np.array([0.01, -0.02, 0.015])
    dR, _ = se3_exp(np.r_[0, 0, 0, w])
    Rg, tg = dR @ R, t + np.array([0.02, -0.01, 0.03])
    out["gn_pose"] = dict(R=Rg.tolist(), t=tg.tolist())
    sub = slice(0, 200)
    vq = valid[sub]
    gn = {}
    for kind, (a, b, c) in {0: (sc.Q, P, None), 1: (sc.Q, P, sc.N), 2: (sc.Q, sc.U, None), 4: (sc.Q, sc.U, None)}.items():
        aa, bb = a[sub], b[sub]
        cc = None if c is None else c[sub]
        if kind not in (2, 4):
            aa, bb = aa[vq], bb[vq]
            cc = None if cc is None else cc[vq]
        H, gvec, cost = normal_eq_numeric(kind, Rg, tg, aa, bb, cc)
        gn[str(kind)] = dict(H=H.tolist(), g=gvec.tolist(), cost=cost, count=len(aa))
    out["gn_numeric_first200"] = gn
    # ---- 8. SE3 exp
    tang = [rng.standard_normal(6) * s for s in (1e-12, 1e-6, 0.1, 1.0, 2.5)]
    out["se3_exp"] = [dict(a=a.tolist(), R=se3_exp(a)[0].tolist(), t=se3_exp(a)[1].tolist()) for a in tang]
    # ---- 9. find_opt_cc and nl_shinji_kneip_ls
    R0, t0 = Rg, tg
    masks = m0.copy()
    out["find_opt_cc"] = find_opt_cc(R0, sc.U, sc.Q, masks[0]).tolist()
    Pz = np.nan_to_num(P)
    scz = dict(scene, P=Pz)
    nl = {}
    for bug in (True, False):
        for weighted in (False, True):
            Rn, tn = nl_shinji_kneip_ls(scz, masks, R0, t0, scene["W"] if weighted else None, bug)
            nl[f"bug{int(bug)}_w{int(weighted)}"] = dict(R=Rn.tolist(), t=tn.tolist())
    out["nl_shinji_kneip_ls"] = nl
    # ---- 10. error metrics
    out["calc_err"] = dict(gt=dict(R=R.tolist(), t=t.tolist()), se=dict(R=Rg.tolist(), t=tg.tolist()), te_re=list(calc_err(R, t, Rg, tg)))
    qa, qb = quat_wxyz(R), quat_wxyz(Rg)
    if np.dot(qa, qb) < 0:
        qb = -qb
    out["calc_percentage_err"] = dict(te=float(np.linalg.norm(R @ t - Rg @ tg) / np.linalg.norm(tg) * 100),
                                      re_sign_aligned=float(np.linalg.norm(qa - qb) * 100))
    # ---- 11. quartics with known real roots
    quartics = []
    for _ in range(6):
        r = np.sort(rng.uniform(-2, 2, 4))
        lead = rng.uniform(0.5, 2.0) * rng.choice([-1, 1])
        quartics.append(dict(coeffs=(lead * np.poly(r)).tolist(), roots=r.tolist()))
    out["quartics"] = quartics
    # ---- 12. noise-free minimal-solver inputs
    sc4 = S.simulate_2d_3d_nl_correspondences(rng, R, t, 40, 0, 0, 0, 0, 0, 0)
    arrays.update(p3p_Q=sc4.Q, p3p_U=sc4.U, p3p_P=sc4.P, p3p_M=sc4.M, p3p_N=sc4.N)

    np.savez_compressed(os.path.join(HERE, "golden_inputs.npz"), **arrays)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", os.path.join(HERE, "golden.json"), os.path.getsize(os.path.join(HERE, "golden.json")), "bytes;",
          os.path.getsize(os.path.join(HERE, "golden_inputs.npz")), "bytes of inputs")


if __name__ == "__main__":
    main()
