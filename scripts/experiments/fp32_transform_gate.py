"""Round-4 experiment, CPU / numpy only: can the Gauss-Newton kernels form p = R Xw + t and the cancelling
residual in fp32 -- pose carried as a two-float (hi + lo) pair, fp64 kept for the accumulators only -- instead of in fp64?
Gate: converged pose vs the fp64 loop <= 1e-7 rad / 1e-7 relative t at 1 M, and the |delta| < 1e-9 stop still reached at N >= 100 000.
The script runs the same GN loop three ways on the same fp32 inputs: (a) fp64 transform (what the kernels do), (b) fp32 FMA-free chain
with the hi part only, (c) hi + lo (two-float pose: the lo part removes the pose's own rounding, not the roundings INSIDE the chain),
and prints the |delta| sequence of each.   usage: fp32_transform_gate.py [n ...]"""
import json
import sys

import numpy as np
from scipy.spatial.transform import Rotation


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def se3_exp(d):
    v, w = d[:3], d[3:]
    th = np.linalg.norm(w)
    W = hat(w)
    if th < 1e-12:
        return np.eye(3) + W, v
    R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
    V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    return R, V @ v


def transform(R, t, X, mode):
    if mode == "fp64":
        return X.astype(np.float64) @ R.T + t
    Rh, th = R.astype(np.float32), t.astype(np.float32)
    # the kernel's chain: fma(R0, x, fma(R1, y, fma(R2, z, t))) -- numpy has no fma; products of fp32 pairs are formed in fp64 (exact)
    # and every partial sum is rounded to fp32, which is what a chain of fp32 FMAs does
    def chain(Rm, tv):
        acc = tv[None, :].astype(np.float64) + X[:, 2:3].astype(np.float64) * Rm[:, 2][None, :].astype(np.float64)
        acc = acc.astype(np.float32).astype(np.float64) + X[:, 1:2].astype(np.float64) * Rm[:, 1][None, :].astype(np.float64)
        acc = acc.astype(np.float32).astype(np.float64) + X[:, 0:1].astype(np.float64) * Rm[:, 0][None, :].astype(np.float64)
        return acc.astype(np.float32)
    p = chain(Rh, th)
    if mode == "fp32_hi":
        return p, None
    Rl, tl = (R - Rh.astype(np.float64)).astype(np.float32), (t - th.astype(np.float64)).astype(np.float32)
    return p, chain(Rl, tl)


def gn(kind, mode, R, t, Xw, Xc, Nc, iters):
    steps = []
    for _ in range(iters):
        if mode == "fp64":
            p = transform(R, t, Xw, mode)
            rv = p - Xc.astype(np.float64)
        else:
            ph, pl = transform(R, t, Xw, mode)
            rv32 = ph - Xc                      # fp32 subtraction: exact when the two are within a factor of two of each other
            if pl is not None:
                rv32 = rv32 + pl
            rv, p = rv32.astype(np.float64), ph.astype(np.float64)
        n = Nc.astype(np.float64)
        if kind == "p2plane":
            r = np.sum(n * rv, axis=1)
            J = np.hstack([n, np.cross(p, n)])
            H, g = J.T @ J, J.T @ r
        else:   # p2p
            H = np.zeros((6, 6)); g = np.zeros(6)
            S = p.sum(0)
            H[:3, :3] = len(p) * np.eye(3); H[:3, 3:] = -hat(S); H[3:, :3] = hat(S)
            H[3:, 3:] = (p * p).sum() * np.eye(3) - p.T @ p
            g[:3] = rv.sum(0); g[3:] = np.cross(p, rv).sum(0)
        d = -np.linalg.solve(H, g)
        dR, dt = se3_exp(d)
        R, t = dR @ R, dR @ t + dt
        steps.append(float(np.linalg.norm(d)))
    return R, t, steps


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [100_000, 1_000_000]
    for n in sizes:
        rng = np.random.default_rng(n)
        Rt = Rotation.from_rotvec(rng.uniform(-1, 1, 3)).as_matrix()
        tt = rng.uniform(-5, 5, 3)
        Pc = np.c_[rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(0.4, 8, n)]
        Xw = ((Pc - tt) @ Rt + 0.05 * rng.standard_normal((n, 3))).astype(np.float32)
        Xc = Pc.astype(np.float32)
        Nc = rng.standard_normal((n, 3)); Nc = (Nc / np.linalg.norm(Nc, axis=1, keepdims=True)).astype(np.float32)
        R0 = Rotation.from_rotvec(0.01 * rng.standard_normal(3)).as_matrix() @ Rt
        t0 = tt + 0.02 * rng.standard_normal(3)
        for kind in ("p2p", "p2plane"):
            ref = None
            for mode in ("fp64", "fp32_hi", "fp32_hilo"):
                R, t, steps = gn(kind, mode, R0, t0, Xw, Xc, Nc, 12)
                if ref is None:
                    ref = (R, t)
                D = R @ ref[0].T
                ang = float(np.arccos(np.clip((np.trace(D) - 1) / 2, -1, 1)))
                print(json.dumps(dict(n=n, kind=kind, transform=mode, steps=["%.1e" % s for s in steps], floor=float(np.median(steps[-5:])),
                                      stop_1e9_reached=bool(min(steps) < 1e-9), rot_vs_fp64_rad=ang,
                                      trans_rel_vs_fp64=float(np.linalg.norm(t - ref[1]) / np.linalg.norm(ref[1])))))


if __name__ == "__main__":
    main()
