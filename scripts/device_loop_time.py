"""Time per iteration of the device-resident Gauss-Newton loops (rpe_gn_refine_device: the resident grid sums, solves and updates by
itself; one launch for the whole loop): K iterations with tol = 0, wall time / K, best of `reps`.  One JSON line per case.
usage: device_loop_time.py [--sizes 307200,1000000] [--iters 2000] [--reps 5] [--tag x] [--out file]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1000,307200,1000000")
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tag", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
    out = open(a.out, "a") if a.out else None
    for n in [int(s) for s in a.sizes.split(",") if s]:
        rng = np.random.default_rng(3)
        R, t = S.random_pose(rng)
        base = S.simulate_3d_3d_correspondences(rng, R, t, min(n, 250000), 0.02, 0.0).astype(np.float32)
        reps_n = (n + len(base.Q) - 1) // len(base.Q)
        tile = lambda x: np.ascontiguousarray(np.tile(x.astype(np.float32), (reps_n, 1))[:n])
        xw, xc = tile(base.Q), tile(base.P)
        nrm = rng.standard_normal((n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        bv = xc / np.linalg.norm(xc, axis=1, keepdims=True)
        ctx = api.Context(0).load(L.F32, xw=xw, xc=xc, nc=nrm.astype(np.float32), bv=bv.astype(np.float32))
        ctx.upload_mask(1, (rng.random(n) < 0.87).astype(np.int16))
        w = 0.02 * rng.standard_normal(3)
        th = np.linalg.norm(w); k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R0, t0 = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx) @ R, t + 0.02 * rng.standard_normal(3)
        p0 = api.pose12(R0, t0)
        for name, terms, flags in (("p2p_masked", [(L.RES_P2P, 1.0)], L.USE_MASK), ("p2plane", [(L.RES_P2PLANE, 1.0)], 0),
                                   ("joint_p2p_bearing", [(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0)], 0)):
            try:
                ctx.gn_refine_device(terms, p0, flags, 50, 0.0)
                best = 1e9
                for _ in range(a.reps):
                    t0s = time.perf_counter()
                    pd, itd, _, _ = ctx.gn_refine_device(terms, p0, flags, a.iters, 0.0)
                    best = min(best, (time.perf_counter() - t0s) / max(itd, 1))
                row = dict(case=name, n=n, iterations=int(itd), us_per_iteration=round(best * 1e6, 3), resident=ctx.resident_state())
            except Exception as e:  # noqa: BLE001
                row = dict(case=name, n=n, error=repr(e))
            if a.tag:
                row["tag"] = a.tag
            line = json.dumps(row)
            print(line, flush=True)
            if out:
                out.write(line + "\n"); out.flush()
        ctx.close()


if __name__ == "__main__":
    main()
