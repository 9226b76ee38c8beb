"""Event-timed launches of the one-launch reduction kernels behind the reference's own least-squares API -- K1' moments_kernel
(shinji_ls*, ao: /root/reference/pose/AbsoluteOrientation.hpp:56-73), K5 nl_round_kernel (nl_shinji_kneip_ls:
AbsoluteOrientationNormal.hpp:457-505) and K4b mask_kernel (the winner's masks of the vote loops) -- with the normal-equation kinds
beside them for comparison: average / minimum launch time from the dispatch's own begin / end timestamps (rpe_timing_enable),
steady (working set left in the Infinity Cache by the previous launch) and cold (a 480 MB pass of another context before every launch).
One JSON line per (kernel, n, state).   usage: kernel_roofline.py [--sizes 307200,1000000,10000000] [--kernels a,b,...] [--out file]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def scene(n, seed=4):
    from rgbd_pose_estimation_amd import simulator as S
    rng = np.random.default_rng(seed)
    R, t = S.random_pose(rng)
    base = S.simulate_3d_3d_correspondences(rng, R, t, min(n, 250000), 0.02, 0.0).astype(np.float32)
    Q, P = base.Q, base.P
    nc = rng.standard_normal((len(Q), 3)); nc /= np.linalg.norm(nc, axis=1, keepdims=True)
    nw = nc @ R                         # rows: R^T nc
    bv = P / np.linalg.norm(P, axis=1, keepdims=True)
    reps = (n + len(Q) - 1) // len(Q)
    tile = lambda a: np.ascontiguousarray(np.tile(a.astype(np.float32), (reps, 1))[:n])
    masks = [(rng.random(n) < 0.8).astype(np.int16) for _ in range(3)]
    weights = [(0.5 + rng.random(n)).astype(np.float32) for _ in range(3)]
    return dict(R=R, t=t, xw=tile(Q), xc=tile(P), bv=tile(bv), nw=tile(nw), nc=tile(nc), masks=masks, weights=weights)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="307200,1000000,10000000")
    ap.add_argument("--kernels", default="")
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--out", default="")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    from rgbd_pose_estimation_amd import _lib as L, api
    sizes = [int(s) for s in a.sizes.split(",") if s]
    want = set(k for k in a.kernels.split(",") if k)
    out = open(a.out, "a") if a.out else None

    def emit(row):
        if a.tag:
            row["tag"] = a.tag
        line = json.dumps(row)
        print(line, flush=True)
        if out:
            out.write(line + "\n"); out.flush()

    evict_n = 20_000_000
    ev = scene(evict_n, seed=9)
    evictor = api.Context(0).load(L.F32, xw=ev["xw"], xc=ev["xc"])
    del ev
    for n in sizes:
        sc = scene(n)
        pose = api.pose12(sc["R"], sc["t"])
        q7 = api.pose7_from_Rt(sc["R"], sc["t"], L.F32)
        Rwc = sc["R"].T
        c_opt = -(sc["R"].T @ sc["t"])
        Cw, Cc = sc["xw"][:1000].mean(0), sc["xc"][:1000].mean(0)
        cos_thr = float(np.cos(np.arctan(8.0 / 585.0)))
        ctx = api.Context(0).load(L.F32, xw=sc["xw"], xc=sc["xc"], bv=sc["bv"], nw=sc["nw"], nc=sc["nc"])
        for m in range(3):
            ctx.upload_mask(m, sc["masks"][m])
        cases = {
            # name: (callable, algorithmic bytes per correspondence)
            "K1p_moments": (lambda: ctx.p2p_moments(0), 24),
            "K1p_moments_mask": (lambda: ctx.p2p_moments(L.USE_MASK), 26),
            "K5_nl_round": (lambda: ctx.nl_round(c_opt, Cw, Cc, Rwc), 66),
            "K4b_mask_33": (lambda: ctx.inlier_mask(L.VOTE_33, q7, thre_3d=0.2), 26),
            "K4b_mask_33_23": (lambda: ctx.inlier_mask(L.VOTE_33_23, q7, thre_3d=0.2, cos_thr=cos_thr), 40),
            "K4b_mask_nn_33_23": (lambda: ctx.inlier_mask(L.VOTE_NN_33_23, q7, thre_3d=0.2, cos_thr=cos_thr, cos_nl=0.95), 66),
            "K1_p2p": (lambda: ctx.normal_eq(L.RES_P2P, pose), 24),
            "K1_p2p_mask": (lambda: ctx.normal_eq(L.RES_P2P, pose, L.USE_MASK), 26),
            "K2_p2plane": (lambda: ctx.normal_eq(L.RES_P2PLANE, pose), 36),
            "K3_bearing": (lambda: ctx.normal_eq(L.RES_BEARING, pose), 24),
            # the fused joint kernel (rpe_joint.hip): bytes = the arrays its terms read
            "J_p2p_bearing": (lambda: ctx.normal_eq_joint([(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0)], pose), 36),
            "J_p2plane_bearing": (lambda: ctx.normal_eq_joint([(L.RES_P2PLANE, 1.0), (L.RES_BEARING, 1.0)], pose), 48),
            "J_p2p_bearing_normal": (lambda: ctx.normal_eq_joint([(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0), (L.RES_NORMAL, 1.0)], pose), 60),
            "J_p2p_bearing_mask": (lambda: ctx.normal_eq_joint([(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0)], pose, L.USE_MASK), 40),
        }
        # masks written by K4b would change what the masked kernels read: K4b cases run last, and the masks are restored after them
        order = ["K1p_moments", "K1p_moments_mask", "K5_nl_round", "K1_p2p", "K1_p2p_mask", "K2_p2plane", "K3_bearing",
                 "J_p2p_bearing", "J_p2plane_bearing", "J_p2p_bearing_normal", "J_p2p_bearing_mask",
                 "K4b_mask_33", "K4b_mask_33_23", "K4b_mask_nn_33_23"]
        weighted_done = False
        for name in order + ["K5_nl_round_weighted"]:
            if want and name not in want:
                continue
            if name == "K5_nl_round_weighted":
                for m in range(3):
                    ctx.upload_mask(m, sc["masks"][m])
                    ctx.upload_weight(m, sc["weights"][m])
                weighted_done = True
                fn, bpc = (lambda: ctx.nl_round(c_opt, Cw, Cc, Rwc)), 78
            else:
                fn, bpc = cases[name]
            for state in ("steady", "cold"):
                if state == "cold" and bpc * n > 400e6:
                    continue   # the set does not fit the Infinity Cache anyway
                for _ in range(5):
                    fn()
                launches = a.launches if state == "steady" else max(10, a.launches // 2)
                ctx.timing_enable(launches, 1)
                for _ in range(launches):
                    if state == "cold":
                        evictor.p2p_moments()
                    fn()
                cnt, tot_ms, mn_ms = ctx.timing_collect()
                ctx.timing_enable(0, 1)
                avg = tot_ms / max(cnt, 1) * 1e-3
                emit(dict(kernel=name, n=n, state=state, bytes_per_corr=bpc, MB=bpc * n / 1e6, launches=cnt, avg_us=round(avg * 1e6, 3),
                          min_us=round(mn_ms * 1e3, 3), GBs=round(bpc * n / avg / 1e9, 1), frac_of_peak=round(bpc * n / avg / 1e9 / HBM_PEAK_GBS, 4)))
        del weighted_done
        ctx.close()
    evictor.close()


if __name__ == "__main__":
    main()
