"""How often does the EXACT 2D vote fall through its band filter?  (Diagnostic build -DRPE_SCORE_STATS: build.py build_score_stats; run with
RPE_LIBRARY=<that library>.)  The configs[2] scene -- 307 200 correspondences, 2 000 of them with a bearing -- and a scene with a bearing
for every correspondence, 512 hypotheses around the truth; one JSON line per case: wave evaluations of a pair, fall-throughs, share.
usage: RPE_LIBRARY=rgbd_pose_estimation_amd/lib/librgbdpose_hip_scorestats.so python scripts/score_filter_stats.py [--out file]"""
import argparse, ctypes as C, json, math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default=""); a = ap.parse_args()
    from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
    lib = L.lib()
    counters = hasattr(lib, "rpe_debug_read_score_stats")   # (without the diagnostic build: wall times per call only)
    n, H = 307200, 512
    rng = np.random.default_rng(5)
    R, t = S.random_pose(rng)
    sc = S.simulate_3d_3d_correspondences(rng, R, t, n, 0.05, 0.1).astype(np.float32)
    bv = (sc.P / np.linalg.norm(sc.P, axis=1, keepdims=True)).astype(np.float32)
    bv += (15.0 / 585.0) * rng.standard_normal(bv.shape).astype(np.float32) * 0.3      # ~ sigma_2d = 15 px of Parameters.yml, in bearing units
    bv /= np.linalg.norm(bv, axis=1, keepdims=True)
    sparse = bv.copy(); sparse[2000:] = np.nan                                            # configs[2]: bearings for the first 2 000 only
    q = api.pose7_from_Rt(sc.R, sc.t, L.F32)
    poses = q[None, :] + 0.002 * rng.standard_normal((H, 7)); poses[:, :4] /= np.linalg.norm(poses[:, :4], axis=1, keepdims=True)
    poses = np.ascontiguousarray(poses.astype(np.float32).astype(np.float64))
    cthr = math.cos(math.atan(8.0 / 585.0))
    out = open(a.out, "a") if a.out else None
    for name, b in (("configs2_2000_bearings", sparse), ("all_bearings", bv)):
        ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, bv=b)
        import time
        st = (C.c_ulonglong * 4)()
        if counters: lib.rpe_debug_read_score_stats(st)
        for kind, kname in ((L.VOTE_33_23, "33_23"), (L.VOTE_23, "23"), (L.VOTE_33, "33")):
            v = ctx.score(kind, poses, 0.2, cthr, mode=L.SCORE_EXACT)
            if counters: lib.rpe_debug_read_score_stats(st)
            ts = {}
            for mname, mode in (("exact", L.SCORE_EXACT), ("fast", L.SCORE_FAST)):
                for _ in range(5): ctx.score(kind, poses, 0.2, cthr, mode=mode)
                per = []
                for _ in range(30):
                    t0 = time.perf_counter(); ctx.score(kind, poses, 0.2, cthr, mode=mode); per.append(time.perf_counter() - t0)
                ts[mname + "_us_per_call"] = round(sorted(per)[len(per) // 2] * 1e6, 1)
            if counters: lib.rpe_debug_read_score_stats(st)   # (the timed calls counted too: cleared)
            row = dict(scene=name, kind=kname, hypotheses=H, wave_pair_evaluations=int(st[0]), fall_throughs=int(st[1]), share=(st[1] / st[0]) if st[0] else None, deferred_elements=int(st[2]), deferred_share_of_elements=(st[2] / (128.0 * st[0])) if st[0] else None,
                       votes_mean=float(np.mean(v)), votes_sum=int(np.sum(v)), defer=os.environ.get("RPE_SCORE_DEFER", "1"), **ts)
            line = json.dumps(row); print(line, flush=True)
            if out: out.write(line + "\n")
        ctx.close()


if __name__ == "__main__":
    main()
