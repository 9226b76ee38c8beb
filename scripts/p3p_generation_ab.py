"""FAST-mode kneip_ransac / shinji_kneip_ransac on a hard scene (55 % outliers, hundreds of iterations): wall time with the later
batches generated on the device (default) against host generation (RPE_HOST_HYPOTHESES=1).  Development aid."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker():
    import numpy as np
    import util
    from rgbd_pose_estimation_amd import _lib as L, api
    sc = util.scene_full(5, 307200, np.float32, n2d=2.0, n3d=0.02, outliers=0.55, nan_frac=0.0)
    for name, m, keys in (("kneip_ransac", api.M_KNEIP_RANSAC, ("xw", "bv")), ("shinji_kneip_ransac", api.M_SK_RANSAC, ("xw", "xc", "bv"))):
        data = dict(xw=sc.Q, xc=sc.P, bv=sc.U)
        kw = dict(thre_3d=0.1, thre_2d=6.0, iters=2000, confidence=0.9999, seed=3, score_mode=L.SCORE_FAST, **{k: data[k] for k in keys})
        r = api.run(m, L.F32, **kw)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); r = api.run(m, L.F32, **kw); best = min(best, time.perf_counter() - t0)
        print(json.dumps(dict(solver=name, host_generation="RPE_HOST_HYPOTHESES" in os.environ, ms_incl_upload=round(best * 1e3, 3), iters=int(r["iters"]),
                              votes=int(r["max_votes"]), rot_err=float(util.rot_err(r["R"], sc.R)))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker()
    else:
        for extra in ({}, {"RPE_HOST_HYPOTHESES": "1"}):   # the variable's presence (any value) selects host generation
            env = dict(os.environ, RPE_QUIET="1", **extra)
            if not extra:
                env.pop("RPE_HOST_HYPOTHESES", None)
            subprocess.run([sys.executable, os.path.abspath(__file__), "w"], env=env, check=False)
