"""Adapter-level pipelines with HOST buffers in (rpe_run: a fresh adapter per call, as the reference's demos build one per frame): wall
time with and without the mask read-back, against the bytes the call has to upload at the link's rate.   usage: pipeline_breakdown.py [n]"""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RPE_QUIET"] = "1"
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
rng = np.random.default_rng(1)
R, T = S.random_pose(rng)
sc = S.simulate_2d_3d_nl_correspondences(rng, R, T, n, 2.0, 0.1, 0.05, 0.1, math.radians(2), 0.1).astype(np.float32)
data = dict(xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
kw = dict(thre_3d=0.2, thre_2d=8.0, thre_nl=0.1, iters=300, confidence=0.99999, seed=3)

def best(f, reps=5):
    f(); f()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return round(b * 1e6, 1)

for name, m, keys, ls in (("shinji_ransac (3D-3D) + shinji_ls", api.M_SHINJI_RANSAC, ("xw", "xc"), api.LS_SHINJI_INLIERS),
                          ("shinji_kneip_ransac + shinji_ls", api.M_SK_RANSAC, ("xw", "xc", "bv"), api.LS_SHINJI_INLIERS),
                          ("shinji_kneip_ransac + joint GN", api.M_SK_RANSAC, ("xw", "xc", "bv"), api.LS_GN_JOINT),
                          ("nl_shinji_kneip_ransac + nl_shinji_kneip_ls", api.M_NL_SK_RANSAC, ("xw", "xc", "bv", "nw", "nc"), api.LS_NL_BUGCOMPAT)):
    sel = {k: data[k] for k in keys}
    mb = sum(a.nbytes for a in sel.values()) / 1e6
    row = dict(pipeline=name, n=n, upload_MB=round(mb, 2), upload_us_at_47GBs=round(mb * 1e6 / 47e9 * 1e6, 1))
    row["with_masks_us"] = best(lambda: api.run(m, L.F32, ls=ls, **sel, **kw))
    row["without_mask_readback_us"] = best(lambda: api.run(m, L.F32, ls=ls, want_masks=False, **sel, **kw))
    row["ransac_only_without_readback_us"] = best(lambda: api.run(m, L.F32, ls=api.LS_NONE, want_masks=False, **sel, **kw))
    print(json.dumps(row), flush=True)
