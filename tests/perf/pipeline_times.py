"""End-to-end wall time of the reference-shaped pipelines on the GPU backend vs the CPU oracle (numbers quoted in DESIGN.md;
lives under tests/ because it times the oracle, which only tests/, smoke() and bench.py's cpu_baseline may touch).  Includes H2D upload of the arrays (the adapter-level entry points take host matrices, like the reference)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["RPE_QUIET"] = "1"
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
import oracle_lib as O
import math

def t(f, reps=3):
    f(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return best, r

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
rng = np.random.default_rng(1)
R, T = S.random_pose(rng)
sc = S.simulate_2d_3d_nl_correspondences(rng, R, T, n, 2.0, 0.1, 0.05, 0.1, math.radians(2), 0.1).astype(np.float32)
rows = []
g, _ = t(lambda: api.ao(sc.Q, sc.P)); c, _ = t(lambda: O.ao(sc.Q, sc.P), 1); rows.append(("ao (Library.cpp)", g, c))
g, _ = t(lambda: api.ao_ransac(sc.Q, sc.P)); c, _ = t(lambda: O.ao_ransac(sc.Q, sc.P, 1), 1); rows.append(("ao_ransac (Library.cpp)", g, c))
kw = dict(thre_3d=0.2, thre_2d=8.0, thre_nl=0.1, iters=300, confidence=0.99999, seed=3)
for name, m, mo, keys, ls, lso in (
        ("shinji_kneip_ransac + shinji_ls", api.M_SK_RANSAC, O.M_SK_RANSAC, ("xw", "xc", "bv"), api.LS_SHINJI_INLIERS, O.LS_SHINJI_INLIERS),
        ("nl_shinji_kneip_ransac + nl_shinji_kneip_ls", api.M_NL_SK_RANSAC, O.M_NL_SK_RANSAC, ("xw", "xc", "bv", "nw", "nc"), api.LS_NL_BUGCOMPAT, O.LS_NL_BUGCOMPAT)):
    data = dict(xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    sel = {k: data[k] for k in keys}
    g, rg = t(lambda: api.run(m, L.F32, ls=ls, **sel, **kw))
    prob = O.Problem(False, **sel)
    c, rc = t(lambda: O.run(prob, mo, ls=lso, **kw), 1)
    rows.append((name + f" [iters {rg['iters']}/{rc['iters']}, votes {rg['max_votes']}/{rc['max_votes']}]", g, c))
for name, g, c in rows:
    print(json.dumps(dict(pipeline=name, n=n, gpu_ms=round(g * 1e3, 3), cpu_oracle_ms=round(c * 1e3, 3), speedup=round(c / g, 1))))
