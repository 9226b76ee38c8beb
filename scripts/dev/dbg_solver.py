import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
import util
n=61000
sc = util.scene_full(700 + n % 97, n, np.float32, n2d=2.0, n3d=0.03, outliers=0.1, nan_frac=0.0)
rng = np.random.default_rng(n)
mask = (rng.uniform(size=n) < 0.8).astype(np.int16)
w = rng.uniform(0.2, 2.0, n).astype(np.float32)
p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(2), sc.R, sc.t, 0.02, 0.05))
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, nc=sc.N, nw=sc.M)
ctx.upload_mask(L.MOD_33, mask); ctx.upload_weight(L.MOD_33, w)
for kind in (L.RES_P2P, L.RES_P2PLANE):
  for flags in (0, L.USE_MASK, L.USE_WEIGHT, L.USE_MASK|L.USE_WEIGHT):
    ph, ith, steph, costh = ctx.gn_refine([kind], p0, None, flags, 25, 1e-9)
    pd, itd, stepd, costd = ctx.gn_refine_device([(kind, 1.0)], p0, flags, 25, 1e-9)
    print(kind, flags, ith, itd, np.max(np.abs(pd-ph)), steph, stepd, costh, costd, abs(costd-costh)/abs(costh))
