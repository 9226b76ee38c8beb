"""Front-end timings at the reference camera (640 x 480, f = 585): wall time per call of each stage and of one ICP round
(host-update and device-resident), for DESIGN.md / profiles.  Run under rocprofv3 --kernel-trace --stats for kernel times."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
from rgbd_pose_estimation_amd.api import pose12

def two_views(cam, noise, as_u16):
    rng = np.random.default_rng(0)
    RA, tA = S._rot_zyx(0.05, -0.1, 0.02), np.array([0.1, -0.05, 0.2])
    dR = S._rot_zyx(0.02, -0.015, 0.01)
    RB, tB = dR @ RA, dR @ tA + np.array([0.03, -0.02, 0.025])
    return ((RA, tA, S.render_depth(RA, tA, cam, noise_sigma=noise, rng=rng, as_u16=as_u16)),
            (RB, tB, S.render_depth(RB, tB, cam, noise_sigma=noise, rng=rng, as_u16=as_u16)))

cam = S.DEFAULT_CAMERA
n = cam[4] * cam[5]
(RA, tA, dA), (RB, tB, dB) = two_views(cam, noise=0.002, as_u16=True)
pA, pB = pose12(RA, tA), pose12(RB, tB)
ctx = api.Context(0)
ctx.frame_set_depth(dA, cam, 0.001, 0.1, 10.0, 0.1)
ctx.model_from_frame(pA)
ctx.frame_set_depth(dB, cam, 0.001, 0.1, 10.0, 0.1)
lib = L.lib()

def timeit(f, K=200, sync=True):
    for _ in range(5): f()
    L.check(lib.rpe_synchronize(ctx._h))
    t0 = time.perf_counter()
    for _ in range(K): f()
    if sync: L.check(lib.rpe_synchronize(ctx._h))
    return (time.perf_counter() - t0) / K * 1e6

rows = []
rows.append(("frame_set_depth (H2D 614 KB + F1 maps)", 2 + 36, timeit(lambda: ctx.frame_set_depth(dB, cam, 0.001, 0.1, 10.0, 0.1))))
rows.append(("model_from_frame (F2)", 24 + 24, timeit(lambda: ctx.model_from_frame(pA))))
ctx.model_from_frame(pA)  # model = view A again (the loop above re-derived it from frame B's maps)
ctx.frame_set_depth(dA, cam, 0.001, 0.1, 10.0, 0.1); ctx.model_from_frame(pA); ctx.frame_set_depth(dB, cam, 0.001, 0.1, 10.0, 0.1)
rows.append(("associate (F3, no count)", 36 + 24 + 60, timeit(lambda: ctx.associate(pA, 0.15, 0.8, True, count=False))))
rows.append(("associate (F3, with count + sync)", 36 + 24 + 60, timeit(lambda: ctx.associate(pA, 0.15, 0.8, True, count=True))))
for name, bpp, us in rows:
    print(json.dumps(dict(stage=name, pixels=n, bytes_per_pixel=bpp, wall_us=us, wall_GBs=bpp * n / us / 1e3)), flush=True)
for dev, fused in ((False, False), (True, False), (False, True), (True, True)):
    K = 20
    ctx.icp(pA, L.RES_P2PLANE, K, 0.0, 0.15, 0.8, device_resident=dev, fused=fused)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        p, it, step, cost, pairs = ctx.icp(pA, L.RES_P2PLANE, K, 0.0, 0.15, 0.8, device_resident=dev, fused=fused)
    dt = (time.perf_counter() - t0) / reps
    Rr = p[:9].reshape(3, 3)
    print(json.dumps(dict(stage="icp point-to-plane, %d rounds, %s, %s" % (K, "device-resident" if dev else "host update", "fused kernel" if fused else "associate + normal_eq kernels"), us_per_round=dt / K * 1e6,
                          pairs=pairs, rot_err_rad=float(np.arccos(min(1.0, (np.trace(Rr @ RB.T) - 1) / 2))), trans_err_m=float(np.linalg.norm(p[9:] - tB)))), flush=True)
