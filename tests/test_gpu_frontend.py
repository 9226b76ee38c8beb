"""GPU parity of the front end (csrc/rpe_frontend.hip through Part 3 of the C ABI) against oracle/frontend_oracle.py.
The bar is BIT-EXACT maps, pairs and counts (same fp32 operations in the same order, no FMA contraction); the ICP loop,
whose normal equations are summed in a different order than the fp64 oracle's, agrees to 1e-6 rad."""
import os
import numpy as np
import pytest

from frontend_util import FO, SMALL_CAM, oracle_icp, pose12, rot, two_views
from rgbd_pose_estimation_amd import _lib as L, simulator as S
from util import rot_err

pytestmark = pytest.mark.gpu
FULL_CAM = S.DEFAULT_CAMERA


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def holes(depth, rng, frac=0.03):
    d = depth.copy()
    h, w = d.shape
    idx = rng.integers(0, h * w, int(frac * h * w))
    d.reshape(-1)[idx] = 0
    return d


@pytest.mark.parametrize("cam,u16", [(SMALL_CAM, False), (SMALL_CAM, True), (FULL_CAM, True), ((100.0, 90.0, 18.3, 11.1, 37, 23), False),
                                      ((50.0, 50.0, 1.0, 1.0, 3, 3), False), ((50.0, 50.0, 0.0, 0.0, 1, 1), False)])
def test_frame_maps_bit_exact(gpu_ctx_factory, cam, u16):
    rng = np.random.default_rng(5)
    depth = S.render_depth(rot(0.05, -0.1, 0.02), np.array([0.1, -0.05, 0.2]), cam, noise_sigma=0.004, rng=rng, as_u16=u16)
    depth = holes(depth, rng)
    ctx = gpu_ctx_factory()
    scale = 0.001 if u16 else 1.0
    ctx.frame_set_depth(depth, cam, scale, 0.3, 4.5, 0.08)
    V, N, B = FO.frame_maps(depth, cam, scale, 0.3, 4.5, 0.08)
    assert same(ctx.frame_download(L.MAP_VERTEX), V)
    assert same(ctx.frame_download(L.MAP_BEARING), B)
    assert same(ctx.frame_download(L.MAP_NORMAL), N)
    if cam[4] >= 160:
        assert 0.3 < (~np.isnan(N).any(1)).mean() < 1.0     # the case exercises both valid and invalid normals


def test_nan_and_out_of_range_float_depth(gpu_ctx_factory):
    fx, fy, cx, cy, w, h = SMALL_CAM
    depth = np.full((h, w), 2.0, np.float32)
    depth[5, 5] = np.nan
    depth[6, 9] = np.inf
    depth[7, 13] = -1.0
    depth[40:, :] = 9.0
    ctx = gpu_ctx_factory()
    ctx.frame_set_depth(depth, SMALL_CAM, 1.0, 0.1, 8.0, 0.1)
    V, N, B = FO.frame_maps(depth, SMALL_CAM, 1.0, 0.1, 8.0, 0.1)
    gv = ctx.frame_download(L.MAP_VERTEX)
    assert same(gv, V) and same(ctx.frame_download(L.MAP_NORMAL), N)
    gv = gv.reshape(h, w, 3)
    assert np.isnan(gv[5, 5]).all() and np.isnan(gv[6, 9]).all() and np.isnan(gv[7, 13]).all() and np.isnan(gv[40:]).all()


@pytest.mark.parametrize("cam", [SMALL_CAM, FULL_CAM, (100.0, 90.0, 18.3, 11.1, 37, 23)])
def test_model_from_frame_bit_exact(gpu_ctx_factory, cam):
    (RA, tA, dA), _ = two_views(cam)
    ctx = gpu_ctx_factory()
    ctx.frame_set_depth(dA, cam, 1.0, 0.1, 10.0, 0.1)
    pA = pose12(RA, tA)
    ctx.model_from_frame(pA)
    V, N, _ = FO.frame_maps(dA, cam, 1.0, 0.1, 10.0, 0.1)
    MV, MN = FO.to_world(V, N, pA)
    assert same(ctx.frame_download(L.MAP_MODEL_VERTEX), MV)
    assert same(ctx.frame_download(L.MAP_MODEL_NORMAL), MN)


def load_pair(ctx, cam, noise=0.0, u16=False, motion=(0.02, -0.015, 0.01, 0.03, -0.02, 0.025)):
    (RA, tA, dA), (RB, tB, dB) = two_views(cam, motion, noise=noise, as_u16=u16)
    scale = 0.001 if u16 else 1.0
    pA, pB = pose12(RA, tA), pose12(RB, tB)
    ctx.frame_set_depth(dA, cam, scale, 0.1, 10.0, 0.1)
    ctx.model_from_frame(pA)
    ctx.frame_set_depth(dB, cam, scale, 0.1, 10.0, 0.1)
    VA, NA, _ = FO.frame_maps(dA, cam, scale, 0.1, 10.0, 0.1)
    MV, MN = FO.to_world(VA, NA, pA)
    V, N, B = FO.frame_maps(dB, cam, scale, 0.1, 10.0, 0.1)
    return (V, N, B, MV, MN), pA, pB


@pytest.mark.parametrize("cam,u16", [(SMALL_CAM, False), (FULL_CAM, True), ((100.0, 90.0, 18.3, 11.1, 37, 23), False)])
@pytest.mark.parametrize("gate", [(0.15, 0.8, True), (0.02, 0.95, True), (0.05, 0.9, False), (0.0, 0.5, True)])
@pytest.mark.parametrize("which_pose", ["stale", "true"])
def test_associate_bit_exact(gpu_ctx_factory, cam, u16, gate, which_pose):
    ctx = gpu_ctx_factory()
    (V, N, B, MV, MN), pA, pB = load_pair(ctx, cam, noise=0.003, u16=u16)
    p = pA if which_pose == "stale" else pB
    dist, cs, un = gate
    cnt = ctx.associate(p, dist, cs, un)
    XW, XC, BV, NW, NC, want = FO.associate(V, N, B, MV, MN, cam, p, pA, dist, cs, un)
    assert cnt == want
    assert ctx.n == cam[4] * cam[5]
    for slot, ref in ((L.XW, XW), (L.XC, XC), (L.BV, BV), (L.NW, NW), (L.NC, NC)):
        assert same(ctx.download(slot), ref), slot
    if dist > 0.1 and cam[4] >= 160:
        assert cnt > 0.3 * len(V)


def test_model_upload_equals_model_from_frame(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    (V, N, B, MV, MN), pA, pB = load_pair(ctx, SMALL_CAM)
    want = ctx.associate(pA, 0.1, 0.8)
    arrays = [ctx.download(s) for s in range(5)]
    ctx2 = gpu_ctx_factory()
    (_, _, dA), (_, _, dB) = two_views(SMALL_CAM)
    ctx2.model_upload(MV, MN, SMALL_CAM, pA)
    ctx2.frame_set_depth(dB, SMALL_CAM, 1.0, 0.1, 10.0, 0.1)
    assert ctx2.associate(pA, 0.1, 0.8) == want
    for s in range(5):
        assert same(ctx2.download(s), arrays[s])


def test_solvers_consume_the_associated_arrays(gpu_ctx_factory, oracle):
    """The arrays the association leaves in HBM are ordinary adapter arrays: NaN columns are skipped by every kernel."""
    ctx = gpu_ctx_factory()
    (V, N, B, MV, MN), pA, pB = load_pair(ctx, SMALL_CAM, noise=0.003)
    cnt = ctx.associate(pB, 0.1, 0.9)
    XW, XC, BV, NW, NC, _ = FO.associate(V, N, B, MV, MN, SMALL_CAM, pB, pA, 0.1, 0.9, True)
    for kind, c in ((L.RES_P2P, None), (L.RES_P2PLANE, NC), (L.RES_BEARING, None)):
        got, _ = ctx.normal_eq(kind, pB)
        b = BV if kind == L.RES_BEARING else XC
        ref = oracle.gn_normal_eq(kind, XW, b, c, pose=pB)
        assert got[28] == cnt == ref[28]
        assert np.allclose(got[:28], ref[:28], rtol=2e-5, atol=1e-6 * np.abs(ref[:21]).max())
    m = ctx.p2p_moments(L.SKIP_INVALID)
    assert m[17] == cnt
    votes = ctx.score(L.VOTE_33, np.array([oracle.pose7_from_Rt(pB[:9].reshape(3, 3), pB[9:], False)]), 0.1001, mode=L.SCORE_EXACT)
    assert votes[0] == cnt      # every pair lies within the association gate (0.1 < 0.1001), so every pair votes; unpaired pixels never do


@pytest.mark.parametrize("kind,iters", [(L.RES_P2PLANE, 12), (L.RES_P2P, 6)])
@pytest.mark.parametrize("cam", [SMALL_CAM, FULL_CAM])
def test_icp_host_loop_matches_oracle_loop(gpu_ctx_factory, oracle, kind, iters, cam):
    ctx = gpu_ctx_factory()
    (V, N, B, MV, MN), pA, pB = load_pair(ctx, cam, noise=0.002)
    p, it, step, cost, pairs = ctx.icp(pA, kind, iters, 0.0, 0.15, 0.8)
    po, hist = oracle_icp(oracle, V, N, B, MV, MN, cam, pA, pA, kind, iters, 0.15, 0.8)
    assert it == iters
    assert rot_err(p[:9].reshape(3, 3), po[:9].reshape(3, 3)) < 1e-6 and np.linalg.norm(p[9:] - po[9:]) < 1e-6
    assert abs(pairs - hist[-1][0]) <= 2 + 1e-4 * hist[-1][0]      # a pose that differs in the 8th digit may flip a border pixel
    assert abs(cost - hist[-1][2]) <= 1e-3 * hist[-1][2]
    if kind == L.RES_P2PLANE:
        assert rot_err(p[:9].reshape(3, 3), pB[:9].reshape(3, 3)) < 1e-3 and np.linalg.norm(p[9:] - pB[9:]) < 4e-3


@pytest.mark.parametrize("cam", [SMALL_CAM, FULL_CAM])
def test_icp_device_resident_matches_host_loop(gpu_ctx_factory, cam):
    ctx = gpu_ctx_factory()
    load_pair(ctx, cam, noise=0.002)
    (RA, tA, _), (RB, tB, _) = two_views(cam)
    pA = pose12(RA, tA)
    ph, ith, steph, costh, pairsh = ctx.icp(pA, L.RES_P2PLANE, 15, 1e-7, 0.15, 0.8, device_resident=False)
    pd, itd, stepd, costd, pairsd = ctx.icp(pA, L.RES_P2PLANE, 15, 1e-7, 0.15, 0.8, device_resident=True)
    assert rot_err(ph[:9].reshape(3, 3), pd[:9].reshape(3, 3)) < 1e-6 and np.linalg.norm(ph[9:] - pd[9:]) < 1e-6
    assert abs(ith - itd) <= 1 and abs(pairsh - pairsd) <= 2 + 1e-4 * pairsh
    assert rot_err(pd[:9].reshape(3, 3), RB) < 1e-3 and np.linalg.norm(pd[9:] - tB) < 4e-3
    # the solver slots hold the pairs of the last round: a plain GN refinement on them moves the pose by ~nothing
    p2, it2, step2, _ = ctx.gn_refine([L.RES_P2PLANE], pd, max_iter=1)
    assert step2 < 1e-3


def test_frontend_errors(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    I = pose12(np.eye(3), np.zeros(3))
    with pytest.raises(L.RpeError) as e:
        ctx.associate(I)
    assert e.value.code == L.RPE_ERR_STATE
    d = np.full((120, 160), 2.0, np.float32)
    ctx.frame_set_depth(d, SMALL_CAM)
    with pytest.raises(L.RpeError) as e:
        ctx.associate(I)
    assert e.value.code == L.RPE_ERR_STATE and "model" in str(e.value)
    ctx.model_from_frame(I)
    with pytest.raises(L.RpeError) as e:
        ctx.icp(I, L.RES_P2PLANE, use_normals=False)
    assert e.value.code == L.RPE_ERR_ARG
    with pytest.raises(L.RpeError) as e:
        ctx.icp(I, L.RES_BEARING)
    assert e.value.code == L.RPE_ERR_ARG
    with pytest.raises(ValueError):
        ctx.frame_set_depth(d[:5], SMALL_CAM)
    # a fronto-parallel plane alone does not constrain the pose: the solve must fail loudly, not return garbage
    with pytest.raises(L.RpeError) as e:
        ctx.icp(I, L.RES_P2PLANE, 3)
    assert e.value.code == L.RPE_ERR_DEGENERATE


@pytest.mark.parametrize("cam", [SMALL_CAM, FULL_CAM, (100.0, 90.0, 18.3, 11.1, 37, 23)])
@pytest.mark.parametrize("kind", [L.RES_P2PLANE, L.RES_P2P])
@pytest.mark.parametrize("device_resident", [False, True])
def test_fused_icp_matches_the_two_kernel_icp(gpu_ctx_factory, cam, kind, device_resident):
    """One kernel per round (pair + accumulate) against association kernel + normal-equation kernel: same pairing function and
    per-pixel arithmetic, different summation order => poses agree to rounding (1e-9), pair counts exactly unless a pose that
    differs in the last digits flips a pixel at a gate."""
    ctx = gpu_ctx_factory()
    (V, N, B, MV, MN), pA, pB = load_pair(ctx, cam, noise=0.002)
    start = pA if cam[4] >= 160 else pB      # the 37 x 23 toy camera only supports a refinement from the true pose
    iters = 8
    two = ctx.icp(start, kind, iters, 0.0, 0.15, 0.8, device_resident=device_resident, fused=False)
    one = ctx.icp(start, kind, iters, 0.0, 0.15, 0.8, device_resident=device_resident, fused=True)
    assert rot_err(one[0][:9].reshape(3, 3), two[0][:9].reshape(3, 3)) < 1e-9 and np.linalg.norm(one[0][9:] - two[0][9:]) < 1e-9
    assert one[1] == two[1] and abs(one[4] - two[4]) <= 2 and abs(one[3] - two[3]) <= 1e-6 * abs(two[3])
    # the slots hold the pairs under the returned pose
    XW, XC, BV, NW, NC, cnt = FO.associate(V, N, B, MV, MN, cam, one[0], pA, 0.15, 0.8, True)
    assert same(ctx.download(L.XC), XC) and same(ctx.download(L.XW), XW) and same(ctx.download(L.NC), NC)


def test_resident_icp_streaming_form_and_early_stop(tmp_path):
    """The host-driven fused ICP runs in ONE resident launch.  (a) With a grid too small to keep a group per thread (RPE_MAX_BLOCKS=16)
    the kernel re-reads the frame every iteration: same poses as the two-kernel ICP.  (b) A tolerance that stops the loop early must
    release the waiting grid (STOP hand-over) and leave the context usable."""
    import subprocess
    import sys
    script = tmp_path / "icp_small_grid.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})
sys.path.insert(0, {repr(os.path.dirname(os.path.abspath(__file__)))})
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
from test_gpu_frontend import load_pair, FULL_CAM, rot_err
ctx = api.Context(0)
(V, N, B, MV, MN), pA, pB = load_pair(ctx, FULL_CAM, noise=0.002)
two = ctx.icp(pA, L.RES_P2PLANE, 8, 0.0, 0.15, 0.8, device_resident=False, fused=False)
one = ctx.icp(pA, L.RES_P2PLANE, 8, 0.0, 0.15, 0.8, device_resident=False, fused=True)
assert rot_err(one[0][:9].reshape(3, 3), two[0][:9].reshape(3, 3)) < 1e-9 and np.linalg.norm(one[0][9:] - two[0][9:]) < 1e-9, (one, two)
assert one[1] == two[1] == 8 and abs(one[4] - two[4]) <= 2
early2 = ctx.icp(pA, L.RES_P2PLANE, 50, 1e-5, 0.15, 0.8, device_resident=False, fused=False)
early1 = ctx.icp(pA, L.RES_P2PLANE, 50, 1e-5, 0.15, 0.8, device_resident=False, fused=True)
assert early1[1] == early2[1] < 50 and early1[2] < 1e-5
again = ctx.icp(pA, L.RES_P2PLANE, 8, 0.0, 0.15, 0.8, device_resident=False, fused=True)   # the context still works after the early stop
assert np.array_equal(again[0], one[0])
print("ok")
""")
    for env_extra in ({"RPE_MAX_BLOCKS": "16"}, {}):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_resident_icp_lost_grid_is_finished_round_by_round(gpu_ctx_factory):
    """The host-driven fused ICP in one resident launch, with a workgroup withholding its sums in round 3 (rpe_debug_inject_resident_fault): the
    collecting workgroup tells the host, the grid is released, the remaining rounds run one launch each -- the call succeeds with the
    undisturbed result (poses to rounding: the per-round kernel adds in another order) and the context counts one lost grid."""
    import time
    ctx = gpu_ctx_factory()
    (V, N, B, MV, MN), pA, pB = load_pair(ctx, FULL_CAM, noise=0.002)
    good = ctx.icp(pA, L.RES_P2PLANE, 8, 0.0, 0.15, 0.8, device_resident=False, fused=True)
    lost0 = ctx.resident_state()["lost"]
    ctx.inject_resident_fault(3, 0.0)
    try:
        t0 = time.perf_counter()
        hit = ctx.icp(pA, L.RES_P2PLANE, 8, 0.0, 0.15, 0.8, device_resident=False, fused=True)
        dt = time.perf_counter() - t0
    finally:
        ctx.inject_resident_fault(0, 0.0)
    assert hit[1] == good[1] == 8 and 1.5 < dt < 6.0
    assert rot_err(hit[0][:9].reshape(3, 3), good[0][:9].reshape(3, 3)) < 1e-8 and np.linalg.norm(hit[0][9:] - good[0][9:]) < 1e-8
    assert abs(hit[4] - good[4]) <= 2
    assert ctx.resident_state()["lost"] == lost0 + 1
