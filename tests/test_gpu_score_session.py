"""-m gpu: the resident scoring session (K4r, rpe_score_session_begin / _end).  Inside a session the batches of a RANSAC run and the
winner's masks are served by ONE resident launch; votes and masks must be the oracle's -- hence those of the same calls outside a
session -- bit for bit, for every vote kind, both scoring modes and both dtypes, for any batch length up to 128, with NaN-marked
points, and whatever ends the session (its end call, another entry point, a list too long for it, other parameters)."""
import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

pytestmark = pytest.mark.gpu

KINDS = [L.VOTE_33, L.VOTE_23, L.VOTE_23_MATRIX, L.VOTE_33_23, L.VOTE_NN_23, L.VOTE_NN_33, L.VOTE_NN_33_23]


def _okind(oracle, kind):
    return {L.VOTE_33: oracle.V_33, L.VOTE_23: oracle.V_23, L.VOTE_23_MATRIX: oracle.V_23_MATRIX, L.VOTE_33_23: oracle.V_33_23,
            L.VOTE_NN_23: oracle.V_NN_23, L.VOTE_NN_33: oracle.V_NN_33, L.VOTE_NN_33_23: oracle.V_NN_33_23}[kind]


def _scene(n, dt, seed):
    sc = util.scene_full(seed, n, np.float64, n2d=1.5 / 585.0, n3d=0.02, nnl_deg=2.0, outliers=0.3)
    sc.P[5] = np.nan          # an invalid depth pixel
    sc.P[n - 1] = np.nan      # ... in the ragged tail
    sc.U[7] = np.nan
    return sc.astype(dt)


def _poses(oracle, sc, f64, count, seed):
    rng = np.random.default_rng(seed)
    q = np.tile(oracle.pose7_from_Rt(sc.R, sc.t, f64), (count, 1))
    q[1:, :4] += 0.01 * rng.standard_normal((count - 1, 4)) * rng.random((count - 1, 1))
    q[:, :4] /= np.linalg.norm(q[:, :4], axis=1, keepdims=True)
    q[1:, 4:] += 0.05 * rng.standard_normal((count - 1, 3)) * rng.random((count - 1, 1))
    dt = np.float64 if f64 else np.float32
    return np.ascontiguousarray(q.astype(dt).astype(np.float64))


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("mode", [L.SCORE_EXACT, L.SCORE_FAST])
@pytest.mark.parametrize("kind", KINDS)
def test_session_votes_and_masks_are_the_oracles(gpu_ctx_factory, oracle, kind, mode, f64):
    dt = np.float64 if f64 else np.float32
    n = 30001                                             # ragged: the last group is partial
    sc = _scene(n, dt, 5)
    ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    thr = dict(thre_3d=0.05, cos_thr=float(np.cos(np.arctan(4.0 / 585.0))), cos_nl=float(np.cos(np.radians(5.0))))
    prob = oracle.Problem(f64, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    assert ctx.score_session_begin(kind, mode=mode, **thr)
    best = None
    for count in (8, 1, 32, 33, 100, 128):               # one batch, a single hypothesis, a full one, split lists
        q = _poses(oracle, sc, f64, count, 100 + count)
        v = ctx.score(kind, q, mode=mode, **thr)
        if mode == L.SCORE_EXACT:
            vo = oracle.votes(prob, _okind(oracle, kind), q, **thr)
            assert np.array_equal(v, vo), (count, v[:8], vo[:8])
        best = q[int(np.argmax(v))] if best is None else best
    has = {L.MOD_23: kind in (L.VOTE_23, L.VOTE_23_MATRIX, L.VOTE_33_23, L.VOTE_NN_23, L.VOTE_NN_33_23),
           L.MOD_33: kind in (L.VOTE_33, L.VOTE_33_23, L.VOTE_NN_33, L.VOTE_NN_33_23),
           L.MOD_NN: kind in (L.VOTE_NN_23, L.VOTE_NN_33, L.VOTE_NN_33_23)}
    mods = [m for m in (L.MOD_23, L.MOD_33, L.MOD_NN) if has[m]]
    votes_in = ctx.inlier_mask(kind, best, mode=mode, **thr)
    ctx.score_session_end()
    masks_in = {m: ctx.download_mask(m) for m in mods}
    assert votes_in == int(ctx.score(kind, best[None], mode=mode, **thr)[0])     # the total of a lazily written mask = the hypothesis' score
    # ... and of a pose the session has NOT scored (its masks are waited for)
    assert ctx.score_session_begin(kind, mode=mode, **thr)
    other = _poses(oracle, sc, f64, 3, 77)[2]
    votes_other = ctx.inlier_mask(kind, other, mode=mode, **thr)
    masks_other = {m: ctx.download_mask(m) for m in mods}
    assert votes_other == int(ctx.score(kind, other[None], mode=mode, **thr)[0])
    assert votes_other == sum(int(masks_other[m].sum()) for m in mods)
    # the same call outside a session
    votes_out = ctx.inlier_mask(kind, best, mode=mode, **thr)
    masks_out = {m: ctx.download_mask(m) for m in mods}
    assert votes_in == votes_out
    for m in mods:
        assert np.array_equal(masks_in[m], masks_out[m]), m
    if mode == L.SCORE_EXACT:
        vo, mo = oracle.votes(prob, _okind(oracle, kind), best[None], mask_for=0, **thr)
        assert votes_in == int(vo[0])
        for m in mods:
            assert np.array_equal(masks_in[m], mo[m]), m


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("kind", KINDS)
def test_fast_mode_session_equals_the_launch_path(gpu_ctx_factory, oracle, kind, f64):
    dt = np.float64 if f64 else np.float32
    sc = _scene(20000, dt, 6)
    ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    thr = dict(thre_3d=0.05, cos_thr=float(np.cos(np.arctan(4.0 / 585.0))), cos_nl=float(np.cos(np.radians(5.0))))
    q = _poses(oracle, sc, f64, 96, 9)
    plain = ctx.score(kind, q, mode=L.SCORE_FAST, **thr)
    assert ctx.score_session_begin(kind, mode=L.SCORE_FAST, **thr)
    inside = np.concatenate([ctx.score(kind, q[:8], mode=L.SCORE_FAST, **thr), ctx.score(kind, q[8:24], mode=L.SCORE_FAST, **thr),
                             ctx.score(kind, q[24:], mode=L.SCORE_FAST, **thr)])
    ctx.score_session_end()
    assert np.array_equal(plain, inside)


def test_whatever_ends_a_session_leaves_a_working_context(gpu_ctx_factory, oracle):
    sc = _scene(40000, np.float32, 7)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(False, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    thr = dict(thre_3d=0.05, cos_thr=float(np.cos(np.arctan(4.0 / 585.0))), cos_nl=2.0)
    q = _poses(oracle, sc, False, 200, 3)
    vo = oracle.votes(prob, oracle.V_33_23, q, **thr)
    pose = api.pose12(sc.R, sc.t)
    for ender in ("end", "end_twice", "long_list", "other_threshold", "other_kind", "normal_eq", "refine", "upload", "download_mask", "none"):
        assert ctx.score_session_begin(L.VOTE_33_23, **thr)
        assert np.array_equal(ctx.score(L.VOTE_33_23, q[:20], **thr), vo[:20])
        if ender == "end":
            ctx.score_session_end()
        elif ender == "end_twice":
            ctx.score_session_end(); ctx.score_session_end()
        elif ender == "long_list":
            assert np.array_equal(ctx.score(L.VOTE_33_23, q, **thr), vo)                     # 200 > 128: the launch path
        elif ender == "other_threshold":
            v2 = ctx.score(L.VOTE_33_23, q[:8], thre_3d=0.1, cos_thr=thr["cos_thr"], cos_nl=2.0)
            assert np.array_equal(v2, oracle.votes(prob, oracle.V_33_23, q[:8], thre_3d=0.1, cos_thr=thr["cos_thr"], cos_nl=2.0))
        elif ender == "other_kind":
            assert np.array_equal(ctx.score(L.VOTE_33, q[:8], thre_3d=0.05), oracle.votes(prob, oracle.V_33, q[:8], thre_3d=0.05))
        elif ender == "normal_eq":
            assert ctx.normal_eq(L.RES_P2P, pose)[0][28] > 0
        elif ender == "refine":
            ctx.gn_refine([L.RES_P2P], pose, max_iter=5)                                     # a resident loop of the same context
        elif ender == "upload":
            ctx.upload(L.XW, sc.Q)
        elif ender == "download_mask":
            ctx.inlier_mask(L.VOTE_33_23, q[0], **thr)
            m = ctx.download_mask(L.MOD_33)
            _, mo = oracle.votes(prob, oracle.V_33_23, q[:1], mask_for=0, **thr)
            assert np.array_equal(m, mo[L.MOD_33])
        # "none": the next begin (or close) ends it
        assert np.array_equal(ctx.score(L.VOTE_33_23, q[:40], **thr), vo[:40])
    st = ctx.resident_state()
    assert st["enabled"] and st["lost"] == 0


def test_lazily_written_masks_are_in_place_for_every_kind_of_reader(gpu_ctx_factory, oracle):
    """The masks of a hypothesis the session has scored are not waited for (the session's last message); whoever reads them next --
    a download, a masked Gauss-Newton launch or resident loop, a new session -- must find them complete."""
    sc = _scene(307200, np.float32, 9)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(False, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    thr = dict(thre_3d=0.05, cos_thr=float(np.cos(np.arctan(4.0 / 585.0))), cos_nl=2.0)
    q = _poses(oracle, sc, False, 24, 11)
    vo = oracle.votes(prob, oracle.V_33_23, q, **thr)
    pose = api.pose12(sc.R, sc.t)
    expect = {}
    for h in (0, 5, 23):
        _, mo = oracle.votes(prob, oracle.V_33_23, q[h:h + 1], mask_for=0, **thr)
        ctx.inlier_mask(L.VOTE_33_23, q[h], **thr)                                            # one launch, waited for
        expect[h] = (mo, ctx.normal_eq(L.RES_P2P, pose, flags=L.USE_MASK)[0], ctx.gn_refine([L.RES_P2P], pose, flags=L.USE_MASK, max_iter=4)[0])
    for reader in ("download", "normal_eq", "refine", "session", "download") * 3:
        for h in (0, 5, 23):
            assert ctx.score_session_begin(L.VOTE_33_23, **thr)
            assert np.array_equal(ctx.score(L.VOTE_33_23, q[:8], **thr), vo[:8])
            assert np.array_equal(ctx.score(L.VOTE_33_23, q[8:], **thr), vo[8:])
            assert ctx.inlier_mask(L.VOTE_33_23, q[h], **thr) == vo[h]
            mo, ne, refined = expect[h]
            if reader == "download":
                assert np.array_equal(ctx.download_mask(L.MOD_33), mo[L.MOD_33]) and np.array_equal(ctx.download_mask(L.MOD_23), mo[L.MOD_23])
            elif reader == "normal_eq":
                assert np.array_equal(ctx.normal_eq(L.RES_P2P, pose, flags=L.USE_MASK)[0], ne)
            elif reader == "refine":
                assert np.array_equal(ctx.gn_refine([L.RES_P2P], pose, flags=L.USE_MASK, max_iter=4)[0], refined)
            else:
                assert ctx.score_session_begin(L.VOTE_33, thre_3d=0.05)
                ctx.score_session_end()
                assert np.array_equal(ctx.download_mask(L.MOD_23), mo[L.MOD_23])
    st = ctx.resident_state()
    assert st["enabled"] and st["lost"] == 0


def test_two_contexts_of_one_thread_do_not_deadlock(gpu_ctx_factory, oracle):
    sc = _scene(30000, np.float32, 8)
    a = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    b = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    q = _poses(oracle, sc, False, 16, 4)
    assert a.score_session_begin(L.VOTE_33, thre_3d=0.05)
    va = a.score(L.VOTE_33, q, thre_3d=0.05)
    b.gn_refine([L.RES_P2P], api.pose12(sc.R, sc.t), max_iter=3)      # needs the device's resident slot: a's session gives it up
    assert b.score_session_begin(L.VOTE_33, thre_3d=0.05)             # one session per thread
    vb = b.score(L.VOTE_33, q, thre_3d=0.05)
    assert np.array_equal(va, vb) and np.array_equal(va, a.score(L.VOTE_33, q, thre_3d=0.05))
    b.score_session_end()


def test_a_problem_beyond_a_frame_is_refused_and_scored_by_launches(gpu_ctx_factory, oracle):
    n = 3_000_000
    rng = np.random.default_rng(0)
    xw = rng.standard_normal((n, 3)).astype(np.float32)
    ctx = gpu_ctx_factory().load(L.F32, xw=xw, xc=xw + np.float32(0.01))
    assert not ctx.score_session_begin(L.VOTE_33, thre_3d=0.05)
    q = np.array([[1.0, 0, 0, 0, 0.01, 0.01, 0.01]])
    assert ctx.score(L.VOTE_33, q, thre_3d=0.05)[0] == n


def test_a_session_may_be_continued_and_ended_by_another_thread(gpu_ctx_factory, oracle):
    """A context handed from one thread to the next with its session open: the second thread scores through it, ends it (giving the
    device's resident slot back -- which the first thread took), and the first thread's next calls on ANOTHER context do not touch the
    session it no longer holds."""
    import threading
    sc = _scene(30000, np.float32, 10)
    a = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    b = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(False, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    q = _poses(oracle, sc, False, 24, 12)
    vo = oracle.votes(prob, oracle.V_33, q, thre_3d=0.05)
    pose = api.pose12(sc.R, sc.t)
    out = {}

    def opener():
        out["begun"] = a.score_session_begin(L.VOTE_33, thre_3d=0.05)
        out["v0"] = a.score(L.VOTE_33, q[:8], thre_3d=0.05)
    t = threading.Thread(target=opener); t.start(); t.join()
    assert out["begun"] and np.array_equal(out["v0"], vo[:8])
    assert np.array_equal(a.score(L.VOTE_33, q[8:], thre_3d=0.05), vo[8:])           # this thread, inside the other thread's session
    assert a.inlier_mask(L.VOTE_33, q[2], thre_3d=0.05) == vo[2]
    a.gn_refine([L.RES_P2P], pose, max_iter=3)                                       # ends it; takes and returns the resident slot
    b.gn_refine([L.RES_P2P], pose, max_iter=3)                                       # the slot is free for another context

    def closer():
        out["v1"] = a.score(L.VOTE_33, q[:8], thre_3d=0.05)
        a.score_session_end()
        out["refined"] = b.gn_refine([L.RES_P2P], pose, max_iter=3)[1]
    assert a.score_session_begin(L.VOTE_33, thre_3d=0.05)                            # opened here ...
    t = threading.Thread(target=closer); t.start(); t.join(timeout=60)
    assert not t.is_alive() and np.array_equal(out["v1"], vo[:8]) and out["refined"] == 3   # ... ended there
    assert b.score_session_begin(L.VOTE_33, thre_3d=0.05)                            # this thread no longer holds a's
    assert np.array_equal(b.score(L.VOTE_33, q, thre_3d=0.05), vo)
    b.score_session_end()
    assert np.array_equal(a.score(L.VOTE_33, q, thre_3d=0.05), vo)


def test_a_session_whose_grid_has_gone_away_still_gives_the_right_answers(gpu_ctx_factory, oracle):
    """The session's grid waits a bounded time for its next message (2 s; 0.5 s here through the test hook).  A host that comes back
    later finds no grid: a batch is then scored by a launch, and masks that were sent as the session's last message without being
    waited for are written by the one-launch kernel when their record turns out to be missing -- either way the oracle's results."""
    import time
    sc = _scene(30000, np.float32, 13)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(False, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    thr = dict(thre_3d=0.05, cos_thr=float(np.cos(np.arctan(4.0 / 585.0))), cos_nl=2.0)
    q = _poses(oracle, sc, False, 16, 14)
    vo = oracle.votes(prob, oracle.V_33_23, q, **thr)
    ctx.inject_resident_fault(0, 0.5)
    # (1) the next batch comes too late
    assert ctx.score_session_begin(L.VOTE_33_23, **thr)
    assert np.array_equal(ctx.score(L.VOTE_33_23, q[:8], **thr), vo[:8])
    time.sleep(0.9)
    assert np.array_equal(ctx.score(L.VOTE_33_23, q[8:], **thr), vo[8:])
    # (2) the masks come too late: sent as the last message (their total is known), found missing at the next call
    assert ctx.score_session_begin(L.VOTE_33_23, **thr)
    assert np.array_equal(ctx.score(L.VOTE_33_23, q[:8], **thr), vo[:8])
    time.sleep(0.9)
    assert ctx.inlier_mask(L.VOTE_33_23, q[5], **thr) == vo[5]
    _, mo = oracle.votes(prob, oracle.V_33_23, q[5:6], mask_for=0, **thr)
    assert np.array_equal(ctx.download_mask(L.MOD_33), mo[L.MOD_33]) and np.array_equal(ctx.download_mask(L.MOD_23), mo[L.MOD_23])
    ctx.inject_resident_fault(0, 0.0)
    # a caller's pause is not a lost grid: nothing was counted, and sessions go on working
    st = ctx.resident_state()
    assert st["lost"] == 0 and st["enabled"]
    assert ctx.score_session_begin(L.VOTE_33_23, **thr)
    assert np.array_equal(ctx.score(L.VOTE_33_23, q, **thr), vo)
    ctx.score_session_end()


def test_sessions_of_two_threads_on_one_gpu_take_turns(gpu_ctx_factory, oracle):
    """One resident grid per GPU at a time: a second thread's session (or resident loop) waits for the first one's to end.  Two threads,
    a context each, 150 RANSAC-shaped runs each, a Gauss-Newton loop in between: every result the oracle's, nobody starves."""
    import threading
    sc = _scene(30000, np.float32, 15)
    prob = oracle.Problem(False, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    q = _poses(oracle, sc, False, 8, 16)
    vo = oracle.votes(prob, oracle.V_33, q, thre_3d=0.05)
    _, mo = oracle.votes(prob, oracle.V_33, q[:1], mask_for=0, thre_3d=0.05)
    pose = api.pose12(sc.R, sc.t)
    ctxs = [gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N) for _ in range(2)]
    bad = []

    def worker(ctx, k):
        try:
            for i in range(150):
                assert ctx.score_session_begin(L.VOTE_33, thre_3d=0.05)
                v = ctx.score(L.VOTE_33, q, thre_3d=0.05)
                tot = ctx.inlier_mask(L.VOTE_33, q[0], thre_3d=0.05)
                ctx.score_session_end()
                if not np.array_equal(v, vo) or tot != vo[0]:
                    bad.append((k, i, "votes"))
                if i % 10 == k:
                    if not np.array_equal(ctx.download_mask(L.MOD_33), mo[L.MOD_33]):
                        bad.append((k, i, "mask"))
                    ctx.gn_refine([L.RES_P2P], pose, flags=L.USE_MASK, max_iter=3)
        except Exception as e:  # noqa: BLE001
            bad.append((k, repr(e)))
    ts = [threading.Thread(target=worker, args=(c, k)) for k, c in enumerate(ctxs)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts) and not bad, bad[:5]
    for c in ctxs:
        st = c.resident_state()
        assert st["enabled"] and st["lost"] == 0


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("n", [1, 2, 5, 64, 2049, 131073])
def test_sessions_on_tiny_and_ragged_problems(gpu_ctx_factory, oracle, n, f64):
    """one correspondence, a partial group, one workgroup and a bit, ... : the session's grid is one group per thread whatever n is"""
    dt = np.float64 if f64 else np.float32
    sc = util.scene_full(40 + n % 7, max(n, 8), np.float64, n2d=1.5 / 585.0, n3d=0.02, nnl_deg=2.0, outliers=0.3)
    sc.Q, sc.P, sc.U, sc.M, sc.N = (a[:n] for a in (sc.Q, sc.P, sc.U, sc.M, sc.N))
    if n > 4:
        sc.P[n // 2] = np.nan
    sc = sc.astype(dt)
    ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(f64, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    thr = dict(thre_3d=0.05, cos_thr=float(np.cos(np.arctan(4.0 / 585.0))), cos_nl=float(np.cos(np.radians(5.0))))
    q = _poses(oracle, sc, f64, 40, 3)
    for kind in (L.VOTE_33, L.VOTE_NN_33_23):
        vo, mo = oracle.votes(prob, _okind(oracle, kind), q, mask_for=1, **thr)
        assert ctx.score_session_begin(kind, **thr)
        assert np.array_equal(ctx.score(kind, q, **thr), vo)
        assert ctx.inlier_mask(kind, q[1], **thr) == vo[1]
        ctx.score_session_end()
        assert np.array_equal(ctx.download_mask(L.MOD_33), mo[L.MOD_33])
        if kind == L.VOTE_NN_33_23:
            assert np.array_equal(ctx.download_mask(L.MOD_23), mo[L.MOD_23]) and np.array_equal(ctx.download_mask(L.MOD_NN), mo[L.MOD_NN])


def test_nobody_blocks_on_a_session_that_was_left_open(gpu_ctx_factory, oracle):
    """Round-5 advisor finding: a session held the device's resident slot until some call ended it, and every other resident loop or
    session of the process blocked on it without a time-out -- for ever if the owner never came back (an exception between _begin and
    _end), a deadlock if the owner waited for the blocked thread.  Now: a resident LOOP of another context never waits for a session
    (it runs with one launch per iteration, at once); another SESSION waits its turn at most until the holder's grid has left by itself
    and then takes the slot over; the abandoned session's owner finds out at its next call and goes on with ordinary launches."""
    import threading
    import time
    sc = _scene(30000, np.float32, 21)
    a = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    b = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(False, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    q = _poses(oracle, sc, False, 16, 22)
    vo = oracle.votes(prob, oracle.V_33, q, thre_3d=0.05)
    pose = api.pose12(sc.R, sc.t)
    ref = b.gn_refine([L.RES_P2P], pose, max_iter=4, tol=0.0)
    a.inject_resident_fault(0, 0.5)                                  # a's grids give up after 0.5 s without a message
    assert a.score_session_begin(L.VOTE_33, thre_3d=0.05)            # ... and a's owner "forgets" the session
    assert np.array_equal(a.score(L.VOTE_33, q[:8], thre_3d=0.05), vo[:8])
    out = {}

    def other_thread():
        t0 = time.perf_counter()
        out["refined"] = b.gn_refine([L.RES_P2P], pose, max_iter=4, tol=0.0)     # a loop: no resident grid now, but no waiting either
        out["loop_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        out["begun"] = b.score_session_begin(L.VOTE_33, thre_3d=0.05)            # a session: waits until a's grid has left, then takes over
        out["begin_s"] = time.perf_counter() - t0
        out["votes"] = b.score(L.VOTE_33, q, thre_3d=0.05)
        b.score_session_end()
    t = threading.Thread(target=other_thread); t.start(); t.join(timeout=30)
    assert not t.is_alive()
    assert out["loop_s"] < 0.2 and out["refined"][1] == 4 and np.abs(out["refined"][0] - ref[0]).max() < 1e-9
    assert out["begun"] and out["begin_s"] < 2.0 and np.array_equal(out["votes"], vo)
    # the abandoned session's owner comes back: its grid and its slot are gone, the answers are the oracle's all the same
    assert np.array_equal(a.score(L.VOTE_33, q[8:], thre_3d=0.05), vo[8:])
    assert a.inlier_mask(L.VOTE_33, q[3], thre_3d=0.05) == vo[3]
    a.inject_resident_fault(0, 0.0)
    for c in (a, b):
        st = c.resident_state()
        assert st["enabled"] and st["lost"] == 0
    # and the context manager of the Python binding ends a session on the way out of an exception
    with pytest.raises(RuntimeError):
        with a.score_session(L.VOTE_33, thre_3d=0.05) as resident:
            assert resident
            raise RuntimeError("caller's bug")
    t0 = time.perf_counter()
    assert b.score_session_begin(L.VOTE_33, thre_3d=0.05) and time.perf_counter() - t0 < 0.2    # the slot was given back at once
    b.score_session_end()
