"""-m gpu: rpe_prosac_order (PROSAC's order prefix on the GPU: radix select of the cut + LDS bitonic sort) against the host order
(rpe_host_sort_indexes = pose/Utility.hpp sortIndexes: weight descending, ties to the lower index).  Index work: must be identical."""
import ctypes as C

import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api

pytestmark = pytest.mark.gpu


def host_order(w):
    wd = np.ascontiguousarray(w, np.float64)
    out = np.zeros(len(wd), np.int32)
    L.lib().rpe_host_sort_indexes(wd.ctypes.data_as(C.c_void_p), len(wd), out.ctypes.data_as(C.c_void_p))
    return out


def device_order(ctx, w, k):
    w = np.ascontiguousarray(w, np.float32)
    out = np.zeros(min(k, len(w)), np.int32)
    rc = L.lib().rpe_prosac_order(ctx._h, w.ctypes.data_as(C.c_void_p), len(w), k, out.ctypes.data_as(C.c_void_p))
    return rc, out


@pytest.mark.parametrize("n", [1000, 65536, 307200, 1000003])
@pytest.mark.parametrize("dist", ["uniform", "inverse_noise", "signed_with_zeros", "few_ties"])
def test_device_prefix_is_the_host_prefix(gpu_ctx_factory, n, dist):
    rng = np.random.default_rng(n + len(dist))
    if dist == "uniform":
        w = rng.random(n)
    elif dist == "inverse_noise":   # the simulator's weights (Simulator.hpp:202): 1 / |noise|, heavy tailed
        w = 1.0 / np.linalg.norm(rng.standard_normal((n, 3)), axis=1)
    elif dist == "signed_with_zeros":
        w = rng.standard_normal(n); w[::7] = 0.0; w[3::11] = -0.0
    else:
        w = np.round(rng.random(n) * 50000) / 50000    # duplicates everywhere, a handful per value
    w = w.astype(np.float32)
    ref = host_order(w)
    ctx = gpu_ctx_factory()
    for k in (1, 10, 1005, 4096):
        rc, got = device_order(ctx, w, k)
        assert rc == L.RPE_OK, L.lib().rpe_last_error()
        assert np.array_equal(got, ref[:min(k, n)]), (n, dist, k)
    rc, got = device_order(ctx, w, 1005)   # scratch state is re-armed between calls
    assert rc == L.RPE_OK and np.array_equal(got, ref[:1005][:len(got)])


def test_heavy_ties_are_reported_not_guessed(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    w = np.ones(307200, np.float32)            # every weight equal: the cut's bin holds everything
    rc, _ = device_order(ctx, w, 1005)
    assert rc == L.RPE_ERR_STATE and "equal weights" in L.lib().rpe_last_error().decode()
    w[:500] = 2.0                               # 500 clear winners, then 306 700 ties for the remaining 505 places
    rc, _ = device_order(ctx, w, 1005)
    assert rc == L.RPE_ERR_STATE
    rc, got = device_order(ctx, w, 400)         # a prefix inside the winners is fine ... as long as the ties fit the sort
    assert rc == L.RPE_OK and np.array_equal(got, np.arange(400))
    rc, _ = device_order(ctx, w, 5000)
    assert rc == L.RPE_ERR_ARG                  # longer than the device prefix: the caller sorts on the host
