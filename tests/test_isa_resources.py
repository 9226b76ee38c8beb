"""CPU: register-allocation gate over EVERY kernel the library ships (round-4 review: launched kernels spilled, nothing noticed).
For each compiled unit the code object's metadata is read (what the loader goes by): no kernel may spill vector registers, and none
may carry more than 16 bytes of scratch -- the call frame of the device-side 6x6 solve, the only out-of-line device function -- except
the minimal-solver generators of rpe_hypotheses.hip, whose scratch is private ARRAYS (polynomial tables indexed at run time), not
spills.  A kernel that cannot meet this is not instantiated; its launcher routes the call elsewhere (rpe_joint.hip joint_resident_fits,
rpe_normal_eq.hip normal_eq_resident_fits) and says so.  Also: the shipped library stays under 10 MB."""
import os

import pytest

import isa_tools as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rgbd_pose_estimation_amd", "lib")
UNITS = ["rpe_normal_eq", "rpe_icp", "rpe_joint", "rpe_score", "rpe_nl", "rpe_frontend", "rpe_hypotheses", "rpe_prosac"]
PRIVATE_ARRAYS = {"rpe_hypotheses": 384}   # bytes of scratch allowed: stack arrays of the P3P / nl_2p generators (no spills either)


def _built():
    if not all(os.path.exists(os.path.join(LIB, u + ".o")) for u in UNITS):
        from rgbd_pose_estimation_amd import build as B
        B.build()


@pytest.mark.parametrize("unit", UNITS)
def test_no_kernel_spills_and_none_carries_scratch(unit):
    _built()
    rows = T.kernel_resources(os.path.join(LIB, unit + ".o"))
    assert rows, unit
    spilling = [(r["name"], r["vgpr_spill"]) for r in rows if r["vgpr_spill"] > 0]
    assert not spilling, spilling[:10]
    limit = PRIVATE_ARRAYS.get(unit, 16)
    scratch = [(r["name"], r["scratch"]) for r in rows if r["scratch"] > limit]
    assert not scratch, scratch[:10]
    assert all(r["vgpr"] + r["agpr"] <= 512 for r in rows)


def test_kernel_counts_and_library_size():
    _built()
    counts = {u: len(T.kernel_resources(os.path.join(LIB, u + ".o"))) for u in UNITS}
    # (round 4: 401 in rpe_normal_eq, 204 in rpe_joint, a 15 MB library)
    assert counts["rpe_normal_eq"] <= 320 and counts["rpe_joint"] <= 160, counts
    so = os.path.join(LIB, "librgbdpose_hip.so")
    assert os.path.getsize(so) < 10 * 1024 * 1024, os.path.getsize(so)


def test_units_without_device_code_have_none():
    _built()
    for u in ("library", "rpe_hostex", "rpe_capi", "rpe_context", "rpe_receive", "rpe_refine", "rpe_session", "rpe_dist", "rpe_frontend_api"):
        assert T.kernel_resources(os.path.join(LIB, u + ".o")) == []


def test_no_host_unit_outgrows_its_job():
    """The C-ABI side is split by job (rpe_host.hpp lists the units); none of them may grow back into a catch-all file."""
    csrc = os.path.join(os.path.dirname(LIB), "csrc")
    for u in ("rpe_capi.hip", "rpe_context.hip", "rpe_receive.hip", "rpe_refine.hip", "rpe_session.hip", "rpe_dist.hip", "rpe_frontend_api.hip", "rpe_host.hpp"):
        assert os.path.getsize(os.path.join(csrc, u)) < 40 * 1024, u
