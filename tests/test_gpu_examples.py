"""-m gpu: the reference's two demo drivers rebuilt on the drop-in C++ headers run end to end on the GPU."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name, *args):
    exe = os.path.join(ROOT, "examples", name)
    if not os.path.exists(exe):
        from rgbd_pose_estimation_amd import build
        build.build_examples()
    env = dict(os.environ, RPE_QUIET="1")
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=env)


def test_simple_main_like_reference_demo():
    r = _run("simple_main", "7")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "%test_3d_3d_2d()" in r.stdout and "sk prosac t_s =[" in r.stdout and "sk ransac r_l =[" in r.stdout
    assert len(re.findall(r"%summary \w+: .* -> ok", r.stdout)) == 3


@pytest.mark.parametrize("args", [(), ("noise_model=Kinect", "noise_2d=2", "test_n=10"), ("total=2000", "test_n=5", "noise_2d=3")])
def test_test_main_all_solvers(args):
    r = _run("test_main", *args)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = dict((m.group(1), (float(m.group(2)), float(m.group(3)))) for m in re.finditer(r"^(\w+)\s+([\d.eE+-]+)\s+([\d.eE+-]+)$", r.stdout, re.M))
    assert set(rows) == {"k", "s", "sk", "nk", "ns", "nsk", "opt", "dw", "gn", "gnf"}


@pytest.mark.parametrize("scale", ["1", "0.3"])
def test_icp_main_dense_frame_registration(scale):
    """Front end + ICP + the reference's RANSAC on device-born pairs, from C++ (pose/DepthFrontEnd.hpp)."""
    r = _run("icp_main", scale)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "%summary icp_main -> ok" in r.stdout
    m = re.search(r"^icp\s+iterations (\d+)\s+pairs (\d+)\s+rot_err ([\d.eE+-]+) rad\s+trans_err ([\d.eE+-]+) m", r.stdout, re.M)
    assert m and int(m.group(2)) > 200000 and float(m.group(3)) < 1e-3 and float(m.group(4)) < 3e-3


def test_device_logic_cpp(tmp_path):
    """Lazy device-resident masks, setInlier column semantics, invalidateDevice, host edits (tests/cpp/device_logic.cpp)."""
    from rgbd_pose_estimation_amd import build
    lib = build.build()
    exe = str(tmp_path / "device_logic")
    inc = os.path.join(ROOT, "rgbd_pose_estimation_amd", "include")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wno-unused-function", "-I", os.path.join(inc, "pose"), "-I", inc,
                           os.path.join(ROOT, "tests", "cpp", "device_logic.cpp"), "-L", os.path.dirname(lib), "-lrgbdpose_hip",
                           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, RPE_QUIET="1"))
    assert r.returncode == 0 and "device_logic: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_gn_refine_main_tunes_its_host_thread():
    """examples/gn_refine_main.cpp: the headline loop from plain C++ over the C ABI, timed as placed and after rpe_tune_host_thread
    (the library-level form of what bench.py used to do for itself): the call reports its trials, pins the thread to the best CPU, and the
    tuned figure is not worse than the untuned one beyond noise."""
    import json
    r = _run("gn_refine_main", "20", "30")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["cpu"] >= 0 and str(j["cpu"]) in j["trials"] and len(j["trials"]) >= 1
    assert j["trials"][str(j["cpu"])] <= min(j["trials"].values()) + 2e-3   # (printed with three decimals)
    assert 1.0 < j["us_per_step_tuned"] < 100.0 and j["us_per_step_tuned"] <= 1.15 * j["us_per_step_untuned"]


def test_rpe_host_cpu_environment(tmp_path):
    """RPE_HOST_CPU=<n> pins the thread that drives the resident loop at its first refinement; =auto lets the library measure."""
    import sys
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, util
from rgbd_pose_estimation_amd import _lib as L, api
allowed = sorted(os.sched_getaffinity(0))
want = allowed[len(allowed) // 2]
os.environ["RPE_HOST_CPU"] = str(want)
sc = util.scene33(3, 50000, np.float32, outliers=0.0)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
p, it, _, _ = ctx.gn_refine([L.RES_P2P], api.pose12(np.eye(3), np.zeros(3)), max_iter=10, tol=1e-9)
assert ctx.resident_state()["host_driven"]
assert sorted(os.sched_getaffinity(0)) == [want], (sorted(os.sched_getaffinity(0)), want)
ctx.close()
os.sched_setaffinity(0, allowed)
os.environ["RPE_HOST_CPU"] = "auto"
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
p2, it2, _, _ = ctx.gn_refine([L.RES_P2P], api.pose12(np.eye(3), np.zeros(3)), max_iter=10, tol=1e-9)
assert len(os.sched_getaffinity(0)) == 1 and it2 == it and np.max(np.abs(p2 - p)) < 1e-12
t = ctx.tune_host_thread(L.RES_P2P, api.pose12(np.eye(3), np.zeros(3)), steps=50, reps=3)
assert t["cpu"] in t["trials"] and t["us_per_step"] == min(t["trials"].values()) and sorted(os.sched_getaffinity(0)) == [t["cpu"]]
print("ok")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
