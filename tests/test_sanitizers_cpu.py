"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the host code (GPU ASan / XNACK are not available on
this pool).  Two programs are built with g++ -fsanitize=address,undefined -fno-sanitize-recover=all and run:
  * tests/cpp/sanitize_host.cpp + csrc/library.cpp (product host side: drop-in headers, samplers, minimal solvers, RANSAC engine in
    capture mode, rpe_host_*) + csrc/rpe_hostex.cpp (the host-side exchange, three ranks as threads) + oracle/oracle_capi.cpp (the whole CPU restatement: pipelines, vote loops, replay);
  * tests/cpp/host_logic.cpp (PROSAC order, lazy index lists, sparse Fisher-Yates).
librgbdpose_hip.so is linked for the GPU-facing symbols the headers reference; no GPU call is made."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
# leak checking off: the HIP runtime the library is linked against keeps process-lifetime allocations of its own at load time
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", RPE_QUIET="1")


def _build(tmp_path, name, sources):
    from rgbd_pose_estimation_amd import build
    lib = build.build()
    inc = os.path.join(ROOT, "rgbd_pose_estimation_amd", "include")
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-Wall", "-Wno-unused-function", "-ffp-contract=off"] + SAN + ["-I", os.path.join(inc, "pose"), "-I", inc] + sources + \
          ["-L", os.path.dirname(lib), "-lrgbdpose_hip", "-Wl,-rpath," + os.path.dirname(lib), "-pthread", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_host_side_and_oracle_under_asan_ubsan(tmp_path):
    exe = _build(tmp_path, "sanitize_host", [os.path.join(ROOT, "tests", "cpp", "sanitize_host.cpp"),
                                            os.path.join(ROOT, "rgbd_pose_estimation_amd", "csrc", "library.cpp"),
                                            os.path.join(ROOT, "rgbd_pose_estimation_amd", "csrc", "rpe_hostex.cpp"),
                                            os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "sanitize_host: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_host_logic_under_asan_ubsan(tmp_path):
    exe = _build(tmp_path, "host_logic_san", [os.path.join(ROOT, "tests", "cpp", "host_logic.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "host_logic: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
