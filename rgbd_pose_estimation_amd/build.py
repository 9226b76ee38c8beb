"""Compile the HIP backend for gfx950 in-tree: rgbd_pose_estimation_amd/lib/librgbdpose_hip.so.

hipcc cross-compiles without a GPU.  Kernel units (shared device code in csrc/rpe_reduce.hpp and csrc/rpe_residuals.hpp):
csrc/rpe_normal_eq.hip (K1-K3 Gauss-Newton normal equations + the resident form), csrc/rpe_icp.hip (fused ICP rounds),
csrc/rpe_joint.hip (joint normal equations), csrc/rpe_score.hip (K4 scoring, K4b masks), csrc/rpe_nl.hip (K1' moments, K5, publish
kernels), csrc/rpe_frontend.hip (depth-frame front end), csrc/rpe_hypotheses.hip (batched hypothesis generation),
csrc/rpe_prosac.hip (PROSAC order: top-k select + sort); the host units behind include/rgbd_pose_hip.h Part 2 / 3 (csrc/rpe_host.hpp
lists them: rpe_context.hip, rpe_receive.hip, rpe_capi.hip = the thin C-ABI shim, rpe_refine.hip, rpe_session.hip, rpe_dist.hip,
rpe_frontend_api.hip), csrc/library.cpp (reference-compatible ao / ao_ransac / py2c and the adapter-level pipelines),
csrc/rpe_hostex.cpp (host-side all-reduce between the rank processes of one node).  The units compile in parallel (RPE_BUILD_JOBS, default 6)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "librgbdpose_hip.so")
SOURCES = ["rpe_normal_eq.hip", "rpe_icp.hip", "rpe_joint.hip", "rpe_score.hip", "rpe_nl.hip", "rpe_frontend.hip", "rpe_hypotheses.hip", "rpe_prosac.hip", "rpe_context.hip", "rpe_receive.hip", "rpe_capi.hip", "rpe_refine.hip", "rpe_session.hip", "rpe_dist.hip", "rpe_frontend_api.hip", "library.cpp", "rpe_hostex.cpp"]
ARCH = "gfx950"
LINK_RT = "--rtlib=libgcc"


def _deps():
    out = []
    for root in (CSRC, os.path.join(PKG, "include"), os.path.join(os.path.dirname(PKG), "include")):
        for d, _, fs in os.walk(root):
            out += [os.path.join(d, f) for f in fs if f.endswith((".hip", ".cpp", ".h", ".hpp"))]
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def _run_parallel(cmds, verbose):
    jobs = max(1, int(os.environ.get("RPE_BUILD_JOBS", "6")))
    running = []
    def reap(block):
        for p, c in list(running):
            rc = p.wait() if block else p.poll()
            if rc is None:
                continue
            running.remove((p, c))
            if rc != 0:
                for q, _ in running:
                    q.kill()
                raise subprocess.CalledProcessError(rc, c)
    for cmd in cmds:
        while len(running) >= jobs:
            reap(False)
            if len(running) >= jobs:
                import time
                time.sleep(0.05)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        running.append((subprocess.Popen(cmd), cmd))
    while running:
        reap(True)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP backend cannot be built (there is no CPU fallback)")
    os.makedirs(LIBDIR, exist_ok=True)
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        if src.endswith(".cpp"):
            # pure host code over the C ABI (the drop-in headers, ao / ao_ransac / rpe_run): the host C++ compiler, IEEE arithmetic
            # without contraction -- the minimal solvers in pose/*.hpp evaluate the reference's expressions operation by operation
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-c", os.path.join(CSRC, src), "-o", obj]
        else:
            # --offload-compress: the gfx950 code objects are stored compressed in the fat binary (a third of the size; the runtime
            # unpacks a unit once, when its first kernel is looked up -- rpe_create touches every unit)
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "--offload-compress",
                   "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if not force and os.path.exists(obj) and all(os.path.getmtime(p) <= os.path.getmtime(obj) for p in _deps()):
            objs.append(obj)   # this unit is newer than every source and header: keep it
            continue
        cmds.append(cmd)
        objs.append(obj)
    _run_parallel(cmds, verbose)
    # --rtlib=libgcc: the complex multiply / divide helpers (__divdc3 ...) that std::complex code in the P3P solver calls must be the
    # host toolchain's (libgcc), as in any g++-built caller of the headers; clang's compiler-rt copies round differently
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", LINK_RT, "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


def build_stamps(level: int = 1, verbose: bool = False) -> str:
    """DIAGNOSTIC build (never loaded by the product): the same library with -DRPE_STAMPS=<level>, whose reduction kernels stamp the
    100 MHz clock at their phase boundaries (scripts/tail_timeline.py).  Only rpe_normal_eq.hip is recompiled.  The outputs are
    The stamps libraries are scratch: build them right before the diagnostic run and delete them after (they are never loaded by
    the product and should not ride along with every push to the GPU box)."""
    build()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(LIBDIR, f"librgbdpose_hip_stamps{level}.so")
    obj = os.path.join(LIBDIR, f"rpe_normal_eq_stamps{level}.o")
    if os.path.exists(out) and all(os.path.getmtime(p) <= os.path.getmtime(out) for p in _deps()):
        return out
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "--offload-compress", f"-DRPE_STAMPS={level}",
           "-x", "hip", "-c", os.path.join(CSRC, "rpe_normal_eq.hip"), "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    objs = [obj] + [os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o") for src in SOURCES if src != "rpe_normal_eq.hip"]
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", LINK_RT, "-o", out] + objs)
    return out


def build_score_stats(verbose: bool = False) -> str:
    """DIAGNOSTIC build (never loaded by the product): the library with -DRPE_SCORE_STATS, whose exact 2D vote counts how often a wave
    falls through its band filter (scripts/score_filter_stats.py).  Only rpe_score.hip is recompiled; scratch, like the stamps builds."""
    build()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out, obj = os.path.join(LIBDIR, "librgbdpose_hip_scorestats.so"), os.path.join(LIBDIR, "rpe_score_stats.o")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "--offload-compress", "-DRPE_SCORE_STATS",
           "-x", "hip", "-c", os.path.join(CSRC, "rpe_score.hip"), "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    objs = [obj] + [os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o") for src in SOURCES if src != "rpe_score.hip"]
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", LINK_RT, "-o", out] + objs)
    return out


EXAMPLES = ["simple_main", "test_main", "icp_main", "engine_profile", "gn_refine_main"]


def build_examples(verbose: bool = False) -> list[str]:
    """The reference's two demo drivers rebuilt on the drop-in headers: plain g++, linking only the C ABI."""
    root = os.path.dirname(PKG)
    outs = []
    for ex in EXAMPLES:
        src, out = os.path.join(root, "examples", ex + ".cpp"), os.path.join(root, "examples", ex)
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", "-I", os.path.join(PKG, "include", "pose"), "-I", os.path.join(PKG, "include"),
               src, "-L", LIBDIR, "-lrgbdpose_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../rgbd_pose_estimation_amd/lib", "-o", out]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--examples" in sys.argv:
        print(build_examples(verbose=True))
