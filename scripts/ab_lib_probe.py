"""A/B of two BUILDS of the library on the same box, same process order: event-timed one-launch normal-equation kernels through a
minimal ctypes binding that needs only symbols both builds export.   usage: ab_lib_probe.py <tag>=<lib.so> [<tag>=<lib.so> ...]"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(path, tag):
    sys.path.insert(0, ROOT)
    import bench
    lib = C.CDLL(path)
    lib.rpe_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.rpe_set_problem.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    lib.rpe_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.rpe_normal_eq.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.rpe_timing_enable.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.rpe_timing_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rpe_destroy.argtypes = [C.c_void_p]
    lib.rpe_last_error.restype = C.c_char_p
    for n in (307200, 1_000_000, 10_000_000, 20_000_000):
        R, t, Q, P, Nc = bench.cheap_scene(n)
        bv = np.ascontiguousarray((P / np.linalg.norm(P, axis=1, keepdims=True)).astype(np.float32))
        h = C.c_void_p()
        assert lib.rpe_create(C.byref(h), 0, None) == 0
        assert lib.rpe_set_problem(h, n, 0) == 0
        for slot, a in ((0, Q), (1, P), (2, bv), (4, Nc)):
            assert lib.rpe_upload(h, slot, a.ctypes.data_as(C.c_void_p)) == 0, lib.rpe_last_error()
        pose = np.concatenate([R.reshape(9), t]).astype(np.float64)
        out = np.zeros(32)
        for kind, name, bpc in ((0, "p2p", 24), (1, "p2plane", 36), (2, "bearing", 24)):
            for _ in range(8):
                assert lib.rpe_normal_eq(h, kind, 0, pose.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0, lib.rpe_last_error()
            k = 40
            lib.rpe_timing_enable(h, k, 1)
            for _ in range(k):
                lib.rpe_normal_eq(h, kind, 0, pose.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
            cnt, tot, mn = C.c_int(0), C.c_double(0), C.c_double(0)
            lib.rpe_timing_collect(h, C.byref(cnt), C.byref(tot), C.byref(mn))
            lib.rpe_timing_enable(h, 0, 1)
            avg = tot.value / max(cnt.value, 1) * 1e-3
            print(json.dumps(dict(tag=tag, kernel=name, n=n, avg_us=round(avg * 1e6, 3), min_us=round(mn.value * 1e3, 3), frac_of_peak=round(bpc * n / avg / 8e12, 4))), flush=True)
        lib.rpe_destroy(h)


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3])
    else:
        for rep in range(2):   # alternate the builds twice: box state drifts
            for spec in sys.argv[1:]:
                tag, path = spec.split("=", 1)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", os.path.abspath(path), tag + "#%d" % rep], check=False)
