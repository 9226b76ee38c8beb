"""Worker of tests/test_host_exchange_cpu.py: one rank of the host-side exchange (no GPU involved)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rgbd_pose_estimation_amd import _lib as L, api  # noqa: E402


def vec(rank, step, n):
    return np.random.default_rng(1000 * step + rank).normal(size=n) * 10.0 ** np.random.default_rng(step).integers(-6, 6)


def main():
    name, world, rank, steps, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    out = {"rank": rank}
    try:
        hx = api.HostExchange(name, world, rank, create=(rank == 0), timeout_s=1.0 if mode == "timeout" else 20.0)
        hx.set_label("gpu0" if mode == "collide" else f"gpu{rank}")
        if mode == "timeout":   # the peer never comes: the call fails after its bounded wait, the caller's buffer is untouched, the handle is spent
            import ctypes as C
            import time
            v = np.arange(1, 9, dtype=np.int32)
            rc = L.lib().rpe_host_exchange_allreduce_i32(hx._h, v.ctypes.data_as(C.c_void_p), len(v))
            out["first_rc"], out["untouched"] = rc, bool(np.array_equal(v, np.arange(1, 9)))
            out["error"] = L.lib().rpe_last_error().decode()
            t0 = time.perf_counter()
            d = np.ones(4)
            rc2 = L.lib().rpe_host_exchange_allreduce_f64(hx._h, d.ctypes.data_as(C.c_void_p), 4)
            out["second_rc"], out["second_error"], out["second_s"] = rc2, L.lib().rpe_last_error().decode(), time.perf_counter() - t0
            hx.close()
            print("RESULT " + json.dumps(out), flush=True)
            return
        bad = 0
        for s in range(1, steps + 1):
            n = 1 + (s * 7) % 64
            got = hx.allreduce_f64(vec(rank, s, n))
            want = np.zeros(n)
            for r in range(world):   # rank order, one add at a time: the exchange's order
                want = want + vec(r, s, n)
            bad += int(not np.array_equal(got, want))
            if s % 3 == 0:
                m = 1 + (s * 131) % 8192
                gi = hx.allreduce_i32(np.full(m, rank + s, np.int32))
                bad += int(not np.array_equal(gi, np.full(m, sum(r + s for r in range(world)), np.int32)))
        out["bad"] = bad
        out["collide"] = hx.labels_collide()
        if rank == 0:
            hx.unlink()
            out["left_in_dev_shm"] = os.path.exists("/dev/shm" + name)
        hx.close()
    except L.RpeError as e:
        out["error"] = str(e)
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
