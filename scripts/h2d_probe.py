"""How fast do a frame's arrays reach HBM from PAGEABLE host memory (what the reference-compatible entry points are handed)?  hipMemcpy
of pageable memory (what rpe_upload does) against a pinned source and against a chunked copy through pinned staging buffers.
usage: h2d_probe.py"""
import json, time
import numpy as np
import torch

dev = torch.device("cuda:0")
for mb in (3.6864, 7.3728, 18.432):
    n = int(mb * 1e6 / 4)
    src = torch.from_numpy(np.random.default_rng(0).standard_normal(n).astype(np.float32))
    pin = src.pin_memory()
    dst = torch.empty(n, dtype=torch.float32, device=dev)
    stage = [torch.empty(256 * 1024, dtype=torch.float32).pin_memory() for _ in range(4)]   # 4 x 1 MB
    def pageable():
        dst.copy_(src); torch.cuda.synchronize()
    def pinned():
        dst.copy_(pin, non_blocking=True); torch.cuda.synchronize()
    def staged():
        k = 0
        for off in range(0, n, 256 * 1024):
            m = min(256 * 1024, n - off)
            s = stage[k % 4]
            if k >= 4: ev[k % 4].synchronize()
            s[:m].copy_(src[off:off + m])
            dst[off:off + m].copy_(s[:m], non_blocking=True)
            ev[k % 4].record()
            k += 1
        torch.cuda.synchronize()
    ev = [torch.cuda.Event() for _ in range(4)]
    row = {"MB": mb}
    for name, f in (("pageable", pageable), ("pinned", pinned), ("staged_1MB_x4", staged)):
        for _ in range(3): f()
        best = 1e9
        for _ in range(10):
            t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
        row[name + "_us"] = round(best * 1e6, 1); row[name + "_GBs"] = round(mb * 1e6 / best / 1e9, 1)
    print(json.dumps(row), flush=True)
