"""Wall time per call of the joint entry points (rpe_normal_eq_joint: one launch, record to the host; rpe_gn_refine_joint /
rpe_gn_refine_device: whole refinements, time per iteration), for A/B between library builds (RPE_LIBRARY=<path> selects the build).
usage: joint_ab.py [--sizes 307200,1000000] [--tag x] [--out file]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="307200,1000000,10000000")
    ap.add_argument("--tag", default=os.environ.get("RPE_LIBRARY", "head"))
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from rgbd_pose_estimation_amd import _lib as L, api
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import kernel_roofline as KR
    out = open(a.out, "a") if a.out else None
    for n in [int(s) for s in a.sizes.split(",") if s]:
        sc = KR.scene(n)
        pose = api.pose12(sc["R"], sc["t"])
        ctx = api.Context(0).load(L.F32, xw=sc["xw"], xc=sc["xc"], bv=sc["bv"], nw=sc["nw"], nc=sc["nc"])
        for m in range(3):
            ctx.upload_mask(m, sc["masks"][m])
        sets = {"p2p+bearing": [(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0)], "p2plane+bearing": [(L.RES_P2PLANE, 1.0), (L.RES_BEARING, 1.0)],
                "p2p+bearing+normal": [(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0), (L.RES_NORMAL, 1.0)]}
        for name, terms in sets.items():
            for flags, fl in ((0, "plain"), (L.USE_MASK, "mask")):
                row = dict(tag=a.tag, n=n, terms=name, flags=fl)
                for _ in range(20): ctx.normal_eq_joint(terms, pose, flags)
                ts = []
                for _ in range(60):
                    t0 = time.perf_counter(); ctx.normal_eq_joint(terms, pose, flags); ts.append(time.perf_counter() - t0)
                ts.sort(); row["one_launch_call_us"] = round(ts[len(ts) // 2] * 1e6, 2)
                K = 500 if n <= 1_000_000 else 60
                for fn, key in ((ctx.gn_refine_joint, "host_loop_us_per_iter"), (ctx.gn_refine_device, "device_loop_us_per_iter")):
                    try:
                        fn(terms, pose, flags, 20, 0.0)
                        best = 1e9
                        for _ in range(3):
                            t0 = time.perf_counter(); _, it, _, _ = fn(terms, pose, flags, K, 0.0); best = min(best, (time.perf_counter() - t0) / max(it, 1))
                        row[key] = round(best * 1e6, 3)
                    except Exception as e:  # noqa: BLE001
                        row[key] = repr(e)
                line = json.dumps(row); print(line, flush=True)
                if out: out.write(line + "\n"); out.flush()
        ctx.close()

if __name__ == "__main__":
    main()
