"""Register / scratch / LDS use of every kernel of one unit, from the compiler's own remarks (no GPU needed):
    python scripts/kernel_resources.py rpe_normal_eq.hip [name-filter]
Development aid: a kernel that spills (ScratchSize > 0) or drops below the intended occupancy shows up here before a GPU run."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "rgbd_pose_estimation_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'occ':>4} {'LDS':>6}  kernel")
for b, n in zip(blocks, dem):
    if flt and flt not in n:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print(f"{g('VGPRs'):>5} {g('AGPRs'):>5} {g('SGPRs'):>5} {g('ScratchSize .bytes/lane.'):>8} {g('Occupancy .waves/SIMD.'):>4} {g('LDS Size .bytes/block.'):>6}  {n[:150]}")
