"""VGPR / AGPR / spill / scratch figures of the kernels in a hipcc -c object (CPU only).  usage: kernel_resources.py <object.o> [filter ...]"""
import os, re, subprocess, sys, tempfile
OBJDUMP, READELF = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
obj, filt = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as tmp:
    local = os.path.join(tmp, os.path.basename(obj))
    open(local, "wb").write(open(obj, "rb").read())
    subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, capture_output=True)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    txt = subprocess.run([READELF, "--notes", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
rows = []
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
    rows.append((re.search(r"\.name:\s+(\S+)", blk).group(1), int(blk.split()[0]), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (m, ag, vg, vs, ss, sc), nm in zip(rows, names):
    short = re.sub(r"\(.*", "", nm).replace("void rpe::", "")
    if all(f in short for f in filt):
        print(f"{short:75s} vgpr {vg:3d} agpr {ag:3d} vgpr_spill {vs:3d} sgpr_spill {ss:3d} scratch {sc}")
