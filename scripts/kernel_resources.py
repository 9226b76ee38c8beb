"""VGPR / AGPR / spill / scratch figures of the kernels in a hipcc -c object (CPU only).  usage: kernel_resources.py <object.o> [filter ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import isa_tools as T
obj, filt = sys.argv[1], sys.argv[2:]
for r in T.kernel_resources(obj):
    short = r["name"].replace("rpe::", "")
    if all(f in short for f in filt):
        print(f"{short:75s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} vgpr_spill {r['vgpr_spill']:3d} sgpr_spill {r['sgpr_spill']:3d} scratch {r['scratch']} lds {r['lds']}")
