"""Instruction mix of the streaming loops of a kernel unit's gfx950 code object (CPU only: llvm-objdump on the compiled object):
for every kernel whose demangled name matches the filter, the largest backward-branch span = the streaming loop, split into packed
fp32 / fp64 / conversions / selects / memory / other.   usage: loop_instruction_mix.py <object.o> [name filter ...]"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import isa_tools as I


def classify(t):
    op = t.split()[0]
    if op.startswith("v_pk_"):
        return "packed_f32"
    if op.startswith("v_cvt_"):
        return "cvt"
    if op.endswith("_f64") and op.startswith("v_"):
        return "fp64"
    if op.startswith("v_cndmask"):
        return "select"
    if op.startswith(("global_", "buffer_", "flat_", "ds_", "s_load", "s_waitcnt")):
        return "memory/wait"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_"):
        return "valu_other"
    return "scalar"


def main():
    obj = sys.argv[1]
    filt = sys.argv[2:]
    funcs = I.disassemble(obj)
    names = I.demangle(list(funcs))
    for mangled, body in funcs.items():
        nm = names[mangled]
        if filt and not all(f in nm for f in filt):
            continue
        spans = I.loops(body)
        if not spans:
            continue
        a, b = max(spans, key=lambda s: s[1] - s[0])
        loop = [i for i in body if a <= i.addr <= b]
        c = Counter(classify(i.text) for i in loop)
        valu = sum(v for k, v in c.items() if k not in ("memory/wait", "scalar"))
        print(f"{nm[:150]}\n   loop {len(loop)} instructions, VALU {valu}: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()
