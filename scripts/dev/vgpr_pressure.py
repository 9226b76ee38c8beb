"""Approximate VGPR liveness over a kernel's disassembly (linear backward scan, branches ignored: loops make it an under-estimate at loop
heads, good enough to see WHERE the pressure peaks).  usage: vgpr_pressure.py <object.o> <kernel name substring> [context]"""
import re, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import isa_tools as T

def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None: out.append(int(m.group(3)))
        else: out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out

obj, pat = sys.argv[1], sys.argv[2]
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 6
f = T.disassemble(obj); dm = T.demangle(list(f))
for n, body in f.items():
    if pat not in dm[n]: continue
    live, counts = set(), [0] * len(body)
    for k in range(len(body) - 1, -1, -1):
        t = body[k].text
        ops = t.split(None, 1)
        args = ops[1].split(",") if len(ops) > 1 else []
        op = ops[0]
        store = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "s_", "v_cmp", "v_writelane")) and not op.startswith("v_cmpx")
        ndef = 0 if store else 1
        if op.startswith(("v_permlane32_swap", "v_permlane16_swap", "v_swap")): ndef = 2
        defs = [r for a in args[:ndef] for r in regs(a)]
        uses = [r for a in args[ndef:] for r in regs(a)]
        if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_permlane", "v_mov_b32_dpp", "v_writelane")): uses += defs if not op.startswith("v_writelane") else regs(args[0])
        if op.startswith("v_writelane"): defs = []
        for r in defs: live.discard(r)
        live.update(uses)
        counts[k] = len(live)
    peak = max(counts); kp = counts.index(peak)
    print(dm[n][:110]); print(" instructions", len(body), "peak live", peak, "at", kp)
    step = max(1, len(body) // 60)
    print(" profile:", " ".join(str(counts[i]) for i in range(0, len(body), step)))
    for k in range(max(0, kp - ctx), min(len(body), kp + ctx)): print("  ", k, counts[k], body[k].text)
