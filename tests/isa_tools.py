"""Disassembly helpers for the ISA-level guard (tests/test_isa_guard.py): extract the gfx950 code object of a compiled unit, split it
into kernels, and answer questions about instruction order and loops.  CPU only: llvm-objdump ships with ROCm."""
import os
import re
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


class Ins:
    __slots__ = ("addr", "text", "target")

    def __init__(self, addr, text, target):
        self.addr, self.text, self.target = addr, text, target


def disassemble(obj_path):
    """{mangled kernel name: [Ins]} of the gfx950 code object bundled in a hipcc -c object file."""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj_path))
        with open(obj_path, "rb") as src, open(local, "wb") as dst:
            dst.write(src.read())
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f]
        assert len(co) == 1, co
        out = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    funcs, labels, cur, pending = {}, {}, None, []
    for line in out.splitlines():
        m = re.match(r"^([0-9a-f]{8,16}) <(.+)>:$", line)
        if m:
            addr, name = int(m.group(1), 16), m.group(2)
            if re.fullmatch(r"L\d+", name):
                labels[name] = addr
            else:
                cur = funcs.setdefault(name, [])
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-F]{8,16}):", line)
        if m and cur is not None:
            text = m.group(1).strip()
            t = re.search(r"\b(L\d+)\b", text) if text.startswith(("s_cbranch", "s_branch")) else None
            cur.append(Ins(int(m.group(2), 16), text, t.group(1) if t else None))
    for body in funcs.values():
        for i in body:
            if i.target is not None:
                i.target = labels.get(i.target)
    return funcs


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return dict(zip(names, out))


def loops(body):
    """[(first address, last address)] of every backward branch's span"""
    return [(i.target, i.addr) for i in body if i.target is not None and i.target <= i.addr]


def in_loop(ins, spans):
    return [s for s in spans if s[0] <= ins.addr <= s[1]]


def between(body, a, b):
    """instructions with a.addr < addr < b.addr in address order"""
    return [i for i in body if a.addr < i.addr < b.addr]
