"""Disassembly helpers for the ISA-level guard (tests/test_isa_guard.py): extract the gfx950 code object of a compiled unit, split it
into kernels, and answer questions about instruction order and loops.  CPU only: llvm-objdump ships with ROCm."""
import os
import re
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_object(obj_path, tmp):
    """Path of the gfx950 code object of a hipcc -c object, extracted into `tmp` (None for a unit without device code).  Through the
    offload bundler, so that compressed fat binaries (--offload-compress, what the build ships) are handled like plain ones."""
    fat, co = os.path.join(tmp, "unit.fatbin"), os.path.join(tmp, "unit.gfx950.co")
    subprocess.run([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat], check=True, capture_output=True)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    subprocess.run([BUNDLER, "--unbundle", "--type=o", "--targets=" + TARGET, "--input=" + fat, "--output=" + co], check=True, capture_output=True)
    return co if os.path.exists(co) and os.path.getsize(co) > 0 else None


class Ins:
    __slots__ = ("addr", "text", "target")

    def __init__(self, addr, text, target):
        self.addr, self.text, self.target = addr, text, target


def disassemble(obj_path):
    """{mangled kernel name: [Ins]} of the gfx950 code object bundled in a hipcc -c object file."""
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(obj_path, tmp)
        assert co is not None, obj_path
        out = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", co], check=True, capture_output=True, text=True).stdout
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


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_resources(obj_path):
    """[{name (demangled), vgpr, agpr, vgpr_spill, sgpr_spill, scratch}] of every kernel in the gfx950 code object of a hipcc -c object
    (the metadata notes of the code object: what the loader goes by).  [] for a unit without device code."""
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(obj_path, tmp)
        if co is None:
            return []
        txt = subprocess.run([READELF, "--notes", co], check=True, capture_output=True, text=True).stdout
    rows = []
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        rows.append(dict(mangled=re.search(r"\.name:\s+(\S+)", blk).group(1), agpr=int(blk.split()[0]), vgpr=g("vgpr_count"), vgpr_spill=g("vgpr_spill_count"),
                         sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    names = demangle([r["mangled"] for r in rows])
    for r in rows:
        r["name"] = re.sub(r"\(.*", "", names[r["mangled"]]).replace("void ", "")
    return rows
