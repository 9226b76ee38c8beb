"""Instruction mix and memory waits of a kernel's streaming loop (the innermost loop holding the most 16-byte global loads), or of
the whole kernel with --all.  usage: loop_mix.py <object.o> <kernel name substring> [--all]"""
import collections, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import isa_tools as T
obj, pat = sys.argv[1], sys.argv[2]
f = T.disassemble(obj); dm = T.demangle(list(f))
for n, b in f.items():
    if pat not in dm[n]: continue
    body = b
    if "--all" not in sys.argv:
        spans = T.loops(b)
        cnt = lambda s: sum(1 for i in b if s[0] <= i.addr <= s[1] and i.text.startswith(("global_load_dwordx4", "ds_read_b128")))
        if spans and max(cnt(s) for s in spans) > 0:
            best = max(cnt(s) for s in spans)
            a, c = min((s for s in spans if cnt(s) == best), key=lambda s: s[1] - s[0])
            body = [i for i in b if a <= i.addr <= c]
    h = collections.Counter(i.text.split()[0] for i in body)
    dp = sum(v for k, v in h.items() if "f64" in k)
    print(dm[n][:100]); print(" instructions", len(body), "| fp64-type", dp, "| packed", sum(v for k, v in h.items() if k.startswith("v_pk_")), "| v_mov", sum(v for k, v in h.items() if k.startswith(("v_mov", "v_pk_mov"))),
          "| cndmask", sum(v for k, v in h.items() if k.startswith("v_cndmask")), "| s_nop", h.get("s_nop", 0), "| loads", sum(v for k, v in h.items() if k.startswith(("global_load", "ds_read"))))
    print(" waits:", [i.text for i in body if i.text.startswith("s_waitcnt")])
    print(" top:", sorted(h.items(), key=lambda x: -x[1])[:24])
