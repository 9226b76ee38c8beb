"""CPU: ISA-level guard of the fence-free cross-workgroup hand-off (rpe_reduce.hpp / rpe_residuals.hpp).

Under the HIP memory model alone the hand-off is a data race; what makes it correct is the gfx950 lowering of a handful of accesses
(cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Valid forms").  This test disassembles the gfx950 code objects the build
produced and fails if a compiler change drops any of the properties the protocol rests on:

  A  every poll of a granule is a load with the sc1 bit INSIDE a loop (not hoisted), followed in the loop by s_waitcnt vmcnt(0), the
     tag compare, and the bounded-wait clock read;
  B  granules are stored with ONE 16-byte store carrying sc1 (agent scope, write-through), run records / tagged pairs for the host with
     ONE 16-byte store carrying sc0 sc1 (system scope);
  C  the resident kernels poll the control block with system-scope (sc0 sc1) loads inside a loop that sleeps; the autonomous ones poll
     the run records with sc1 16-byte loads inside a loop;
  D  no 16-byte store of a resident kernel lacks the sc1 bit;
  E  arrival-counter tail: the partial record is stored with sc1, and s_waitcnt vmcnt(0) + s_barrier separate it from the arrival
     (global_atomic_add); the last workgroup re-reads the records with sc1 loads after the arrival;
  F  the sequence word a host spins on is stored with sc0 sc1 and an s_waitcnt vmcnt(0) stands between it and the stores it publishes.
"""
import os
import re

import pytest

import isa_tools as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rgbd_pose_estimation_amd", "lib")
UNITS = ["rpe_normal_eq", "rpe_icp", "rpe_joint", "rpe_score", "rpe_nl"]

_cache = {}


def kernels(unit):
    if unit not in _cache:
        obj = os.path.join(LIB, unit + ".o")
        if not os.path.exists(obj):
            from rgbd_pose_estimation_amd import build as B
            B.build()
        f = T.disassemble(obj)
        dm = T.demangle(list(f))
        _cache[unit] = {dm[n]: body for n, body in f.items() if "kernel" in dm[n] and "(" in dm[n]}
    return _cache[unit]


def has(i, prefix, *bits, without=()):
    return i.text.startswith(prefix) and all(re.search(r"\b%s\b" % b, i.text) for b in bits) and not any(re.search(r"\b%s\b" % b, i.text) for b in without)


@pytest.mark.parametrize("unit", UNITS)
def test_granule_polls_stay_inside_their_loops(unit):
    seen = 0
    for name, body in kernels(unit).items():
        spans = T.loops(body)
        for i in body:
            if not has(i, "buffer_load_dwordx4", "sc1"):
                continue
            seen += 1
            inside = T.in_loop(i, spans)
            assert inside, f"{name}: granule poll at {i.addr:#x} is outside every loop (hoisted?)"
            # the loop may be laid out rotated (address order is not execution order): the wait, the tag compare and the clock read of
            # the bounded wait must all be inside the widest loop around the poll
            lo, hi = max(inside, key=lambda s: s[1] - s[0])
            loop = [j for j in body if lo <= j.addr <= hi]
            assert any(j.text.startswith("s_waitcnt vmcnt(0)") for j in loop), f"{name}: no s_waitcnt vmcnt(0) in the loop of the poll at {i.addr:#x}"
            assert any(j.text.startswith(("v_cmp_ne_u64", "v_cmp_eq_u64")) for j in loop), f"{name}: no tag compare in the poll loop"
            assert any(j.text.startswith("s_memrealtime") for j in loop), f"{name}: the poll loop has no bounded-wait clock read"
    if unit in ("rpe_normal_eq", "rpe_icp", "rpe_joint", "rpe_score"):
        assert seen > 0, f"{unit}: no granule poll found at all"


@pytest.mark.parametrize("unit", UNITS)
def test_granule_and_pair_stores_carry_their_scope_bits(unit):
    for name, body in kernels(unit).items():
        polls = [i for i in body if has(i, "buffer_load_dwordx4", "sc1")]
        if not polls:
            continue
        gran = [i for i in body if has(i, "global_store_dwordx4", "sc1", without=("sc0",))]
        pairs = [i for i in body if has(i, "global_store_dwordx4", "sc0", "sc1")]
        assert gran, f"{name}: polls granules but stores none with sc1"
        # run records stay on the device; the result goes out as 8-byte system-scope stores.  The solving workgroup of the autonomous
        # loops (auto_solver_kernel, round 5) is of that kind too: it polls every worker's granules, hands the poses out as sc1
        # granules and publishes the result itself
        autonomous = any(has(i, "global_load_dwordx4", "sc1") for i in body) or "auto_solver_kernel" in name
        if autonomous:
            assert any(has(i, "global_store_dwordx2", "sc0", "sc1") for i in body), f"{name}: autonomous loop without a system-scope result store"
        else:
            assert pairs, f"{name}: collects runs but sends no sc0 sc1 pair to the host"
        if "resident_kernel" in name:
            for i in body:
                if i.text.startswith("global_store_dwordx4"):
                    assert re.search(r"\bsc1\b", i.text), f"{name}: 16-byte store without sc1 at {i.addr:#x}: {i.text}"


@pytest.mark.parametrize("unit", ["rpe_normal_eq", "rpe_icp", "rpe_joint"])
def test_resident_kernels_poll_with_scope_bits_inside_loops(unit):
    seen_host = seen_auto = 0
    for name, body in kernels(unit).items():
        if "resident_kernel" not in name:
            continue
        spans = T.loops(body)
        ctl = [i for i in body if has(i, "global_load_dwordx2", "sc0", "sc1")]
        auto = [i for i in body if has(i, "global_load_dwordx4", "sc1")]
        assert ctl or auto, f"{name}: neither a control-block poll nor a run-record poll"
        for i in ctl:
            seen_host += 1
            inside = T.in_loop(i, spans)
            assert inside, f"{name}: control-block poll at {i.addr:#x} outside every loop"
            lo, hi = max(inside, key=lambda s: s[1] - s[0])
            assert any(j.text.startswith("s_sleep") for j in body if lo <= j.addr <= hi), f"{name}: pose wait loop without s_sleep"
            assert any(j.text.startswith("s_memrealtime") for j in body if lo <= j.addr <= hi), f"{name}: pose wait loop without a bounded wait"
        for i in auto:
            seen_auto += 1
            assert T.in_loop(i, spans), f"{name}: run-record poll at {i.addr:#x} outside every loop"
    assert seen_host > 0
    if unit != "rpe_joint":
        assert seen_auto > 0


@pytest.mark.parametrize("unit", UNITS)
def test_arrival_counter_tail_orders_record_before_arrival(unit):
    seen = 0
    for name, body in kernels(unit).items():
        arrivals = [i for i in body if has(i, "global_atomic_add", "sc0")]   # returning form: the ticket
        if not arrivals:
            continue
        rec = [i for i in body if has(i, "global_store_dwordx2", "sc1", without=("sc0",))]
        assert rec, f"{name}: arrives at a counter but stores no sc1 partial record"
        for a in arrivals:
            before = [r for r in rec if r.addr < a.addr]
            assert before, f"{name}: arrival at {a.addr:#x} with no partial record stored before it"
            gap = T.between(body, before[-1], a)
            k = [n for n, j in enumerate(gap) if j.text.startswith("s_waitcnt vmcnt(0)")]
            assert k, f"{name}: no s_waitcnt vmcnt(0) between the record store and the arrival at {a.addr:#x}"
            assert any(j.text.startswith("s_barrier") for j in gap[k[0]:]), f"{name}: no s_barrier between the drained record store and the arrival"
            seen += 1
        first = arrivals[0]
        assert any(has(i, "global_load_dwordx2", "sc1") and i.addr > first.addr for i in body), f"{name}: the records are not re-read with sc1 loads"
    assert seen > 0


@pytest.mark.parametrize("unit", UNITS)
def test_sequence_word_is_published_behind_a_drain(unit):
    seen = 0
    for name, body in kernels(unit).items():
        sys_stores = [i for i in body if has(i, "global_store_dwordx2", "sc0", "sc1")]
        flags = [i for i in sys_stores if re.search(r"offset:(256|512)\b", i.text)]   # the word behind the 32- / 64-double record
        for f in flags:
            prev = [p for p in sys_stores if p.addr < f.addr]
            if not prev:
                continue
            gap = T.between(body, prev[-1], f)
            assert any(j.text.startswith("s_waitcnt vmcnt(0)") for j in gap), f"{name}: sequence word at {f.addr:#x} not behind s_waitcnt vmcnt(0)"
            seen += 1
    assert seen > 0
