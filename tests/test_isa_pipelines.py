"""CPU: ISA-level guard of the software pipelines reworked in round 4.  Their speed rests on properties of the generated code that a
compiler change can silently undo (each of these WAS undone by the optimiser at some point while the kernels were written):

  K5  nl_round_full_kernel (rotating one-register-set pipeline): no `s_waitcnt vmcnt(0)` and no wait with fewer than 8 loads still
      outstanding inside the streaming loop (a "load, wait, use" chain -- the round-3 weighted form -- shows as vmcnt(0) after every
      load), no scratch traffic;
  K4b mask_kernel: the next group's loads are issued BEFORE the current group's arithmetic and waited for AFTER the mask stores;
  K4  score_kernel: the hypothesis loop of the 3D fast kind has no predicate materialised as 0 / 1 and compared with zero again
      (v_cndmask 0,1 + v_cmp_ne 0): the compare's lane mask is the ballot.
"""
import os
import re

import pytest

import isa_tools as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rgbd_pose_estimation_amd", "lib")
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


def streaming_loop(body):
    """the innermost loop that holds the most 16-byte global loads"""
    spans = T.loops(body)
    count = lambda s: sum(1 for i in body if s[0] <= i.addr <= s[1] and i.text.startswith("global_load_dwordx4"))
    best = max(count(s) for s in spans)
    assert best >= 3
    a, b = min((s for s in spans if count(s) == best), key=lambda s: s[1] - s[0])
    return [i for i in body if a <= i.addr <= b]


def vmcnt(i):
    m = re.search(r"vmcnt\((\d+)\)", i.text)
    return int(m.group(1)) if m else None


@pytest.mark.parametrize("sig", ["<float, 256, false>", "<float, 256, true>", "<double, 256, false>", "<double, 256, true>"])
def test_k5_rotating_pipeline_keeps_its_loads_in_flight(sig):
    ks = {n: b for n, b in kernels("rpe_nl").items() if "nl_round_full_kernel" + sig in n}
    assert len(ks) == 1, list(kernels("rpe_nl"))
    body = next(iter(ks.values()))
    loop = streaming_loop(body)
    assert not any(i.text.startswith("scratch_") for i in loop)
    loads = [i for i in loop if i.text.startswith("global_load_dword")]
    weighted = "true" in sig
    assert len(loads) >= (21 if weighted else 18), len(loads)          # 5 arrays x 3 + masks (+ weights): every load of a group
    waits = [vmcnt(i) for i in loop if vmcnt(i) is not None]
    assert waits and min(waits) >= 8, waits                            # never drained: 13-21 outstanding at each wait today


@pytest.mark.parametrize("kind", [0, 2, 5])
def test_k4b_mask_kernel_prefetches_before_the_predicates(kind):
    ks = {n: b for n, b in kernels("rpe_score").items() if "mask_kernel<float, %d, true>" % kind in n}
    assert len(ks) == 1
    body = next(iter(ks.values()))
    spans = T.loops(body)
    # the streaming loop may be split into blocks by the 2D filter's wave-uniform branch: take the widest loop with 16-byte loads
    count = lambda s: sum(1 for i in body if s[0] <= i.addr <= s[1] and i.text.startswith("global_load_dwordx4"))
    a, b = max((s for s in spans if count(s) >= 6), key=lambda s: s[1] - s[0])
    loop = [i for i in body if a <= i.addr <= b]
    # the layout may enter the loop anywhere (the 2D filter's branch splits the body): rotate it so that the batch of loads comes first
    is_load = [i.text.startswith("global_load_dwordx4") for i in loop]
    n = len(loop)
    start = next(k for k in range(n) if is_load[k] and not any(is_load[(k - d) % n] for d in range(1, 25)))
    loop = loop[start:] + loop[:start]
    math = ("v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_fma_f32", "v_mul_f32", "v_fmac_f32")
    last_load = max(k for k, i in enumerate(loop) if i.text.startswith("global_load_dwordx4"))
    first_math = next(k for k, i in enumerate(loop) if i.text.startswith(math))
    first_store = next(k for k, i in enumerate(loop) if i.text.startswith("global_store_dword"))
    last_store = max(k for k, i in enumerate(loop) if i.text.startswith("global_store_dword"))
    assert last_load < first_math < first_store, (last_load, first_math, first_store)
    # no wait that needs the new loads before the arithmetic has started: every wait between the loads and the first store leaves at
    # least the batch outstanding; the wait for the next group's vectors stands behind the last mask store of the trip
    nloads = sum(is_load)
    early = [vmcnt(i) for i in loop[last_load:first_store] if vmcnt(i) is not None]
    assert all(w >= nloads for w in early), (early, nloads)
    tail_waits = [vmcnt(i) for i in loop[last_store:] if vmcnt(i) is not None]
    assert tail_waits, "no wait behind the stores: the next group's loads were sunk into the following trip"
    assert all("nt" in i.text.split() for i in loop if i.text.startswith("global_store_dword")), "mask stores are nontemporal"


def test_k4_score_loop_counts_votes_from_the_compare_masks():
    ks = {n: b for n, b in kernels("rpe_score").items() if "score_kernel<float, 0, false>" in n}
    assert len(ks) == 1
    body = next(iter(ks.values()))
    spans = T.loops(body)
    inner = [s for s in spans if not any(t[0] >= s[0] and t[1] <= s[1] and t != s for t in spans)]
    hyp = [[i for i in body if a <= i.addr <= b] for a, b in inner]
    hyp = [l for l in hyp if sum(i.text.startswith("v_pk_fma_f32") for i in l) >= 20]
    assert len(hyp) == 1
    loop = hyp[0]
    assert sum(i.text.startswith("s_bcnt1_i32_b64") for i in loop) >= 4
    assert not any(re.match(r"v_cndmask_b32\S* v\d+, 0, 1,", i.text) for i in loop), "a predicate is materialised as 0 / 1 again"
    # four hypotheses per trip (two scalar loads each) behind at most two waits (one per hypothesis before)
    assert sum(i.text.startswith("s_load_dword") for i in loop) == 8 and sum(i.text.startswith("s_waitcnt lgkmcnt") for i in loop) <= 2
    assert sum(i.text.startswith("s_bcnt1_i32_b64") for i in loop) == 16


@pytest.mark.parametrize("sig,arrays", [("<float, 5, 256, false>", 3), ("<float, 5, 256, true>", 3), ("<float, 13, 256, true>", 5), ("<double, 5, 256, false>", 3)])
def test_joint_rotating_pipeline_keeps_its_loads_in_flight(sig, arrays):
    """rpe_joint.hip, one-launch kernel: one register set, each array's next-group loads issued right behind the term that consumed it.
    Every trip issues the same loads (clamped index, dummy address for absent masks / weights), so the waits count exactly: no
    vmcnt(0) and never fewer than five loads outstanding at a wait inside the streaming loop (round 5, first form: conditional reloads
    made the compiler wait for everything before the 2D-3D term)."""
    ks = {n: b for n, b in kernels("rpe_joint").items() if "normal_eq_joint_kernel" + sig in n}
    assert len(ks) == 1, [n for n in kernels("rpe_joint") if "joint_kernel" in n][:5]
    body = next(iter(ks.values()))
    spans = T.loops(body)
    count = lambda s: sum(1 for i in body if s[0] <= i.addr <= s[1] and i.text.startswith("global_load_dwordx4"))
    best = max(count(s) for s in spans)
    assert best >= 3 * arrays, best
    a, b = max((s for s in spans if count(s) == best), key=lambda s: s[1] - s[0])   # the widest loop with all of a trip's loads
    loop = [i for i in body if a <= i.addr <= b]
    assert not any(i.text.startswith("scratch_") for i in loop)
    waits = [vmcnt(i) for i in loop if vmcnt(i) is not None]
    if arrays == 5:
        # sets with another term beside the normal-normal one ask for the normal arrays right before their term (JointNeeds::LATE_NN:
        # no room to hold them through the heavier terms): that wait drains the queue, and so does the first wait of the next trip;
        # the other waits still leave the prefetched arrays in flight
        assert waits and sum(1 for w in waits if w == 0) <= 2 and max(waits) >= 5, waits
    else:
        assert waits and min(waits) >= 5, waits
