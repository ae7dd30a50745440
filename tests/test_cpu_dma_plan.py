"""Host-side check of the hand-counted `s_waitcnt vmcnt(N)` of the LDS-fed prompt-step GEMMs (csrc/qmm_prefill.inc: qpg_gemm_lds2_kernel,
qpg_gemm_q6k_lds_kernel).  Their loops issue ONLY LDS-DMA instructions, in a fixed order per wave, and wait by count in front of every
chunk's barrier; the counts are exported by the library (`mi355_internal_qpg_dma_plan`: the constants the kernels are compiled with) and
checked here against a model of the instruction stream:
  * loads retire in order, `vmcnt(N)` retires everything but the N instructions issued last;
  * in front of the barrier of chunk (kb, c) this wave's share of image chunk 4 kb + c -- and, for c = 0, of the weights of k-block kb --
    must have retired (SAFE), and the count must not be smaller than that needs in the steady state (TIGHT: a smaller count would throw
    away prefetch depth, a larger one would read LDS before the DMA has landed);
  * a DMA never targets a ring slot / weight stage that a wave may still be reading."""
import ctypes

import pytest


def _plan(lib, kind):
    f = lib.mi355_internal_qpg_dma_plan                               # internal symbol (not in the public header): typed here
    f.restype = ctypes.c_int32
    f.argtypes = [ctypes.c_int32, ctypes.c_void_p]
    out = (ctypes.c_int32 * 6)()
    assert f(kind, ctypes.addressof(out)) == 0
    return list(out)


def _simulate(plan, nkb):
    lead, ring, xu, nw, n0, n1 = plan
    nchunk = 4 * nkb
    issued = []                                                       # the wave's VMEM instructions in program order: ("I", chunk, slot) / ("W", k-block, stage)
    retired = 0                                                       # instructions [0, retired) have landed

    def issue_image(kc, slot):
        for _ in range(xu):
            issued.append(("I", min(kc, nchunk - 1), slot, kc))

    def issue_weights(kb, stage):
        for _ in range(nw):
            issued.append(("W", min(kb, nkb - 1), stage, kb))

    def wait(n):
        nonlocal retired
        retired = max(retired, len(issued) - n)

    def landed(kind, idx):                                            # every instruction of (kind, unclamped index) has retired
        pos = [i for i, op in enumerate(issued) if op[0] == kind and op[3] == idx]
        assert pos, (kind, idx)
        return max(pos) < retired

    slack = []                                                        # how many MORE instructions could have stayed in flight at each wait
    issue_weights(0, 0)
    for c in range(lead):
        issue_image(c, c % ring)
    for kb in range(nkb):
        for c in range(4):
            kc = 4 * kb + c
            n = n0 if c == 0 else n1
            wait(n)
            assert landed("I", kc), (plan, nkb, kb, c)
            if c == 0:
                assert landed("W", kb), (plan, nkb, kb, c)
            # tightness: the newest instruction that HAD to retire
            need = max(i for i, op in enumerate(issued) if (op[0] == "I" and op[3] == kc) or (c == 0 and op[0] == "W" and op[3] == kb))
            slack.append((len(issued) - 1 - need) - n)
            # ---- behind the barrier: every wave has finished chunk kc - 1; some may already read chunk kc (and this k-block's weights)
            slot = (kc + lead) % ring
            assert slot != kc % ring                                  # the refill never lands in the slot being read ...
            in_flight_slots = {(kc + d) % ring for d in range(1, lead)}
            assert slot not in in_flight_slots                        # ... nor in one whose data has not been consumed yet
            issue_image(kc + lead, slot)
            if c == 0:
                assert (kb + 1) & 1 != kb & 1                         # weight stage of k-block kb + 1 is not the one being read
                issue_weights(kb + 1, (kb + 1) & 1)
    return slack


@pytest.mark.parametrize("kind", [6, 61])                            # 61: the Q6_K kernel with one row tile per wave (5 weight DMAs)
def test_counted_waits_are_safe_and_tight(lib, kind):
    plan = _plan(lib, kind)
    lead, ring, xu, nw, n0, n1 = plan
    assert 1 <= lead < ring and n1 < 64                               # vmcnt is a six-bit field
    for nkb in (1, 2, 3, 4, 7, 16, 56):
        slack = _simulate(plan, nkb)
        assert min(slack) >= 0
        assert min(slack) == 0, (plan, nkb, min(slack))               # exact somewhere in every run ...
    steady = _simulate(plan, 16)[8:]
    assert set(steady) == {0}, (plan, sorted(set(steady)))            # ... and everywhere in the steady state


def test_plan_matches_the_documented_counts(lib):
    assert _plan(lib, 6) == [3, 4, 1, 8, 2, 10]                       # Q6_K: ring of 4 chunks, 3 ahead, eight weight DMAs per wave and k-block
    assert _plan(lib, 2) == [3, 4, 1, 6, 1, 8]                        # Q4_K, two workgroups per CU (its own order: the test below)


def _simulate_two_wg(plan, nkb):
    """qpg_gemm_lds2_kernel: per wave and k-block  I P1 P1 | I | I H S P0 P0 | I  behind the barriers of chunks 0..3 (image 3 chunks ahead,
    plane 1 of THIS k-block in chunk 0, headers / sums / plane 0 of the NEXT one in chunk 2); prologue  I(0) | I(1) H S P0 P0 | I(2)"""
    lead, ring, xu, nw, ne, no = plan
    assert (lead, ring, xu, nw) == (3, 4, 1, 6)
    nchunk = 4 * nkb
    issued, retired = [], 0

    def put(kind, idx):
        issued.append((kind, idx))

    def wait(n):
        nonlocal retired
        retired = max(retired, len(issued) - n)

    def landed(kind, idx):
        pos = [i for i, op in enumerate(issued) if op == (kind, idx)]
        assert pos, (kind, idx)
        return max(pos) < retired

    put("I", 0); put("I", 1); put("H", 0); put("S", 0); put("P0", 0); put("P0", 0); put("I", 2)
    slack = []
    for kb in range(nkb):
        for c in range(4):
            kc = 4 * kb + c
            wait(ne if c % 2 == 0 else no)
            need = [("I", kc)]
            if c == 0:
                need += [("H", kb), ("S", kb), ("P0", kb)]
            if c == 2:
                need += [("P1", kb)]
            for kind, idx in need:
                assert landed(kind, idx), (nkb, kb, c, kind)
            newest = max(i for i, op in enumerate(issued) if op in need)
            slack.append((len(issued) - 1 - newest) - (ne if c % 2 == 0 else no))
            assert (kc + lead) % ring != kc % ring and (kc + lead) % ring not in {(kc + d) % ring for d in range(1, lead)}
            put("I", kc + lead)                                       # (past the end the kernel re-reads the last chunk into the same free slot)
            if c == 0:
                put("P1", kb); put("P1", kb)                          # plane-1 region: last read (into registers) in chunk 2 of kb - 1
            if c == 2:
                put("H", kb + 1); put("S", kb + 1); put("P0", kb + 1); put("P0", kb + 1)   # plane-0 region: last read in chunk 0 of kb
    return slack


def test_two_workgroup_kernel_counted_waits(lib):
    plan = _plan(lib, 2)
    assert plan == [3, 4, 1, 6, 1, 8]
    for nkb in (1, 2, 3, 16, 56):
        slack = _simulate_two_wg(plan, nkb)
        assert min(slack) >= 0 and min(slack) == 0, (nkb, slack[:12])
    assert set(_simulate_two_wg(plan, 16)[8:]) == {0}
