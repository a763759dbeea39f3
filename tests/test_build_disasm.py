"""Build-time check of the hand-written wait counts of the tile SpMM (csrc/spmm_tile.hip).

The kernels issue LDS-DMA copies (`global_load_lds_dwordx4`, inline asm) and then -- BEFORE waiting -- the prefetch loads of the next
groups; the wait is a hand-written `s_waitcnt vmcnt(N)` with N = the number of VMEM loads issued after the copies, so that the
copies have landed while the prefetches stay in flight.  That is only correct if hipcc emits AT LEAST N vector-memory
instructions between the last copy and the wait: were it to scalarise or merge some of them (the wave-uniform record words, say),
vmcnt(N) would let copies stay in flight past the barrier -- a silent race.  build.sh keeps the device assembly of
spmm_tile.hip beside its object (-save-temps=obj: a by-product of the normal compile); this test counts."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "krylov.jl_amd", "build", "spmm_tile-hip-amdgcn-amd-amdhsa-gfx950.s")
VMEM = re.compile(r"(global_load|buffer_load|flat_load|scratch_load|global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic)")


def _functions():
    txt = open(ASM).read()
    for f in re.split(r"\n(?=_ZN4khip[^\n]*:\s*;\s*@)", txt):
        name = f.split(":", 1)[0]
        if not name.startswith("_ZN4khip"):
            continue
        ins = [l.strip() for l in f.split("\n")]
        yield name, [l for l in ins if l and not l.startswith((".", ";")) and not l.endswith(":")]


def _waits_after_dma(ins):
    """(N of an s_waitcnt vmcnt(N), VMEM instructions between the nearest preceding LDS-DMA copy and it, barrier follows) for every
    wait whose nearest preceding vmcnt event is a copy."""
    out = []
    for i, l in enumerate(ins):
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
        if not m:
            continue
        j, cnt, hit = i - 1, 0, None
        while j >= 0:
            if "global_load_lds" in ins[j]:
                hit = "dma"
                break
            if re.match(r"s_waitcnt vmcnt", ins[j]):
                hit = "wait"
                break
            if VMEM.match(ins[j]):
                cnt += 1
            j -= 1
        if hit == "dma":
            out.append((int(m.group(1)), cnt, any(x.startswith("s_barrier") for x in ins[i + 1:i + 4])))
    return out


@pytest.mark.skipif(not os.path.exists(ASM), reason="krylov.jl_amd/build.sh has not produced the device assembly of spmm_tile.hip")
def test_tile_spmm_wait_counts_match_the_emitted_loads():
    seen = {"spmm_tile_kernel": 0, "spmm_tile2_kernel": 0}
    for name, ins in _functions():
        kind = next((k for k in seen if ("%d%s" % (len(k), k)) in name), None)
        if kind is None:
            continue
        assert any("global_load_lds_dwordx4" in l for l in ins), name              # the copies are LDS-DMA
        waits = _waits_after_dma(ins)
        assert waits, name
        if kind == "spmm_tile2_kernel" and re.search(r"ELi\d+ELb1EEEv", name):
            # The look-ahead instantiation (round 6) has no counted wait of its own: before the one barrier of a group it waits for
            # EVERYTHING it has in flight -- `s_waitcnt vmcnt(0)` (+ lgkmcnt(0)) then `s_barrier` -- and the copies it issues
            # afterwards go to window slots the current group does not read (tests/test_gpu_block.py checks the bits).  hipcc's own
            # counted waits behind a copy only ever over-wait (the asm copies are not in its count).
            top = [i for i, l in enumerate(ins) if re.match(r"s_waitcnt vmcnt\(0\)", l) and any(x.startswith("s_barrier") for x in ins[i + 1:i + 4])]
            assert top, f"{name}: no vmcnt(0) + s_barrier pair"
            assert sum(1 for l in ins if l.startswith("s_barrier")) >= 2, name
            seen["ahead"] = seen.get("ahead", 0) + 1
            continue
        for n, cnt, _barrier in waits:
            # at least n younger vector-memory instructions: vmcnt(n) then implies every copy has landed
            assert cnt >= n, f"{name}: s_waitcnt vmcnt({n}) with only {cnt} vector-memory instructions after the last LDS-DMA copy"
        # the hand-written wait itself: exact (a larger count would only cost speed, a smaller one is the race)
        exact = [w for w in waits if w[0] == w[1] and w[0] > 0]
        assert exact, (name, waits)
        if kind == "spmm_tile2_kernel":
            assert any(b for _n, _c, b in exact), f"{name}: the wait before the s_barrier is not the exact one: {waits}"
        seen[kind] += 1
    assert seen["spmm_tile_kernel"] >= 16 and seen["spmm_tile2_kernel"] >= 8 and seen.get("ahead", 0) >= 8, seen
