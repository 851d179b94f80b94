"""Generated-code checks that need hipcc but no GPU.

conv3x3_s3x.hip issues the staging waves' prefetch loads and their counted `s_waitcnt vmcnt(N)` from inline asm, which is
only correct if hipcc neither copies nor reuses the destination registers while a load is in flight and adds no other
vector-memory instruction to that loop.  tools/check_async_loads.py verifies exactly that in the assembly hipcc produces
with the flags of csrc/Makefile; a compiler or source change that breaks the hand-made count fails here, on the CPU,
before anything runs on a GPU.
"""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_role_split_conv_counted_loads():
    import check_async_loads as chk
    assert chk.main() in (None, 0)


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_diagnostic_build_keeps_the_counted_loads(monkeypatch):
    """The diagnostic library (`make diag`: in-kernel phase stamps; tools/stamps*.py) is what the cycle
    analysis rests on, so it has to run the same staging loops.  Round 3 found that it had not: run-time ablation flags and
    branched stamps inside the staging loops made hipcc copy in-flight load registers (the weight gradients' first tile per
    workgroup was staged from registers the loads had not reached: non-finite gradients at full size).  Ablations are
    compile-time now (-DXSD_ABL=n) and the staging stamps branch-free; this holds the -DXSD_DIAG code to the same check."""
    import check_async_loads as chk
    monkeypatch.setenv("XSD_CHECK_FLAGS", "-DXSD_DIAG")
    assert chk.main() in (None, 0)


def test_checker_rejects_broken_loops():
    """the checker itself: a copy of a destination while the load is in flight, a foreign vmem op, a wrong count"""
    import check_async_loads as chk

    def loop(body):
        pro = ["\tbuffer_load_dwordx4 v[%d:%d], v90, s[0:3], 0 offen" % (4 * k, 4 * k + 3) for k in range(15)]
        pro = pro + ["\ts_waitcnt vmcnt(0)"] + pro
        return "\n".join(pro + [".LBB0_1:                ; =>This Inner Loop Header: Depth=1"] + body +
                         ["\ts_cbranch_scc1 .LBB0_1", "\ts_endpgm"])

    def rounds(wait="s_waitcnt vmcnt(14)", extra=None):
        out = []
        for k in range(15):
            out += ["\t" + wait, "\tv_mov_b32_e32 v100, v%d" % (4 * k), "\tbuffer_load_dwordx4 v[%d:%d], v90, s[0:3], 0 offen" % (4 * k, 4 * k + 3)]
            if extra and k == 3:
                out += ["\t" + extra]
        return out

    assert chk.check(loop(rounds())) > 0
    with pytest.raises(AssertionError):
        chk.check(loop(rounds(extra="v_mov_b32_e32 v101, v2")))          # reads load 0's destination outside its window
    with pytest.raises(AssertionError):
        chk.check(loop(rounds(extra="global_load_dword v101, v[102:103], off")))   # foreign vector-memory instruction
    with pytest.raises(AssertionError):
        chk.check(loop(rounds(wait="s_waitcnt vmcnt(13)")))              # wrong count for single rounds


def test_checker_handles_lds_dma_pieces():
    """LDS-DMA pieces (no register destination) share the vmcnt queue: they must have landed at the barrier, and a wait that
    retires them early (or a register load's data too early) is rejected"""
    import check_async_loads as chk

    NX, ND = 10, 5

    def loop(final_wait):
        pro = ["\tbuffer_load_dwordx4 v[%d:%d], v90, s[0:3], 0 offen" % (4 * k, 4 * k + 3) for k in range(NX)]
        pro = pro + ["\ts_waitcnt vmcnt(0)"] + pro
        body = ["\tbuffer_load_dwordx4 v91, s[4:7], s8 offen lds" for _ in range(ND)]
        for k in range(0, NX, 2):
            body += ["\ts_waitcnt vmcnt(13)", "\tv_mov_b32_e32 v100, v%d" % (4 * k), "\tv_mov_b32_e32 v101, v%d" % (4 * k + 4),
                     "\tbuffer_load_dwordx4 v[%d:%d], v90, s[0:3], 0 offen" % (4 * k, 4 * k + 3),
                     "\tbuffer_load_dwordx4 v[%d:%d], v90, s[0:3], 0 offen" % (4 * k + 4, 4 * k + 7)]
        body += ["\t" + final_wait, "\ts_barrier"] if final_wait else ["\ts_barrier"]
        return "\n".join(pro + [".LBB0_1:                ; =>This Inner Loop Header: Depth=1"] + body + ["\ts_cbranch_scc1 .LBB0_1", "\ts_endpgm"])

    assert chk.check(loop("s_waitcnt vmcnt(10)"), NX + ND) > 0
    with pytest.raises(AssertionError):
        chk.check(loop(None), NX + ND)                          # pieces still in flight at the barrier
    with pytest.raises(AssertionError):
        chk.check(loop("s_waitcnt vmcnt(8)"), NX + ND)          # retires two register loads nobody consumes before the next wait

