"""ISA check for the hand-counted prefetch of the staging waves of the role-split kernels (csrc/conv3x3_s3x.hip,
csrc/conv3x3_h2x.hip, csrc/wgrad_s3x.hip, csrc/wgrad_h2x.hip).

The staging loop issues its global loads and their `s_waitcnt vmcnt(14)` from inline asm, so hipcc does not know that a
load's destination registers are written asynchronously.  That is only correct if, in the generated code,
  * the loop holds exactly N asm loads (15 for the conv, 11 / 19 for the weight gradients), each group of m loads behind one asm wait
    `vmcnt(N - m)` -- and no other vector-memory instruction (anything else would shift the hand-made count),
  * the destination registers of a load are touched nowhere in the loop except between its group's wait and the last
    load of the group, and not after its own reload was issued (no copy made while the load is in flight, no reuse
    as a temporary).
This script compiles the kernel to assembly and verifies both; tests/test_isa.py runs it on every CPU test run.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xmm-superres-denoise_amd", "csrc")
KERNELS = (("conv3x3_s3x.hip", 15), ("conv3x3_h2x.hip", 15), ("wgrad_s3x.hip", 11), ("wgrad_h2x.hip", 19))     # source, counted loads per pass of the staging loop


def regs_of(tok):
    """registers named by one operand token: v12 -> {12}; v[2:5] -> {2,3,4,5}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def check(asm_text, NLOADS=15):
    lines = asm_text.splitlines()
    heads = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
    # the staging loop is the one that holds the asm waits
    loops = []
    for hpos in heads:
        hdr = lines[hpos].split(":")[0].strip()                       # .LBB0_239
        tag = "Header=" + hdr.lstrip(".L")                             # blocks of the loop carry "Header=BB0_239"
        members = [i for i, l in enumerate(lines) if i == hpos or tag in l]
        labels = {lines[i].split(":")[0].strip() for i in members}
        start = min(members)
        end = max(j for j, l in enumerate(lines) if re.search(r"s_c?branch\w*\s+(\S+)", l) and l.split()[-1] in labels)
        loops.append((start, end))
    cand = [(a, b) for a, b in loops if sum("buffer_load_dwordx4" in l for l in lines[a:b]) == NLOADS]
    assert len(cand) == 1, "staging loop not found (loops with %d buffer loads: %d)" % (NLOADS, len(cand))
    a, b = cand[0]
    body = [l.split(";")[0].strip() for l in lines[a:b + 1]]
    body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
    events = []          # (index in body, kind, regs)
    for i, l in enumerate(body):
        op = l.split()[0]
        if op.startswith(("global_load", "buffer_load")):
            events.append((i, "load", regs_of(l.split(",")[0])))
        elif re.match(r"s_waitcnt vmcnt\([1-9][0-9]*\)$", l):      # a counted wait (its count is checked below); vmcnt(0) is a drain
            events.append((i, "wait", None))
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")) or (op == "s_waitcnt" and "vmcnt" in l):
            raise AssertionError("foreign vector-memory instruction in the staging loop: " + l)
    # group the events: every wait owns the loads up to the next wait
    groups = []
    for i, kind, regs in events:
        if kind == "wait":
            groups.append([i, []])
        else:
            assert groups, "a load precedes the first counted wait of the loop"
            groups[-1][1].append((i, regs))
    loads = [ld for g in groups for ld in g[1]]
    assert len(loads) == NLOADS, "expected %d counted loads, got %d" % (NLOADS, len(loads))
    for w, lds in groups:
        m = len(lds)
        assert m >= 1
        # when the group's data is needed, the loads issued after its youngest member number NLOADS - m
        want = "s_waitcnt vmcnt(%d)" % (NLOADS - m)
        assert body[w].startswith(want), "wait owning %d loads must be %s, found %s" % (m, want, body[w])
        last = lds[-1][0]
        for _, dst in lds:
            assert len(dst) == 4
            for i, l in enumerate(body):
                if w <= i <= last:
                    continue
                assert not (regs_of(l) & dst), "load destination v%s touched outside its window: %s" % (sorted(dst), l)
        # inside the window a destination may be read before its own reload only
        for k, (li, dst) in enumerate(lds):
            for i in range(li + 1, last + 1):
                assert not (regs_of(body[i]) & dst), "destination v%s used after its reload was issued: %s" % (sorted(dst), body[i])
    # prologue: the 15 loads of half-step 1 are consumed by the loop's first pass -> same registers, untouched until the loop;
    # the 15 loads of half-step 0 before them are followed by an explicit vmcnt(0) -> untouched until that wait
    loop_dsts = [dst for _, dst in loads]
    pre = [l.split(";")[0].strip() for l in lines[:a]]
    pl = [i for i, l in enumerate(pre) if l.startswith(("global_load_dwordx4", "buffer_load_dwordx4"))]
    # the staging branch's prologue is the code right before the loop: its last 30 loads
    assert len(pl) >= 2 * NLOADS
    second, first = pl[-NLOADS:], pl[-2 * NLOADS:-NLOADS]
    for k, i in enumerate(second):
        dst = regs_of(pre[i].split(",")[0])
        assert dst == loop_dsts[k], "prologue load %d lands in v%s, the loop expects v%s" % (k, sorted(dst), sorted(loop_dsts[k]))
        for l in pre[i + 1:]:
            assert not (regs_of(l) & dst) or l.startswith(("global_load", "buffer_load")) and regs_of(l.split(",")[0]) != dst and not (regs_of(l.split(",", 1)[1]) & dst), \
                "prologue load %d destination touched before the loop: %s" % (k, l)
    w0 = [i for i in range(first[-1], second[0]) if pre[i].startswith("s_waitcnt vmcnt(0)")]
    assert w0, "no vmcnt(0) between the two prologue batches"
    for k, i in enumerate(first):
        dst = regs_of(pre[i].split(",")[0])
        for l in pre[i + 1:w0[0]]:
            assert not (regs_of(l) & dst), "first-batch load %d destination touched before its wait: %s" % (k, l)
    return len(body)


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    kernels = KERNELS
    extra = os.environ.get("XSD_CHECK_FLAGS", "").split()          # experiment builds: e.g. XSD_CHECK_FLAGS=-DV3_TH_ROWS=8
    if os.environ.get("XSD_CHECK_ONLY"):                            # "wgrad_h2x.hip:19" -> that kernel with that load count
        name, n = os.environ["XSD_CHECK_ONLY"].split(":")
        kernels = ((name, int(n)),)
    for src, nloads in kernels:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",   # the flags of csrc/Makefile
                            "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out] + extra, check=True, stderr=subprocess.DEVNULL)
            n = check(open(out).read(), nloads)
        print("%s staging loop: %d instructions, %d counted loads, destinations private to their windows" % (src, n, nloads))


if __name__ == "__main__":
    sys.exit(main())
