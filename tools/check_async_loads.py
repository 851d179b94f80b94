"""ISA check for the hand-counted prefetch of the staging waves of the role-split kernels (csrc/conv3x3_s3x.hip,
csrc/conv3x3_h2x.hip, csrc/wgrad_s3x.hip, csrc/wgrad_h2x.hip).

The staging loop issues its global loads and their `s_waitcnt vmcnt(14)` from inline asm, so hipcc does not know that a
load's destination registers are written asynchronously.  That is only correct if, in the generated code,
  * the loop holds exactly N asm vector-memory operations (15 for the conv -- register loads and, with the weights by
    LDS-DMA, `... lds` pieces --, 11 / 17 for the weight gradients) and no other vector-memory instruction or wait
    (anything else would shift the hand-made count),
  * with the operations retiring in issue order and `vmcnt(n)` returning when all but the youngest n have (the loop is
    simulated from the state the prologue leaves), no instruction touches a load's destination registers while that
    load is in flight (no copy, no reuse as a temporary, no reload), every wait retires exactly what is consumed before
    the next one (no over-waiting: that would shorten the prefetch), an LDS-DMA piece has landed before the barrier,
    and a pass leaves the loads in flight that it was entered with;
  * the kernel holds no `scratch_` instruction at all (production flags).
This script compiles the kernel to assembly and verifies both; tests/test_isa.py runs it on every CPU test run.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xmm-superres-denoise_amd", "csrc")
KERNELS = (("conv3x3_s3x.hip", 15), ("conv3x3_h2x.hip", 15), ("wgrad_s3x.hip", 11), ("wgrad_h2x.hip", 17))     # source, counted loads per pass of the staging loop


def regs_of(tok):
    """registers named by one operand token: v12 -> {12}; v[2:5] -> {2,3,4,5}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def check(asm_text, NLOADS=15):
    """NLOADS: vector-memory operations per pass of the staging loop (register loads + LDS-DMA pieces)"""
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
    # (a loop whose vector-memory operations are all LDS-DMA pieces -- the H2-input variant of the conv -- has no register
    # destination to protect and drains with vmcnt(0): not a candidate)
    cand = [(a, b) for a, b in loops if sum("buffer_load_dwordx4" in l for l in lines[a:b]) == NLOADS
            and any("buffer_load_dwordx4" in l and not l.split(";")[0].rstrip().endswith(" lds") for l in lines[a:b])]
    assert len(cand) == 1, "staging loop not found (loops with %d buffer loads: %d)" % (NLOADS, len(cand))
    a, b = cand[0]
    body = [l.split(";")[0].strip() for l in lines[a:b + 1]]
    body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]

    def is_load(l):
        return l.split()[0].startswith(("global_load", "buffer_load"))

    def is_dma(l):
        return is_load(l) and (l.endswith(" lds") or "_load_lds_" in l)

    kinds = []           # per body line: ("rload", dst, srcs) | ("dma", srcs) | ("wait", N) | ("barrier",) | ("other", regs)
    for l in body:
        op = l.split()[0]
        m = re.match(r"s_waitcnt vmcnt\(([1-9][0-9]*)\)$", l)      # a counted wait; vmcnt(0) would drain the prefetch
        if is_dma(l):
            kinds.append(("dma", regs_of(l)))
        elif is_load(l):
            kinds.append(("rload", regs_of(l.split(",")[0]), regs_of(l.split(",", 1)[1])))
        elif m:
            kinds.append(("wait", int(m.group(1))))
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")) or (op == "s_waitcnt" and "vmcnt" in l):
            raise AssertionError("foreign vector-memory instruction in the staging loop: " + l)
        elif op == "s_barrier":
            kinds.append(("barrier",))
        else:
            kinds.append(("other", regs_of(l)))
    nvm = sum(k[0] in ("rload", "dma") for k in kinds)
    assert nvm == NLOADS, "expected %d counted loads, got %d" % (NLOADS, nvm)
    loop_dsts = [k[1] for k in kinds if k[0] == "rload"]
    assert all(len(d) == 4 for d in loop_dsts)
    assert any(k[0] == "wait" for k in kinds), "no counted wait in the loop"

    # prologue: the register loads of the half-step the loop's first pass consumes -> same registers in the same order,
    # untouched until the loop; the batch before them is followed by an explicit vmcnt(0) -> untouched until that wait
    nreg = len(loop_dsts)
    pre = [l.split(";")[0].strip() for l in lines[:a]]
    # (the counted loads are `buffer_load_dwordx4` from inline asm; hipcc's own loads in the prologue -- descriptor table, bias, since
    # round 6 issued BEHIND the first batch -- are `global_load_*`: they must not be mistaken for a batch, and if one of them lands in
    # a first-batch destination before the wait, the touch test below catches it)
    pl = [i for i, l in enumerate(pre) if l.startswith("buffer_load_dwordx4") and not is_dma(l)]
    assert len(pl) >= 2 * nreg
    second, first = pl[-nreg:], pl[-2 * nreg:-nreg]
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

    # the loop, simulated: vector-memory operations retire in issue order, `vmcnt(N)` returns when all but the youngest N
    # have.  Two passes, the first entered with the prologue's second batch in flight; the state a pass leaves must be
    # the state it was entered with.
    queue = [("r", d) for d in loop_dsts]                             # oldest first
    for npass in range(2):
        entered = list(queue)
        for i, k in enumerate(kinds):
            inflight = set().union(*[q[1] for q in queue if q[0] == "r"]) if queue else set()
            if k[0] == "wait":
                n = k[1]
                retired = queue[:max(0, len(queue) - n)]
                assert retired, "counted wait retires nothing: " + body[i]
                queue = queue[len(retired):]
                # tight: everything a wait retires is consumed before the next wait (a register load by an instruction
                # that touches its destination, an LDS-DMA piece by the barrier)
                nxt = next((j for j in range(i + 1, len(kinds)) if kinds[j][0] == "wait"), len(kinds))
                for q in retired:
                    if q[0] == "r":
                        used = any(kinds[j][0] in ("other", "rload", "dma") and (regs_of(body[j]) & q[1]) for j in range(i + 1, nxt))
                        assert used, "%s retires the load into v%s, which is not consumed before the next wait" % (body[i], sorted(q[1]))
                    else:
                        assert any(kinds[j][0] == "barrier" for j in range(i + 1, nxt)), "%s retires an LDS-DMA piece early" % body[i]
            elif k[0] == "rload":
                assert not (k[1] & inflight), "reload of v%s while the previous load into it is in flight" % sorted(k[1])
                assert not (k[2] & inflight), "load address taken from a register with a load in flight: " + body[i]
                queue.append(("r", k[1]))
            elif k[0] == "dma":
                assert not (k[1] & inflight), "LDS-DMA address taken from a register with a load in flight: " + body[i]
                queue.append(("d", set()))
            elif k[0] == "barrier":
                assert not any(q[0] == "d" for q in queue), "LDS-DMA piece still in flight at the barrier"
            else:
                assert not (k[1] & inflight), "load destination touched while the load is in flight: %s" % body[i]
        assert queue == entered, "a pass of the loop does not leave the in-flight loads it was entered with"
    return len(body)


def makefile_flags():
    """the device-code flags of csrc/Makefile (`make flags` prints $(CXXFLAGS)), so that this check compiles exactly what the
    build compiles; -fPIC / warnings are irrelevant to the generated device code and dropped"""
    out = subprocess.run(["make", "-s", "-C", CSRC, "flags"], check=True, stdout=subprocess.PIPE, text=True).stdout.split()
    keep = [f for f in out if not f.startswith("-W") and f != "-fPIC"]
    assert any(f.startswith("--offload-arch=") for f in keep) and "-O3" in keep, keep
    return keep


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    kernels = KERNELS
    flags = makefile_flags()
    extra = os.environ.get("XSD_CHECK_FLAGS", "").split()          # experiment builds: e.g. XSD_CHECK_FLAGS=-DV3_TH_ROWS=8
    if os.environ.get("XSD_CHECK_ONLY"):                            # "wgrad_h2x.hip:19" -> that kernel with that load count
        name, n = os.environ["XSD_CHECK_ONLY"].split(":")
        kernels = ((name, int(n)),)
    for src, nloads in kernels:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out] + extra, check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
            # one check per kernel of the file (conv3x3_h2x.hip holds the catch-all kernel and its per-kind instances): a kernel's
            # text runs from its label to its s_endpgm / .Lfunc_end
            funcs = re.split(r"(?m)^(?=\S+:\s*; @)", text)
            funcs = [f for f in funcs if re.match(r"\S+:\s*; @", f) and "buffer_load_dwordx4" in f]
            assert funcs, "no kernel with buffer loads in " + src
            n = 0
            for f in funcs:
                n = check(f, nloads)
            if len(funcs) > 1:
                print("%s: %d kernels checked" % (src, len(funcs)))
            # no register spills: a scratch reload under these kernels' memory load is a ~4 us round trip, and the ones hipcc
            # once put into the conv epilogues and the accumulator initialisation cost 2-11 %
            spills = [l.strip() for l in text.splitlines() if l.strip().startswith("scratch_")]
            assert not spills or os.environ.get("XSD_CHECK_FLAGS"), "%s: %d scratch instructions, e.g. %s" % (src, len(spills), spills[0])
        print("%s staging loop: %d instructions, %d counted loads, destinations private to their windows" % (src, n, nloads))


if __name__ == "__main__":
    sys.exit(main())
