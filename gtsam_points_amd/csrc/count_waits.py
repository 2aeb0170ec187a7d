"""Build-time checks of the hand-placed vmcnt arithmetic (called by the Makefile on the device assembly of gp_vgicp.hip).

vgicp_stream_kernel issues every vector-memory instruction of its streaming part from inline asm and counts the requests in flight
itself.  Two things can silently break that, and both are checked here per instantiation:

  compiler_vmcnt_waits   a load that hipcc tracks on its own makes it insert `s_waitcnt vmcnt(N)` instructions of its own -- normally vmcnt(0)
                         in front of an LDS read or of a register the tracked load writes -- which drain the source requests in flight and
                         serialise the pipeline (it happened three times while the second generation was written).  Counted: vmcnt waits
                         that are NOT inside an inline-asm block while asm-issued requests are in flight (layout order).  Compiler waits
                         with nothing of the asm's in flight -- the fused finalize behind the stream, whose row loads are the compiler's
                         own -- are listed as idle_vmcnt_waits and are harmless.
  inflight_reg_touches   the destination registers of an asm-issued load hold nothing until the matching wait, but the compiler believes
                         they are defined at the issue: a copy it places in between (a phi at a loop back-edge, the operand copy in front of
                         one of two alternative wait statements) reads them before the data lands (round 3: wild record offsets, a memory
                         fault on the GPU box).  Counted, walking the text in layout order: instructions outside asm blocks that read or
                         write a register of a load still in flight (in-order retirement: an asm `s_waitcnt vmcnt(N)` retires all but the
                         N youngest requests; LDS-DMA requests have no destination registers but take part in the count).
tests/test_build_cpu.py wants 0 for both -- and so does the BUILD: this script exits non-zero (the Makefile stops) when either count is not 0 for some instantiation."""
import re
import sys

KERNEL = re.compile(r"^(_ZN2gp19vgicp_stream_kernel\w+):")
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


name, inside = None, False
waits, idle, touches, examples = {}, {}, {}, {}
inflight = []  # one entry per asm-issued vector-memory request, oldest first: set of destination registers (empty for LDS-DMA)
for line in open(sys.argv[1]):
    m = KERNEL.match(line)
    if m:
        name = m.group(1)
        waits[name], idle[name], touches[name], examples[name] = 0, 0, 0, []
        inflight = []
        continue
    if name is None:
        continue
    code = line.split(";")[0].strip()
    if line.startswith(".Lfunc_end"):  # (a kernel may hold several s_endpgm)
        name = None
        continue
    if "#ASMSTART" in line:
        inside = True
        continue
    if "#ASMEND" in line:
        inside = False
        continue
    if not code or code.endswith(":") or code.startswith("."):
        continue
    if inside:
        if code.startswith("global_load_lds"):
            inflight.append(set())
        elif code.startswith("global_load"):
            inflight.append(regs_of(code.split(",")[0]))
        else:
            w = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", code)
            if w:
                keep = int(w.group(1))
                inflight = inflight[len(inflight) - keep:] if keep else []
        continue
    if "s_waitcnt" in code and "vmcnt" in code:
        if inflight:
            waits[name] += 1
        else:
            idle[name] += 1
    pending = set().union(*inflight) if inflight else set()
    if pending and not code.startswith("s_") and (regs_of(code) & pending):
        touches[name] += 1
        if len(examples[name]) < 3:
            examples[name].append(code)
bad = []
for k in sorted(waits):
    print(f"{k} compiler_vmcnt_waits {waits[k]} inflight_reg_touches {touches[k]} idle_vmcnt_waits {idle[k]}" + ("   e.g. " + " | ".join(examples[k]) if examples[k] else ""))
    if waits[k] or touches[k]:
        bad.append(k)
# the guarantee is "this compiler": a build whose compiler places a wait or a copy where the hand count forbids it must FAIL, not wait for a test
if not waits:
    sys.stderr.write("count_waits.py: no vgicp_stream_kernel instantiation found in the device assembly\n")
    sys.exit(2)
if bad:
    sys.stderr.write("count_waits.py: the hand-counted vmcnt schedule of gp_vgicp_stream.hpp does not hold with this compiler for:\n  " + "\n  ".join(bad)
                     + "\n(compiler_vmcnt_waits / inflight_reg_touches must be 0; validated with the compiler named in gp_vgicp_stream.hpp)\n")
    sys.exit(1)
