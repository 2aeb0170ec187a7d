"""Build-time check of the hand-placed vmcnt arithmetic (called by the Makefile on the device assembly of gp_vgicp.hip).

vgicp_pipeline2_kernel issues every vector-memory instruction from inline asm and counts the requests in flight itself.  A load that
hipcc tracks on its own would make it insert `s_waitcnt vmcnt(N)` instructions of its own -- normally vmcnt(0) in front of an LDS read
or of a register the tracked load writes -- which drain the source requests in flight and silently serialise the pipeline (it happened
three times while the kernel was written: the LDS-DMA builtin, the surface-validation normals, the per-lane loads of the partial-wave
path).  This script lists, per instantiation, the vmcnt waits that are NOT inside an inline-asm block; tests/test_build_cpu.py wants 0."""
import re
import sys

name, inside, counts = None, False, {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_ZN2gp(?:22vgicp_pipeline2_kernel|19vgicp_stream_kernel)\w+):", line)
    if m:
        name = m.group(1)
        counts[name] = 0
        continue
    if name is None:
        continue
    if "s_endpgm" in line:
        name = None
    elif "#ASMSTART" in line:
        inside = True
    elif "#ASMEND" in line:
        inside = False
    elif "s_waitcnt" in line and "vmcnt" in line and not inside:
        counts[name] += 1
for k, v in sorted(counts.items()):
    print(f"{k} compiler_vmcnt_waits {v}")
