#!/usr/bin/env python
"""Static check of conv_fast's load ring (csrc/spconv_gather.hip): no compiler-visible vector-memory instruction between ring loads.

The unit loop of conv_fast keeps D register sets filled by INLINE-ASM buffer loads that hipcc's wait-count pass does not see, with
hand-placed `s_waitcnt vmcnt((D-1) L)`: "at most (D-1) L loads outstanding" means "the oldest ring unit has arrived" only if every
outstanding vector-memory instruction IS a ring load.  hipcc may move an ordinary load (a `__builtin_amdgcn_raw_buffer_load_*`, a
pointer dereference) across `asm volatile` statements — they carry no memory clobber — so a request placed in front of the loop and
consumed behind it can be sunk between ring loads: the count is then off by one and a unit is consumed before it has arrived.  Round 6
did exactly that (the epilogue's operands requested early: one instantiation, PBF16P with statistics, got them between ring loads —
found in the disassembly while a flaky two-rank test was being chased — its cause was elsewhere) and removed it again.

The check works on the built object (doda_amd/csrc/_obj/spconv_gather.o, no GPU needed): per conv_fast kernel, ring loads are the
buffer loads with a scalar-register soffset (every compiler-visible buffer access of that file passes the literal 0), and between the
first and the last of them no other buffer / global / flat / scratch load or store may appear.
usage: isa_ring_check.py [object]   -> prints the offenders, exit code 1 if any"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
RING = re.compile(r"^\s*buffer_load_dwordx[24]\s+v\[\d+:\d+\],\s*v\d+,\s*s\[\d+:\d+\],\s*s\d+\s+offen\b")
VMEM = re.compile(r"^\s*(buffer_(load|store|atomic)\w*|global_(load|store|atomic)\w*|flat_(load|store|atomic)\w*|scratch_(load|store)\w*)\s")
FUNC = re.compile(r"^[0-9a-f]+ <(.*)>:")


def check_kernel(body):
    """body: the disassembly lines of one kernel.  -> list of (line index, text) of vector-memory instructions that sit between the
    first and the last ring load and are not ring loads themselves."""
    ring = [k for k, l in enumerate(body) if RING.match(l)]
    if len(ring) < 2:
        return []
    lo, hi = ring[0], ring[-1]
    return [(k, body[k].strip()) for k in range(lo + 1, hi) if VMEM.match(body[k]) and not RING.match(body[k])]


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "o.o")
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, capture_output=True, check=False)
        dev = [p for p in os.listdir(tmp) if "gfx950" in p]
        if not dev:
            raise RuntimeError("no gfx950 code object in " + obj)
        return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, dev[0])], capture_output=True, text=True, check=True).stdout.split("\n")


def check_object(obj, name_filter="conv_fast"):
    lines = disassemble(obj)
    starts = [(i, FUNC.match(l).group(1)) for i, l in enumerate(lines) if FUNC.match(l)]
    report, n = {}, 0
    for (a, name), (b, _) in zip(starts, starts[1:] + [(len(lines), None)]):
        if name_filter not in name:
            continue
        n += 1
        bad = check_kernel(lines[a:b])
        if bad:
            report[name] = bad
    return n, report


if __name__ == "__main__":
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "doda_amd", "csrc", "_obj", "spconv_gather.o")
    n, rep = check_object(obj)
    print("%d conv_fast kernels, %d with a foreign vector-memory instruction inside the ring" % (n, len(rep)))
    for name, bad in rep.items():
        print(" ", name[:110])
        for k, t in bad[:6]:
            print("     +%d  %s" % (k, t[:100]))
    sys.exit(1 if rep else 0)
