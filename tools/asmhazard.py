"""Static check of a hand-counted vmcnt pipeline in hipcc's ISA output (gfx950).

Inline-asm loads are invisible to hipcc's s_waitcnt bookkeeping: between an asm `buffer_load` and the asm
`s_waitcnt vmcnt(N)` that retires it, NOTHING may read or write its destination registers (a register-allocator
copy there reads garbage).  This script walks a kernel's instruction stream in layout order (twice, to carry the
loop's state over the back edge), keeps the queue of asm-issued vector-memory operations, retires them at asm waits
(in order: all but the newest N), and reports every instruction that touches a register with a load in flight.

  python tools/asmhazard.py /tmp/isa/spconv_dma.s conv_dma16ILb0
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main(path, name):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and ": ;" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    queue = []   # (dest regs, text, line)
    lqueue = []  # the same for LGKM operations
    in_asm = False
    hazards = []
    for rep in range(2):
        for n, l in enumerate(body):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            op = t.split()[0]
            flying = set().union(*[q[0] for q in queue]) if queue else set()
            lflying = set().union(*[q[0] for q in lqueue]) if lqueue else set()
            if in_asm and op.startswith("buffer_load"):
                args = t[len(op):].split(",")
                lds = " lds" in t
                dest = set() if lds else regs(args[0])
                src = regs(",".join(args[0 if lds else 1:]))
                if src & flying or dest & flying:
                    hazards.append((start + n + 1, t, sorted((src | dest) & flying)))
                queue.append((dest, t, start + n + 1))
                continue
            if op.startswith("buffer_store") or op.startswith("global_store"):  # (asm or compiler-issued: both count)
                if regs(t) & flying:
                    hazards.append((start + n + 1, t, sorted(regs(t) & flying)))
                queue.append((set(), t, start + n + 1))
                continue
            # LDS reads issued from inline asm (ds_read_b64_tr_b16 ...): same rule against lgkmcnt.  Every LGKM operation
            # enters the queue (LDS returns in order; a scalar load may return early, which only retires MORE); any
            # s_waitcnt lgkmcnt(N), hand-written or hipcc's, retires all but the newest N.
            if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_memtime") or op.startswith("s_buffer_load"):
                dest = regs(t[len(op):].split(",")[0]) if (in_asm and op.startswith("ds_read")) else set()
                touched = regs(t)
                if touched & lflying:
                    hazards.append((start + n + 1, t, sorted(touched & lflying)))
                lqueue.append((dest, t, start + n + 1))
                continue
            if op == "s_waitcnt" and "lgkmcnt" in t:
                keep = int(re.search(r"lgkmcnt\((\d+)\)", t).group(1))
                while len(lqueue) > keep:
                    lqueue.pop(0)
                if not ("vmcnt" in t):
                    continue
            if in_asm and op == "s_waitcnt" and "vmcnt" in t:
                keep = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                while len(queue) > keep:
                    queue.pop(0)
                continue
            if regs(t) & (flying | lflying):
                hazards.append((start + n + 1, t, sorted(regs(t) & (flying | lflying))))
    seen = set()
    for ln, t, r in hazards:
        if (ln, t) in seen:
            continue
        seen.add((ln, t))
        print("HAZARD line %d: %s   (in flight: %s)" % (ln, t, r))
    print("%d hazards" % len(seen))
    return 1 if seen else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
