#!/usr/bin/env python
"""Static check of the compiled kernels: can a wave reach an s_barrier with an LDS write still in flight?

For every kernel of the given csrc/*.hip files (all by default; cross-compiled to gfx950 ISA with the library's flags, no GPU needed)
the control-flow graph is rebuilt from the assembly text and a forward may-analysis runs over it: a ds_write* / ds_* update without
a following `s_waitcnt ... lgkmcnt(0)` leaves the state "LDS store pending"; states are OR-ed at joins (back edges included) until
nothing changes; an s_barrier reached in the pending state is reported.  The same analysis runs a second time for LDS-DMA
(`buffer_load_* ... lds`: the LDS write is counted by vmcnt, so the state clears at `s_waitcnt vmcnt(0)` only -- every wait of the one
kernel that uses it, spconv_wdma, is vmcnt(0)).  That count is informational: a double-buffered kernel holds barriers with the NEXT
buffer's DMA in flight by design (wgrad_dma16: the eight barriers of its two inlined flush() sites, which exchange through the buffer
the DMA does not target; the barriers that publish a staged buffer -- prologue and the two loop tops -- must NOT be in the list).  This is the hazard of DESIGN.md §9 (round 5): hipcc's
wait-count pass lost the stores of a conditionally executed block across a loop back edge and emitted a bare s_barrier in the tilebook
builder's sort.  Since then every barrier of the library is doda_sync() (explicit wait): the expected result is zero reports;
`--source <file.hip>` checks any other source (e.g. an old revision: `git show <rev>:doda_amd/csrc/tilebook.hip > /tmp/old.hip`).
usage: isa_barrier_check.py [name ...] | --source path.hip"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from doda_amd.build import FLAGS, HIPCC  # noqa: E402

LDS_WRITE = re.compile(r"^\s*ds_(write|add|sub|rsub|inc|dec|min|max|and|or|xor|mskor|cmpst|wrxchg|append|consume|pk_add|bpermute_never)")
WAIT = re.compile(r"^\s*s_waitcnt\b(.*)")
BRANCH = re.compile(r"^\s*(s_cbranch\w*|s_branch)\s+(\.L\w+)")
LABEL = re.compile(r"^(\.L\w+):")
KERNEL = re.compile(r"^(_Z\w+):")


def waits_lgkm0(rest):
    m = re.search(r"lgkmcnt\((\d+)\)", rest)
    if m:
        return int(m.group(1)) == 0
    m = re.search(r"^\s*(0x[0-9a-fA-F]+|\d+)\s*$", rest)      # raw immediate: lgkmcnt = bits 11:8
    if m:
        return ((int(m.group(1), 0) >> 8) & 0xF) == 0
    return False


def waits_vm0(rest):
    m = re.search(r"vmcnt\((\d+)\)", rest)
    if m:
        return int(m.group(1)) == 0
    m = re.search(r"^\s*(0x[0-9a-fA-F]+|\d+)\s*$", rest)      # raw immediate: vmcnt = bits 15:14 | 3:0
    if m:
        v = int(m.group(1), 0)
        return ((v & 0xF) | ((v >> 14) & 3) << 4) == 0
    return False


# LDS-DMA (`buffer_load_* ... lds`, `global_load_lds_*`): the LDS write is counted by vmcnt, not lgkmcnt
LDS_DMA = re.compile(r"^\s*(buffer_load_\w+\s.*\blds\b|global_load_lds_\w+)")
HAZARDS = {"store": (LDS_WRITE, waits_lgkm0), "dma": (LDS_DMA, waits_vm0)}


def check_kernel(name, lines, hazard="store"):
    SET, cleared = HAZARDS[hazard]
    # basic blocks
    leaders = {0}
    for i, l in enumerate(lines):
        if LABEL.match(l):
            leaders.add(i)
        if BRANCH.match(l) or "s_endpgm" in l or "s_setpc" in l:
            leaders.add(i + 1)
    order = sorted(x for x in leaders if x < len(lines))
    blocks = [(a, b) for a, b in zip(order, order[1:] + [len(lines)])]
    label_block = {}
    for bi, (a, b) in enumerate(blocks):
        for i in range(a, b):
            m = LABEL.match(lines[i])
            if m:
                label_block[m.group(1)] = bi
            elif lines[i].strip() and not lines[i].strip().startswith(";"):
                break
    succ = []
    for bi, (a, b) in enumerate(blocks):
        s = set()
        last = next((lines[i] for i in range(b - 1, a - 1, -1) if lines[i].strip() and not lines[i].strip().startswith((";", "."))), "")
        m = BRANCH.match(last)
        if m:
            if m.group(2) in label_block:
                s.add(label_block[m.group(2)])
            if m.group(1) != "s_branch" and bi + 1 < len(blocks):
                s.add(bi + 1)
        elif "s_endpgm" in last or "s_setpc" in last:
            pass
        elif bi + 1 < len(blocks):
            s.add(bi + 1)
        succ.append(s)
    state_in = [False] * len(blocks)
    reports = {}
    work = list(range(len(blocks)))
    while work:
        bi = work.pop()
        pending = state_in[bi]
        a, b = blocks[bi]
        for i in range(a, b):
            l = lines[i]
            if SET.match(l):
                pending = i
            else:
                m = WAIT.match(l)
                if m and cleared(m.group(1)):
                    pending = False
                elif re.match(r"^\s*s_barrier\b", l) and pending is not False:
                    reports[i] = pending if pending is not True else None
        for t in succ[bi]:
            if pending is not False and state_in[t] is False:
                state_in[t] = pending if pending is not True else True
                work.append(t)
    return reports


def check_file(src, extra=(), verbose=True):
    out = "/tmp/_isa_check.s"
    r = subprocess.run([HIPCC, *FLAGS, *extra, "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
    if r.returncode:
        print("%s: compile failed\n%s" % (src, r.stderr[-1500:]))
        return -1
    text = open(out).read().splitlines()
    starts = [i for i, l in enumerate(text) if KERNEL.match(l)]
    n_bar = n_rep = n_dma = n_dma_bar = 0
    for k, a in enumerate(starts):
        b = starts[k + 1] if k + 1 < len(starts) else len(text)
        body = text[a:b]
        n_bar += sum(1 for l in body if re.match(r"^\s*s_barrier\b", l))
        n_dma += sum(1 for l in body if LDS_DMA.match(l))
        for hazard, what in (("store", "LDS store"), ("dma", "LDS-DMA load")):
            rep = check_kernel(text[a], body, hazard)
            if hazard == "dma":     # informational: a double-buffered kernel holds barriers with the NEXT buffer's DMA in flight by design
                n_dma_bar += len(rep)
                if rep and verbose:
                    print("   note: %d barrier(s) with an LDS-DMA load in flight, asm lines %s" % (len(rep), [a + x for x in sorted(rep)]))
                continue
            if rep:
                dem = subprocess.run(["c++filt", text[a].rstrip(":")], capture_output=True, text=True).stdout.strip()
                for line, w in sorted(rep.items())[:6]:
                    n_rep += 1
                    print("   %s: s_barrier at line %d reachable with a pending %s (e.g. `%s`, line %s)" % (
                        dem[:90], a + line, what, body[w].strip() if isinstance(w, int) else "?", a + w if isinstance(w, int) else "?"))
    print("%-24s kernels %3d  barriers %4d  reachable with a pending LDS store: %d   (LDS-DMA loads %2d, barriers with one in flight %d)" % (
        os.path.basename(src), len(starts), n_bar, n_rep, n_dma, n_dma_bar), flush=True)
    check_file.last_dma_barriers = n_dma_bar
    return n_rep


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--source":
        sys.exit(1 if check_file(args[1], ["-I", os.path.join(ROOT, "doda_amd", "csrc"), "-I", os.path.join(ROOT, "include")]) else 0)
    names = args or sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(ROOT, "doda_amd", "csrc", "*.hip")))
    total = 0
    for n in names:
        total += max(0, check_file(os.path.join(ROOT, "doda_amd", "csrc", n + ".hip")))
    print("total reports:", total)
    sys.exit(1 if total else 0)
