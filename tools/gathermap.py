#!/usr/bin/env python
"""Which layer shapes of one bench step go through doda_spconv_gather_ex, and how often (DODA_TRACE_GATHER=1 lines of the library,
counted per step).  Read next to the conv_* rows of a kernel-statistics summary to see which instantiation serves which layer."""
import collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, DODA_TRACE_GATHER="1")
steps, warm = 4, 2
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warm), "--fp32-steps", "0",
                    "--no-cpu-baseline", "--kernel-reps", "0", "--no-train-entry", "--config5-steps", "0", "--refgraph-steps", "0"], env=env, capture_output=True, text=True)
c = collections.Counter(l for l in r.stderr.split("\n") if l.startswith("doda_gather"))
tot = steps + warm
for line, n in sorted(c.items(), key=lambda kv: (-int(kv[0].split("n_out=")[1].split()[0]), kv[0])):
    if n >= tot:
        print("%5.1f x  %s" % (n / tot, line[len("doda_gather "):]))
