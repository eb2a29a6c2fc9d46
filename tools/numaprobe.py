"""Launch-issue cost from each NUMA node's CPUs (2000 tiny launches each, three rounds) next to what sysfs says about
the GPU, then the bench step pinned to each node in turn (in-process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd.host import device_numa_node, _parse_cpulist
allowed = os.sched_getaffinity(0)
nodes = {}
for n in range(8):
    try:
        nodes[n] = _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % n).read()) & allowed
    except OSError:
        break
print("sysfs node of cuda:0:", device_numa_node(0), "| nodes:", {k: len(v) for k, v in nodes.items()})
d = torch.device("cuda:0")
x = torch.zeros(64, device=d)
for _ in range(2000):
    x.add_(1)
torch.cuda.synchronize()
for rnd in range(3):
    for n, cpus in nodes.items():
        if not cpus:
            continue
        os.sched_setaffinity(0, cpus)
        time.sleep(0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3000):
            x.add_(1)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print("round %d node %d: %.2f us per launch" % (rnd, n, (t1 - t0) / 3000 * 1e6))
os.sched_setaffinity(0, allowed)
