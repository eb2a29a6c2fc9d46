"""Host-side placement of the issuing process.

The U-Net step issues ~540 launches per 6 ms from one thread plus a rulebook helper thread: where those threads run
matters.  Measured on the 2-socket MI355X box (bench.py, 5 processes each): unpinned 6.5-7.5 ms per step, pinned to the
GPU's NUMA node 6.15-6.33 ms (launch doorbells and the pinned staging buffers cross the socket interconnect otherwise;
`OMP_NUM_THREADS` alone does not help).  pin_to_device_numa() restricts the calling thread — call it before any helper
thread exists: threads inherit the mask — to the CPUs of the NUMA node the device hangs off."""
import glob
import os


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_numa_node(device_index=0):
    """NUMA node of a GPU (sysfs), or None when it cannot be determined."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read())
        if node >= 0:
            return node
    except (AttributeError, OSError, ValueError, RuntimeError, ImportError):
        pass
    nodes = set()
    for dev in glob.glob("/sys/class/drm/card*/device"):
        try:
            with open(os.path.join(dev, "vendor")) as f:
                if f.read().strip() != "0x1002":
                    continue
            with open(os.path.join(dev, "numa_node")) as f:
                node = int(f.read())
            if node >= 0:
                nodes.add(node)
        except (OSError, ValueError):
            continue
    return nodes.pop() if len(nodes) == 1 else None     # several GPUs on several nodes: no guess


def pin_to_device_numa(device_index=0):
    """Restrict the calling thread (and every thread it creates afterwards) to the CPUs of the device's NUMA node.
    Returns {"node": n, "cpus": count} or None when nothing was changed (DODA_NO_PIN=1, unknown topology, a mask that
    the launcher already narrowed to other CPUs)."""
    if os.environ.get("DODA_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    node = device_numa_node(device_index)
    if node is None:
        return None
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            cpus = _parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        target = cpus & allowed
        if not target or target == allowed:
            return None if not target else {"node": node, "cpus": len(target)}
        os.sched_setaffinity(0, target)
        return {"node": node, "cpus": len(target)}
    except (OSError, ValueError):
        return None


def raise_issue_priority():
    """Scheduling priority of the calling (kernel-issuing) thread.  A training step here is ~500 dependent launches whose issue
    takes 4-5 ms of host time against ~5.5 ms of GPU time: on a host shared with other jobs every preemption of the issuing
    thread is GPU idle time.  OPT-IN (a benchmark must not boost itself over a shared host's other jobs by default):
    DODA_HOST_PRIO = "fifo" (SCHED_FIFO 10, needs CAP_SYS_NICE) or "nice" (nice -15); unset / "0" / "off": nothing.
    Returns what was applied (a string) or None."""
    mode = os.environ.get("DODA_HOST_PRIO", "off").lower()
    if mode in ("0", "off", "none", ""):
        return None
    try:
        if mode == "fifo" and hasattr(os, "sched_setscheduler"):
            os.sched_setscheduler(0, os.SCHED_FIFO, os.sched_param(10))
            return "SCHED_FIFO 10"
        cur = os.getpriority(os.PRIO_PROCESS, 0)
        if cur > -15:
            os.setpriority(os.PRIO_PROCESS, 0, -15)
        return "nice %d" % os.getpriority(os.PRIO_PROCESS, 0)
    except (OSError, PermissionError, AttributeError):
        return None
