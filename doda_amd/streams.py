"""Side streams that really run beside the main stream.

HIP maps streams onto a small number of hardware queues (4 by default, round robin at creation).  A side stream that
lands on the main stream's queue is serialised with it: measured on MI355X with the rulebook prefetch stream, the
U-Net step takes 7.05 ms instead of 6.2 ms for one stream creation in four (tools/abgate.py QUEUE_SCAN=1) — which one
depends on how many streams the process created before.  independent_stream() creates candidates until one is seen to
make progress while the current stream is busy."""
import time

import torch

_CACHE = {}


def _runs_beside(main, cand, device):
    """True when a tiny kernel on `cand` completes while `main` is still spinning."""
    probe = torch.zeros(64, device=device)
    torch.cuda.synchronize(device)
    with torch.cuda.stream(main):
        torch.cuda._sleep(30_000_000)           # ~12-15 ms of spinning on the main stream's queue
        busy = torch.cuda.Event()
        busy.record(main)
    with torch.cuda.stream(cand):
        probe.add_(1.0)
        done = torch.cuda.Event()
        done.record(cand)
    t0 = time.perf_counter()
    done.synchronize()
    waited = time.perf_counter() - t0
    still_spinning = not busy.query()
    torch.cuda.synchronize(device)
    return still_spinning and waited < 0.008


def independent_stream(device, tries=6, tag="side"):
    """A torch.cuda.Stream on `device` that does not share the current stream's hardware queue (cached per device and
    tag: the calibration costs ~15 ms per candidate).  Falls back to a plain stream when nothing qualifies (one
    hardware queue, or no `torch.cuda._sleep`)."""
    device = torch.device(device)
    import os
    if os.environ.get("DODA_NO_STREAM_CAL") == "1":
        return torch.cuda.Stream(device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    if key in _CACHE:
        return _CACHE[key]
    main = torch.cuda.current_stream(device)
    rejected, chosen = [], None
    try:
        for _ in range(tries):
            cand = torch.cuda.Stream(device=device)
            if _runs_beside(main, cand, device):
                chosen = cand
                break
            rejected.append(cand)      # kept alive until the choice is made: a released queue slot could be handed out again
    except (AttributeError, RuntimeError):
        chosen = None
    if chosen is None:
        chosen = rejected[0] if rejected else torch.cuda.Stream(device=device)
    _CACHE[key] = chosen
    return chosen
