"""In-process A/B of the rulebook prefetch gate: the bench step with the next pyramid built (a) from the start of the
step, (b) from the step's coarse phase on (PyramidPrefetcher(gated=True)), (c) not at all (pyramid re-used: the floor).
Alternating blocks of `steps` steps, several rounds: box-to-box and run-to-run noise (+-0.5 ms) cancels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = torch.device("cuda:0")
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, 150000, 1000).items()}
wp, wt = True, tile_levels_for(torch.bfloat16)


def run(mode):
    if mode.startswith("inline"):
        return run_inline(mode)
    if mode.startswith("dummy"):
        _, n, big = mode.split("_")
        return run_dummy(int(n), int(big))
    pf = PyramidPrefetcher(d, 7, gated=(mode == "gated")) if mode != "reuse" else PyramidPrefetcher(d, 7, gated=False)
    pend = [pf.submit(bd, wp, wt, resident=True, now=True)]
    fixed = PyramidPrefetcher.take(pf.submit(bd, wp, wt, resident=True, now=True), d) if mode == "reuse" else None

    acc = [0.0, 0.0, 0.0]

    def step():
        opt.zero_grad(set_to_none=True)
        if mode == "reuse":
            pyr = fixed
        else:
            a = time.perf_counter()
            pend[0].result()
            b = time.perf_counter()
            pyr = PyramidPrefetcher.take(pend[0], d)
            c = time.perf_counter()
            pend[0] = pf.submit(bd, wp, wt, resident=True)
            e = time.perf_counter()
            acc[0] += b - a; acc[1] += c - b; acc[2] += e - c
        loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
        loss.backward()
        opt.step()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    pend[0].result()
    pf.shutdown()
    if mode != "reuse":
        print("   %s: per step wait-for-build %.3f ms, take %.3f ms, submit %.3f ms" % (mode, acc[0] / (steps + 10) * 1e3, acc[1] / (steps + 10) * 1e3, acc[2] / (steps + 10) * 1e3))
    return dt


def run_dummy(n_launch, big):
    """reuse mode + n_launch tiny (or `big`-element) fill kernels on an independent side stream per step: what do side-stream
    LAUNCHES cost the step, apart from their work?"""
    from doda_amd.streams import independent_stream
    side = independent_stream(d, tag="rulebooks")
    pf = PyramidPrefetcher(d, 7, gated=False)
    fixed = PyramidPrefetcher.take(pf.submit(bd, wp, wt, resident=True, now=True), d)
    buf = torch.empty(max(big, 64), dtype=torch.int32, device=d)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.cuda.stream(side):
            for _ in range(n_launch):
                buf.fill_(1)
        loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=fixed), bd["labels"])
        loss.backward()
        opt.step()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    pf.shutdown()
    return (time.perf_counter() - t0) / steps * 1e3


def run_inline(mode):
    """The NEXT batch's pyramid built by the ISSUING thread itself on the side stream — no second thread in the HIP runtime;
    inline_start: at the top of the step (its six size read-backs block the issuing thread while the main stream still has
    the previous step's tail queued), inline_mid: between forward and backward."""
    from doda_amd.streams import independent_stream
    side = independent_stream(d, tag="rulebooks")

    def build():
        with torch.cuda.stream(side):
            idx32 = bd["voxel_locs"].int()
            probe = spconv.SparseConvTensor(None, idx32, bd["spatial_shape"], 4)
            spconv.ops.build_pyramid(probe, 7, with_pairs=wp, with_tiles=wt)
            ev = torch.cuda.Event(); ev.record(side)
        return idx32, probe.indice_dict, ev
    nxt = [build()]

    def step():
        opt.zero_grad(set_to_none=True)
        idx32, book, ev = nxt[0]
        main = torch.cuda.current_stream(d)
        main.wait_event(ev)
        idx32.record_stream(main)
        for data in book.values():
            for t in vars(data).values():
                for u in (t if isinstance(t, tuple) else (t,)):
                    if torch.is_tensor(u) and u.is_cuda:
                        u.record_stream(main)
        if mode == "inline_start":
            nxt[0] = build()
        loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=(idx32, book)), bd["labels"])
        if mode == "inline_mid":
            nxt[0] = build()
        loss.backward()
        opt.step()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


MODES = ("start", "inline_start", "inline_mid", "reuse") if os.environ.get("ABGATE_INLINE") == "1" else ("start", "gated", "reuse")
if os.environ.get("ABGATE_DUMMY") == "1":
    MODES = ("reuse", "dummy_90_64", "dummy_45_64", "dummy_10_25000000", "start")
res = {m: [] for m in MODES}
for r in range(rounds):
    for mode in MODES:
        res[mode].append(run(mode))
for k, v in res.items():
    print(k, " ".join("%.2f" % x for x in v), "| median %.2f" % sorted(v)[len(v) // 2])

if os.environ.get("QUEUE_SCAN") == "1":
    keep = []
    for k in range(8):
        import doda_amd.streams as _st
        _st._CACHE.clear()          # a fresh calibration per run
        print("streams created before this run:", len(keep), "-> start %.2f ms" % run("start"), flush=True)
        keep.append(torch.cuda.Stream(device=d))
