#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json on synthetic ScanNet-shaped scenes.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one training pass of the SparseConv U-Net over one resident batch on each GPU:
voxel mean-pooling (pointgroup_ops.voxelization) -> SparseConvTensor -> 7-level U-Net forward
(all 13 rulebooks rebuilt, as every real batch differs) -> cross-entropy -> backward (weight + data
gradients of all 71 sparse convs; DDP gradient all-reduce over RCCL when N > 1) -> SGD step.
Inputs are already in HBM when the timed region starts.  value = (sum over ranks of level-1 active
voxels) * K / max-over-ranks time.  Weak scaling: every rank holds its own batch of
`--scenes` scenes (cfgs/scannet BATCH_SIZE_PER_GPU = 4) of ~`--voxels` active voxels.

The JSON line also carries
  roofline     : the dominant kernel (SubM 16->16 gather at level-1 size), timed live with HIP
                 events on the launch stream, against the HBM roofline with ALGORITHMIC bytes
                 (DESIGN.md §4);
  cpu_baseline : the CPU oracle's U-Net (oracle/unet_cpu.py — a port: spconv's CPU path cannot be
                 built here) fwd+bwd on a bounded sample, on this box's host cores (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=30)
    p.add_argument("--dtype", choices=["f32", "bf16"], default="bf16",
                   help="feature storage dtype; bf16 = BASELINE config 2 (fp32 accumulate, fp32 weights)")
    p.add_argument("--scenes", type=int, default=4, help="scenes per GPU (BATCH_SIZE_PER_GPU)")
    p.add_argument("--voxels", type=int, default=150000, help="target active voxels per scene")
    p.add_argument("--voxel-scale", type=int, default=50)
    p.add_argument("--voxel-order", choices=["auto", "morton", "first"], default="auto",
                   help="numbering of a batch's voxels: the reference's first-appearance order, the loader's Z-order renumbering "
                        "(doda_amd.collate.reorder_voxels), or what the loader decides from the batch's tile overflow "
                        "(doda_amd.collate.choose_voxel_order: 'first' at 2 cm, 'morton' at 1 cm)")
    p.add_argument("--config5-steps", type=int, default=12,
                   help="timed steps of the config5 sub-record (BASELINE config 5: 1 cm voxels, ~500 k active voxels per scene, "
                        "1 and 4 scenes; N = 1 only; 0 = skip)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-voxels", type=int, default=0, help="CPU-baseline sample: 0 = the bench batch itself (~10 s of host work at 16 threads), else one scene of this many voxels")
    p.add_argument("--kernel-reps", type=int, default=50)
    p.add_argument("--fp32-steps", type=int, default=40,
                   help="timed steps of the fp32 sub-record (reference precision); 0 = skip")
    p.add_argument("--matrix-head", dest="matrix_head", action="store_true", default=False,
                   help="the head as a [points, classes] score matrix + cross-entropy (round 5) instead of the voxel-level fused head + loss")
    p.add_argument("--refgraph-steps", dest="refgraph_steps", type=int, default=15,
                   help="timed steps of the reference_graph sub-record (the zero-change route: doda_amd.refgraph; 0: skip)")
    p.add_argument("--no-train-entry", action="store_true",
                   help="skip the train_entry sub-record (python -m doda_amd.train on fresh batches, 1 rank and 2 gloo ranks on this GPU)")
    p.add_argument("--train-entry-scenes", type=int, default=640, help="scenes per run of the train_entry sub-record (4 per iteration)")
    p.add_argument("--prefetch", type=int, default=1,
                   help="1: rulebooks of the next batch are built on a helper thread during the step; 0: in line")
    return p.parse_args()


COLD_BYTES = 320 << 20   # operands of consecutive timed launches cycle through more than the 256 MB Infinity Cache


def _timed(fn, reps, per=1):
    """Average launch time over back-to-back launches, HIP events on the launch stream.  The reps are timed in
    five event-bracketed chunks and the MEDIAN chunk average is returned: one host stall (allocator growth, a
    descheduled issue thread) inside a single bracket once produced a 5x outlier for a kernel whose rocprofv3
    average in the same run was unchanged."""
    for _ in range(5):
        fn()
    n_chunk, per_chunk = 5, max(1, reps // 5)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_chunk + 1)]
    torch.cuda.synchronize()
    ev[0].record()  # current stream == the stream ops.* launches on
    for c in range(n_chunk):
        for _ in range(per_chunk):
            fn()
        ev[c + 1].record()
    torch.cuda.synchronize()
    chunks = sorted(ev[c].elapsed_time(ev[c + 1]) for c in range(n_chunk))
    return chunks[n_chunk // 2] * 1e-3 / per_chunk / per


def _subm16_times(idx, shape, nb, dtype, reps):
    """SubMConv3d 16->16 on the level-1 rulebook of `idx`: forward gather, data-grad gather and weight gradient,
    HIP events on the launch stream, issued as the training step issues them (weights fragment-packed beforehand,
    tilebook / pair lists built once per rulebook, the weight gradient through the multi-layer call with the
    level's 8 block convolutions in one call).

    Every kernel is timed twice: WARM (the same buffers back to back: at these sizes the working set sits in the
    256 MB Infinity Cache, which a kernel inside the training step never enjoys) and COLD (operands cycle through
    enough buffer sets to exceed COLD_BYTES).  The forward and data-grad kernels are timed in two forms: PLAIN
    (no epilogue options: the SURVEY 8d gate) and STEP — the instantiation the training step launches: BatchNorm
    statistics in the epilogue plus the fused residual add (forward) or the BatchNorm-backward sums over the
    BatchNorm input (data gradient).  Algorithmic bytes: B_f = s(M Cin + M Cout) + 4 K Cin Cout + 8 P,
    B_b = s(2 M Cin + M Cout) + 2*4 K Cin Cout + 8 P; the STEP forms add the one extra operand they read
    (s M C: residual rows / BatchNorm input rows)."""
    from doda_amd import ops, spconv
    dev = idx.device
    m = idx.shape[0]
    data = spconv.ops.build_subm(idx, nb, shape, 3)
    pairs_total = int((data.tbl >= 0).sum().item())
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    s = 4 if dtype == "f32" else 2
    w = torch.randn(27, 16, 16, device=dev) * 0.1
    b_f = s * (m * 16 + m * 16) + 4 * 27 * 16 * 16 + 8 * pairs_total
    b_b = s * (2 * m * 16 + m * 16) + 2 * 4 * 27 * 16 * 16 + 8 * pairs_total
    b_extra = s * m * 16
    plan = ops.PackPlan([(w, 27, 16, 16, 0, s), (w, 27, 16, 16, 2, s)], dev)
    plan.run()
    pk_f, pk_d = plan.outputs
    from doda_amd.model import tile_levels_for
    use_tile = spconv.ops.TILE_KERNEL and tile_levels_for(tdt) > 0
    use_pairs = dtype == "bf16" and spconv.functional.WGRAD_PAIRS
    # one buffer set = everything one fwd + dgrad + wgrad of a layer touches
    per_set = 3 * s * m * 16 + (56 * m if use_tile else 108 * m)
    n_sets = max(2, -(-COLD_BYTES // per_set) + 1)

    class Set:
        pass
    sets = []
    for j in range(n_sets):
        st = Set()
        st.x = torch.randn(m, 16, device=dev).to(tdt)
        st.gy = torch.randn(m, 16, device=dev).to(tdt)
        st.res = torch.randn(m, 16, device=dev).to(tdt)
        st.y = torch.empty(m, 16, device=dev, dtype=tdt)
        st.tbl = data.tbl if j == 0 else data.tbl.clone()
        st.tb = ops.tilebook_build(st.tbl) if use_tile else None
        st.pairs = None
        if use_pairs:
            st.pairs = data.wgrad_lists() if j == 0 else tuple(
                t.clone() if torch.is_tensor(t) else t for t in data.wgrad_lists())
        sets.append(st)
    mean = torch.zeros(16, device=dev)
    invstd = torch.ones(16, device=dev)
    gamma = torch.ones(16, device=dev)
    beta = torch.zeros(16, device=dev)
    k = [0]

    def nxt(cold):
        if cold:
            k[0] = (k[0] + 1) % n_sets
        return sets[k[0] if cold else 0]

    # prepared calls (ops.GatherCall: one native call per launch, the host cost of the extension's call sites): the Python
    # wrapper around doda_spconv_gather_ex takes longer than the 9 us kernels of one 150k scene on a busy host
    calls = {}
    # the statistics form the step launches: fp64 totals (ABI 9, default) or per-workgroup rows
    _e = spconv.functional._ext
    stats_form = "totals" if (_e is not None and hasattr(_e, "get_stats_totals") and _e.get_stats_totals()) else True

    def call(name, st, step):
        key = (name, id(st), step)
        c = calls.get(key)
        if c is None:
            if name == "fwd":
                c = ops.GatherCall(st.x, None, st.tbl, m, 0, 16, packed=pk_f, tilebook=st.tb, out=st.y,
                                   residual=st.res if step else None, want_stats=stats_form if step else False)
            else:
                c = ops.GatherCall(st.gy, None, st.tbl, m, 2, 16, packed=pk_d, tilebook=st.tb, out=st.y,
                                   want_stats=stats_form if step else False,
                                   bn=(st.x, mean, invstd, gamma, beta, True) if step else None)
            calls[key] = c
            return
        c.run()

    for st_ in sets:                      # (every call exists before anything is timed)
        for name_ in ("fwd", "dgrad"):
            for step_ in (False, True):
                call(name_, st_, step_)

    def fwd(cold, step):
        call("fwd", nxt(cold), step)

    def dgrad(cold, step):
        call("dgrad", nxt(cold), step)

    n_layers = 8   # the 16 -> 16 block convolutions of level 1 share the rulebook and one multi-layer call
    # (a job carries the rulebook's pair lists AND its tilebook, as the model's deferred queue does; the library picks
    # the LDS-staged tile kernel when it is enabled, else the pair-list kernel)
    use_wtile = use_tile and os.environ.get("DODA_NO_WDMA", "0") != "1"

    def job(st, rb=None):
        rb = st if rb is None else rb      # the set whose rulebook (table, pair lists, tilebook) the layer uses
        return (st.x, st.gy, rb.tbl, m, rb.pairs if use_pairs else None, None, rb.tb if use_wtile else None)
    # (prepared calls, ops.WgradPlan: one native call per launch as in the extension's deferred flush — the Python wrapper's
    # ~10 us per job paced this loop on busy hosts: 10.5 instead of 7.5 us per layer on one 150k scene, bimodal between runs)
    plan_warm = ops.WgradPlan([job(sets[0]) for _ in range(n_layers)])
    # cold: as in the step, the layers of one call share ONE rulebook (the deferred flush issues a level's layers together)
    # and differ in their operands; consecutive calls take the next rulebook copy, so nothing of a call is cache-resident
    plans_cold = [ops.WgradPlan([job(sets[(c + j) % n_sets], sets[c]) for j in range(n_layers)]) for c in range(n_sets)]
    kc = [0]

    def wgrad_cold():
        kc[0] = (kc[0] + 1) % n_sets
        plans_cold[kc[0]].run()
    wg_kernel = ("wgrad_tile_f32 (LDS-staged over the tilebook, exact fp32 MFMA)" if (use_wtile and dtype == "f32") else
                 "wgrad_dma16 (LDS-staged over the tilebook)" if use_wtile else
                 "wgrad_pairs_kernel<1,1> (pair lists)" if use_pairs else "wgrad_multi_kernel (gather table)")

    t = {}
    for name, fn in (("fwd", fwd), ("dgrad", dgrad)):
        for cold in (False, True):
            for step in (False, True):
                t[(name, cold, step)] = _timed(lambda: fn(cold, step), reps)
    t_w = {False: _timed(plan_warm.run, max(5, reps // 4), per=n_layers),
           True: _timed(wgrad_cold, max(5, reps // 4), per=n_layers)}

    def rec(sec, nbytes):
        return {"us": sec * 1e6, "GBs": nbytes / sec / 1e9, "frac_of_hbm_peak": nbytes / sec / 1e9 / HBM_PEAK_GBS}

    def kernel(name):
        return {"plain_warm": rec(t[(name, False, False)], b_f), "plain_cold": rec(t[(name, True, False)], b_f),
                "step_warm": rec(t[(name, False, True)], b_f + b_extra), "step_cold": rec(t[(name, True, True)], b_f + b_extra),
                "algorithmic_bytes": {"plain": b_f, "step": b_f + b_extra}}

    def gate(cold):
        tot = t[("fwd", cold, False)] + t[("dgrad", cold, False)] + t_w[cold]
        return {"fwd_us": t[("fwd", cold, False)] * 1e6, "dgrad_us": t[("dgrad", cold, False)] * 1e6,
                "wgrad_us": t_w[cold] * 1e6, **rec(tot, b_f + b_b), "algorithmic_bytes": b_f + b_b}
    return {"M": m, "P": pairs_total, "tile_kernel": use_tile, "buffer_sets": n_sets, "bytes_per_set": per_set,
            "fwd": kernel("fwd"), "dgrad": kernel("dgrad"),
            "wgrad": {"warm": rec(t_w[False], b_f), "cold": rec(t_w[True], b_f), "kernel": wg_kernel,
                      "note": "%d layers per multi-layer call, time per layer incl. the partial reduce" % n_layers},
            "fwd_bwd": {"cold": gate(True), "warm": gate(False)}, "b_f": b_f}


PROFILE_ROUND = "r06"   # in-step / PMC evidence quoted by the bench line must come from THIS round's profiles or be absent
# the roofline kernel's instantiation as rocprofv3 prints it (template arguments up to the ones that name the epilogue),
# shared by the live measurement's label and the look-up in the committed kernel statistics
ROOF_KERNEL = {"bf16": "conv_tile16<false, true>", "f32": "conv_tile<2, true, true", "f32_dense": "::PF32,"}


def in_step_average(dtype):
    """AverageNs of the roofline kernel's in-step instantiation from THIS round's committed rocprofv3 --kernel-trace
    --stats summary of `python bench.py` (profiles/<PROFILE_ROUND>_*_kernel_stats.csv; a counter/trace pass cannot run
    inside this process).  Matches on the instantiation prefix ROOF_KERNEL[dtype] (a new trailing template parameter
    must not silently send the look-up to an older round's file).  Returns (microseconds, file,
    kernel name in the file) or (None, None, None) — never a figure from an earlier round."""
    import csv
    path = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (PROFILE_ROUND, "bf16" if dtype == "bf16" else "f32"))
    from doda_amd.model import tile_levels_for
    want = ROOF_KERNEL["bf16" if dtype == "bf16" else ("f32" if tile_levels_for(torch.float32) > 0 else "f32_dense")]
    try:
        with open(path) as f:
            rows = [r for r in csv.DictReader(f) if want in r["Name"]]
        if rows:   # (several matches: the instantiation with the largest total time = the level-1 layers)
            top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
            return float(top["AverageNs"]) / 1e3, "profiles/" + os.path.basename(path), top["Name"]
    except (OSError, KeyError, ValueError):
        pass
    return None, None, None


def kernel_roofline(batch_dev, dtype, reps, gate_scene=None):
    """Dominant kernel live: the SubMConv3d 16->16 forward gather on the batch's level-1 rulebook IN THE FORM THE
    STEP LAUNCHES IT (BatchNorm statistics + residual add in the epilogue), operands cold; plus the north-star gate
    (whole 16->16 fwd+bwd, plain kernels as SURVEY 8d states it) at the batch size and on ONE ~150 k-voxel scene,
    cold and warm side by side."""
    idx = batch_dev["voxel_locs"].int()
    nb = int(batch_dev["offsets"].numel() - 1)
    big = _subm16_times(idx, batch_dev["spatial_shape"], nb, dtype, reps)
    out = {"subm16_fwd": big["fwd"], "subm16_dgrad": big["dgrad"], "subm16_wgrad": big["wgrad"],
           "subm16_fwd_bwd": {**big["fwd_bwd"]["cold"], "measured": "cold", "warm": big["fwd_bwd"]["warm"]},
           "buffer_sets": big["buffer_sets"], "bytes_per_set": big["bytes_per_set"]}
    if gate_scene is not None:
        one = _subm16_times(gate_scene["voxel_locs"].int(), gate_scene["spatial_shape"], 1, dtype, reps)
        out["gate_150k"] = {"M": one["M"], "P": one["P"], **one["fwd_bwd"]["cold"], "measured": "cold",
                            "warm": one["fwd_bwd"]["warm"], "buffer_sets": one["buffer_sets"],
                            "note": "north_star gate: SubMConv3d 16->16 fwd+bwd on one ~150k-voxel scene; cold = operands "
                                    "cycled through %d buffer sets (> 256 MB), warm = the same 45 MB back to back "
                                    "(Infinity-Cache resident)" % one["buffer_sets"]}
    m, pairs_total = big["M"], big["P"]
    step_cold = big["fwd"]["step_cold"]
    b_step = big["fwd"]["algorithmic_bytes"]["step"]
    rk = ROOF_KERNEL["bf16" if dtype == "bf16" else ("f32" if big["tile_kernel"] else "f32_dense")]
    if big["tile_kernel"]:
        kname = rk + (" (LDS-staged over the tilebook, software-pipelined across a workgroup's tiles;" if dtype == "bf16" else
                      ", ...> (LDS-staged over the tilebook;") + " BatchNorm statistics + residual add in the epilogue: the instantiation the step launches)"
    else:
        kname = ("conv_fast<PF32, ..., STATS> [" + rk + "]" if dtype == "f32" else "conv_fast<PBF16P,1,2,3,STATS>") + " (statistics + residual epilogue)"
    traffic, traffic_src = pmc_traffic(dtype)
    in_step_us, in_step_src, in_step_name = in_step_average(dtype)
    if in_step_us is not None and not (rk in kname and rk in in_step_name):
        in_step_us = in_step_src = in_step_name = None       # the profile describes another kernel than the live figure
    # strict SURVEY 8d bytes of the forward gather (B_f: x + y + weights + 8P; the fused residual operand NOT counted)
    b_8d = big["b_f"]
    roof = {"kernel": "%s, SubMConv3d 16->16 fwd gather, M=%d, P=%d" % (kname, m, pairs_total),
            "M": m, "P": pairs_total, "bound": "hbm", "achieved": step_cold["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": step_cold["frac_of_hbm_peak"],
            "frac_8d": b_8d / (step_cold["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_8d": b_8d,
            "measured": "cold (operands cycled through %d buffer sets, %d MB)" % (
                big["buffer_sets"], big["buffer_sets"] * big["bytes_per_set"] >> 20),
            "frac_cold": step_cold["frac_of_hbm_peak"], "frac_warm": big["fwd"]["step_warm"]["frac_of_hbm_peak"],
            "frac_plain_cold": big["fwd"]["plain_cold"]["frac_of_hbm_peak"],
            "frac_plain_warm": big["fwd"]["plain_warm"]["frac_of_hbm_peak"],
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": b_step, "avg_launch_us": step_cold["us"],
            "in_step_rocprof_avg_us": in_step_us, "in_step_rocprof_source": in_step_src,
            "in_step_rocprof_kernel": in_step_name,
            "in_step_frac": (b_step / (in_step_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if in_step_us else None,
            "in_step_frac_8d": (b_8d / (in_step_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if in_step_us else None,
            "detail": out}
    return roof, pairs_total / max(m, 1)


def step_algorithmic_bytes(net, batch_dev, dtype):
    """Algorithmic bytes of one training step over the 71 sparse convolutions: per layer
    B_f + B_b = s(3 M_in Cin + 2 M_out Cout) + 3 s K Cin Cout + 16 P, with the layer's own row counts and
    pair count P (SubM: non-empty table entries; k2 s2 and its inverse: one pair per fine voxel; 1x1: M).
    BN / ReLU / loss traffic is not counted (fusable).  Returns (bytes, number of layers)."""
    from doda_amd import spconv
    s = 4 if dtype == "f32" else 2
    idx = batch_dev["voxel_locs"].int()
    probe = spconv.SparseConvTensor(None, idx, batch_dev["spatial_shape"], int(batch_dev["offsets"].numel() - 1))
    book = spconv.ops.build_pyramid(probe, len(net.unet.nPlanes))
    total, layers = 0, 0
    first_rows = idx.shape[0]
    for mod in net.modules():
        if not isinstance(mod, spconv.SparseConvolution):
            continue
        K = mod.weight.shape[0] * mod.weight.shape[1] * mod.weight.shape[2]
        cin, cout = mod.in_channels, mod.out_channels
        data = book.get(mod.indice_key)
        if mod.conv1x1:
            # the 1x1 shortcuts carry no indice_key: find the level by the channel count (c = 16 * level)
            lvl = max(1, cout // 16)
            rows = book["subm%d" % lvl].outids.shape[0]
            m_in = m_out = pairs = rows
        elif data is None:       # input conv: level-1 SubM rulebook under its own key
            data = book["subm1"]
            m_in = m_out = first_rows
            pairs = int((data.tbl >= 0).sum())
        elif data.kind == "subm":
            m_in = m_out = data.outids.shape[0]
            pairs = int((data.tbl >= 0).sum())
        else:
            fine, coarse = data.indices.shape[0], data.outids.shape[0]
            m_in, m_out = (coarse, fine) if mod.inverse else (fine, coarse)
            pairs = fine
        total += s * (3 * m_in * cin + 2 * m_out * cout) + 3 * s * K * cin * cout + 16 * pairs
        layers += 1
    return total, layers


def pmc_traffic(dtype):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes
    of THIS round (tools/profile_round5.sh: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of the same kernel — the
    statistics + residual instantiation the step launches — on the same 4 x 150k-voxel batch; no file of this round:
    null).  Counters are in KiB; FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads on gfx950 (check: the doubled value,
    86.5 MB, sits 2.5 % above the compulsory x + table bytes, 84.4 MB; WRITE_SIZE equals the output
    bytes exactly).  A counter pass cannot run inside this process, hence the file."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        d, path = None, None
        for rnd in (PROFILE_ROUND,):   # this round's pass only
            cand = os.path.join(prof, "%s_pmc_traffic_raw.json" % rnd)
            if os.path.exists(cand):
                with open(cand) as f:
                    got = json.load(f).get(dtype)
                if got and got.get("FETCH_SIZE_KB_mean") is not None:
                    d, path = got, cand
                    break
        if d is None:
            return None, None
        fetch, write = d["FETCH_SIZE_KB_mean"], d["WRITE_SIZE_KB_mean"]
        if fetch is None or write is None:
            return None, None
        return (2.0 * fetch + write) * 1024.0, "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE, KiB)" % os.path.basename(path)
    except (OSError, KeyError, ValueError):
        return None, None


def config5_record(args, dev, run_training, ordered, make_batch):
    """BASELINE config 5 (1 cm voxels, ~500 k active voxels per scene: the rulebook + LDS stress case) as a sub-record: the same
    training step, same clock, one scene and four; the roofline of the same kernel on the four-scene batch.  Both arms of the
    voxel numbering are timed on the four-scene batch: the reference's first-appearance order (74 % of the level-1 tiles above
    the list capacity: dense-table kernels) and the loader's choice."""
    import torch
    from doda_amd import spconv
    rec = {"workload": "cfgs/scannet step at voxel scale 100 (1 cm), ~500 k active voxels per scene, bf16 features, synthetic scenes",
           "steps": args.config5_steps}
    warm = max(4, args.config5_steps // 2)
    for scenes in (1, 4):
        raw = make_batch(scenes, 500000, 1000, 100)
        b, order = ordered(raw)
        bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
        spconv.ops._tile_state["skip"] = 0
        dt, loss, _ = run_training("bf16", args.config5_steps, warm, bd)
        m = int(b["voxel_locs"].shape[0])
        r = {"voxels": m, "voxel_order": order, "ms_per_step": dt / args.config5_steps * 1e3, "value": m * args.config5_steps / dt,
             "unit": "voxels/s", "final_loss": loss, "tiles_state": spconv.ops._tile_state["last"]}
        if scenes == 4:
            roof, _ = kernel_roofline(bd, "bf16", max(5, args.kernel_reps // 5), None)
            r["roofline"] = {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_8d", "avg_launch_us", "M", "P")}
            if order != "first":      # the other arm: the reference's numbering on the same scenes
                del bd
                bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in raw.items()}
                spconv.ops._tile_state["skip"] = 0
                dt1, loss1, _ = run_training("bf16", args.config5_steps, warm, bd)
                r["first_appearance_order"] = {"ms_per_step": dt1 / args.config5_steps * 1e3, "value": m * args.config5_steps / dt1,
                                               "final_loss": loss1, "tiles_state": spconv.ops._tile_state["last"]}
                spconv.ops._tile_state["skip"] = 0
        rec["B%d" % scenes] = r
        del bd
        torch.cuda.empty_cache()
    return rec


def voxelize_legs(batch, batch_dev, reps=20):
    """SURVEY 8d CPU-baseline leg (1): point -> voxel maps (`voxelize_idx`, reference voxelize.cpp:61-155 — the
    collate step the reference runs single-threaded in each DataLoader worker) on the bench batch's points.  Host:
    doda_voxelize_idx_h, one thread, and x `workers` threads each voxelising its own copy of the batch (how
    n_workers DataLoader processes scale it).  Device: doda_voxelize_idx_assign + _fill incl. the size read-back,
    against the HBM roofline with the algorithmic bytes of DESIGN.md §3 (36 B per point + 32 + 4(1+maxActive) per
    voxel).  The reference's own compiled voxelize_idx cannot be built in this image (needs sparsehash +
    cuda_runtime_api.h): the host figure is this repository's restatement of it."""
    from concurrent.futures import ThreadPoolExecutor
    from doda_amd import ops
    locs, nb = batch["locs"], int(batch["offsets"].numel() - 1)
    n = locs.shape[0]
    # ONE torch thread for the host legs: the wrapper's three output allocations are torch.zeros calls, and with the
    # process's default intra-op pool (128 threads on the GPU box, a shared host) every one of them was a fork/join
    # of that pool — r03's "1 thread" figure (0.195 s) timed the pool, not doda_voxelize_idx_h (25 ms)
    thr_before = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        vl, _, v2p = ops.voxelize_idx_host(locs, nb, 4)   # warm-up (first-touch page faults of the allocator)
        t1 = float("inf")
        for _ in range(5):
            t0 = time.perf_counter()
            ops.voxelize_idx_host(locs, nb, 4)
            t1 = min(t1, time.perf_counter() - t0)
        workers = max(1, min(8, (os.cpu_count() or 1)))
        copies = [locs.clone() for _ in range(workers)]
        with ThreadPoolExecutor(workers) as ex:   # (the C entry point releases the GIL)
            list(ex.map(lambda c: ops.voxelize_idx_host(c, nb, 4), copies))
            t0 = time.perf_counter()
            list(ex.map(lambda c: ops.voxelize_idx_host(c, nb, 4), copies))
            tw = time.perf_counter() - t0
    finally:
        torch.set_num_threads(thr_before)
    locs_dev = batch_dev["locs"]
    t_dev = _timed(lambda: ops.voxelize_idx_device(locs_dev, nb, 4), reps)
    m, width = vl.shape[0], v2p.shape[1]
    # the two kernels alone: output sizes handed in (a loader that knows an upper bound needs no read-back)
    t_k = _timed(lambda: ops.voxelize_idx_device(locs_dev, nb, 4, sizes=(m, width - 1)), reps)
    nbytes = 36 * n + (32 + 4 * width) * m
    return {"points": n, "voxels": m, "host_1_thread": {"seconds": t1, "points_per_s": n / t1},
            "host_workers": {"workers": workers, "seconds": tw, "points_per_s": workers * n / tw},
            "device": {"us": t_dev * 1e6, "points_per_s": n / t_dev, "algorithmic_bytes": nbytes,
                       "GBs": nbytes / t_dev / 1e9, "frac_of_hbm_peak": nbytes / t_dev / 1e9 / HBM_PEAK_GBS,
                       "note": "assign + fill kernels incl. the D2H read-back of the output sizes",
                       "kernels_only": {"us": t_k * 1e6, "GBs": nbytes / t_k / 1e9,
                                        "frac_of_hbm_peak": nbytes / t_k / 1e9 / HBM_PEAK_GBS,
                                        "note": "output sizes passed in: no read-back between assign and fill"}},
            "kind": "port", "note": "host = doda_voxelize_idx_h (restatement of voxelize.cpp:61-155; the reference's own "
                                    "translation unit does not build here)"}


def cpu_baseline(args, batch=None, batch_dev=None):
    """Oracle U-Net fwd+bwd (fp32, torch-CPU threads = all host cores) on ONE scene; plus the voxelisation leg."""
    from doda_amd.scene import make_batch
    from oracle.unet_cpu import OracleUNet, forward_backward
    # the bench batch itself when it is given (the same 4 x 150k-voxel workload), else one scene of --cpu-voxels
    sample = batch if (batch is not None and args.cpu_voxels <= 0) else make_batch(1, args.cpu_voxels, 1000, args.voxel_scale)
    torch.manual_seed(0)
    net = OracleUNet().train()
    # torch-CPU threads: at most 16 — the port issues ~4000 small matrix products, and with all 128+ host threads of
    # the GPU box their fork/join dominates (measured there: the same 40k-voxel sample in 26 s on one run and 133 s
    # on another with 128 threads)
    n_thr_before = torch.get_num_threads()
    n_thr = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(n_thr)
    try:
        t0 = time.time()
        forward_backward(net, sample)
        dt = time.time() - t0
    finally:
        torch.set_num_threads(n_thr_before)
    m = sample["voxel_locs"].shape[0]
    out = {"value": m / dt, "unit": "voxels/s", "cores": n_thr, "kind": "port",
           "sample": "%d scene(s), %d active voxels, 1 U-Net fwd+bwd incl. serial rulebook build, fp32, "
                     "oracle/unet_cpu.py (spconv CPU path cannot be built: restatement stands in)" % (
                         int(sample["offsets"].numel() - 1), m),
           "seconds": dt}
    if batch is not None:
        out["voxelize_idx"] = voxelize_legs(batch, batch_dev)
    return out


def train_entry(args):
    """The TRAINING ENTRY POINT on fresh batches (north_star states its scaling target on tool/train.py):
    `python -m doda_amd.train` as a subprocess — its own loader (dataset resident in HBM, augmentation + collate + rulebooks
    of batch k+1 on helper threads / side streams while step k is issued: doda_amd/loader.py), meters, LR schedule,
    optimizer — one rank, then two ranks sharing this GPU over gloo (the collective path at world size 2; RCCL cannot put
    two ranks on one device).  Steady-state ms per iteration from the trainer's own clock (device drained on both sides)."""
    import subprocess
    import tempfile
    out = {}
    root = tempfile.mkdtemp(prefix="doda_train_entry_")
    base = ["-m", "doda_amd.train", "--cfg_file", "doda_amd/cfgs/synthetic/spconv.yaml", "--dtype", args.dtype, "--batch_size"]
    tail = ["--synthetic_voxels", str(args.voxels), "--synthetic_base", "8", "--epochs", "1", "--print_freq", "100000",
            "--output_root", root, "--scene_cache", os.path.join(root, "scenes"), "--ckpt_save_freq", "1000",
            "--set", "EVALUATION.evaluate", "False"]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    runs = (("one_rank", [sys.executable] + base + [str(args.scenes), "--synthetic_scenes", str(args.train_entry_scenes),
                          "--timing_json", os.path.join(root, "t1.json")] + tail, env, "t1.json"),
            ("two_ranks_gloo_one_gpu", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                        "--master-addr", "127.0.0.1", "--master-port", "29631"] + base +
             [str(2 * args.scenes), "--synthetic_scenes", str(2 * args.train_entry_scenes), "--launcher", "pytorch",
              "--timing_json", os.path.join(root, "t2.json")] + tail, dict(env, DODA_DIST_BACKEND="gloo"), "t2.json"))
    try:
        for name, cmd, e, js in runs:
            try:
                r = subprocess.run(cmd, cwd=here, env=e, capture_output=True, text=True, timeout=420)
                with open(os.path.join(root, js)) as f:
                    t = json.load(f)
                out[name] = {"ms_per_iter": t["ms_per_iter"], "iterations": t["iterations"], "world": t["world"],
                             "scenes_per_rank": t["batch_size_per_gpu"],
                             "voxels_per_s_whole_job": None,
                             "feeder_ms_per_batch": [t.get("feeder_wait_for_workers_ms"), t.get("feeder_collate_and_rulebooks_ms")]}
            except Exception as ex:   # noqa: BLE001  (a sub-record must not take the bench line with it)
                out[name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
    finally:
        import shutil
        shutil.rmtree(root, ignore_errors=True)
    return out


def main():
    args = parse()
    from doda_amd import dist as ddist
    world, rank, local_rank = ddist.setup()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local_rank %= torch.cuda.device_count()   # (several ranks on one GPU only happens in gloo smoke tests)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the issuing threads next to the GPU (2-socket host: 6.2 ms per step pinned to the GPU's NUMA node, 6.5-7.5 unpinned)
    from doda_amd.host import pin_to_device_numa, raise_issue_priority
    pinned = pin_to_device_numa(local_rank)
    prio = raise_issue_priority()

    from doda_amd.build import build_native
    from doda_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and rank == 0:
        build_native(verbose=False)
    ddist.barrier()
    _lib.lib()
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.scene import make_batch

    from doda_amd.collate import choose_voxel_order, reorder_voxels

    def ordered(b):
        """the batch as the loader would hand it over (doda_amd.loader.DeviceFeeder): (batch, numbering used)"""
        order = choose_voxel_order(b, dev) if args.voxel_order == "auto" else args.voxel_order
        return reorder_voxels(b, order), order

    batch, voxel_order = ordered(make_batch(args.scenes, args.voxels, ddist.seed_for_rank(1000, rank), args.voxel_scale))
    batch_dev = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    m_local = batch["voxel_locs"].shape[0]
    n_local = batch["locs"].shape[0]

    cfg = default_cfg()
    from doda_amd.spconv import functional as Fsp
    from doda_amd.model import PyramidPrefetcher
    from doda_amd.optim import FusedSGD
    from doda_amd import spconv
    deferred = Fsp.set_deferred_wgrad(True)
    labels = batch_dev["labels"]

    gate_info = {}
    sync_info = {}

    def run_training(dtype_name, steps, warmup, batch_dev=batch_dev):
        """`warmup` untimed + `steps` timed training steps with a fresh network; returns
        (max-over-ranks seconds, final loss, network)."""
        labels = batch_dev["labels"]
        torch.manual_seed(0)
        net = SparseConvNet(cfg).to(dev).train()
        # gradients: weight gradients deferred to one multi-layer launch at the end of backward, then (N > 1)
        # bucketed all-reduces over RCCL; torch DDP (the reference's wrapper) when the extension is absent
        model = net if deferred else ddist.wrap_ddp(net, local_rank)
        reducer = ddist.GradAllReduce(net) if deferred else None
        # torch.optim.SGD with its update in one native launch (doda_amd.optim; same state and arithmetic)
        opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        fdt = torch.float32 if dtype_name == "f32" else torch.bfloat16
        # rulebooks ride in the data pipeline: those of the NEXT batch are built on a helper thread + side
        # stream while this step is issued (every step still builds one full pyramid; nothing is cached)
        with_pairs = bool(spconv.functional.WGRAD_PAIRS and fdt == torch.bfloat16)
        prefetch = PyramidPrefetcher(dev, len(net.unet.nPlanes), gated=os.environ.get("DODA_PREFETCH_GATE", "0") == "1") if args.prefetch else None
        from doda_amd.model import tile_levels_for
        with_tiles = tile_levels_for(fdt)
        pending = [prefetch.submit(batch_dev, with_pairs, with_tiles, resident=True, now=True)] if prefetch else None

        def step():
            opt.zero_grad(set_to_none=True)
            if reducer is not None:
                reducer.arm()       # (the exchange of the deep levels' gradients starts inside this step's backward pass)
            pyramid = None
            if prefetch is not None:
                pyramid = PyramidPrefetcher.take(pending[0], dev)
                pending[0] = prefetch.submit(batch_dev, with_pairs, with_tiles, resident=True)
            # head + CrossEntropyLoss at voxel level (reference model/unet.py:62-64,107-108,196; csrc/head.hip): same loss and
            # gradients, no [points, classes] score matrix (--matrix-head: the round-5 form, scores then cross_entropy)
            if args.matrix_head:
                scores = voxelize_and_run(cfg, model, batch_dev, dev, feature_dtype=fdt, inputs_ready=True, pyramid=pyramid)
                loss = cross_entropy(scores, labels, ignore_index=255)
            else:
                loss = voxelize_and_run(cfg, model, batch_dev, dev, feature_dtype=fdt, inputs_ready=True, pyramid=pyramid,
                                        labels=labels, ignore_index=255)
            loss.backward()
            if reducer is not None:
                reducer.reduce()
            opt.step()
            return loss

        for _ in range(warmup):
            step()
        if reducer is not None and reducer.active and warmup > 0:   # (one extra untimed step with event timing: where the exchange starts)
            reducer.time_window = True
            step()
            sync_info[dtype_name] = {"early_start_to_end_of_backward_ms": reducer.overlap_window_ms(),
                                     "early_bucket_MB": sum(b.flat.numel() for b in reducer.buckets) * 4 / 1e6,
                                     "late_bucket_MB": sum(b.flat.numel() for b in reducer.late_buckets) * 4 / 1e6}
            reducer.time_window = False
        ddist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        ddist.barrier()
        dt = time.perf_counter() - t0
        if prefetch is not None:
            pending[0].result()
            prefetch.shutdown()
            if prefetch.gated:   # (A/B aid: steps whose build started without the coarse-phase event)
                gate_info[dtype_name] = {"gate_timeouts": prefetch.gate_timeouts, "builds": steps + warmup}
                sys.stderr.write("[bench] gated prefetch: %d of %d builds ungated (timeouts)\n" % (prefetch.gate_timeouts, steps + warmup))
        if reducer is not None:
            reducer.close()      # (gradient homes are process-wide: the fp32 leg builds its own reducer)
        return dt, float(loss.detach()), net

    def run_reference_graph(dtype_name, steps, warmup):
        """The ZERO-CHANGE route: the module tree and glue a DODA checkout runs over the shims —
        doda_amd.refgraph: plain SparseSequential, `output.features += identity.features`, torch.cat, features[p2v] + nn.Linear,
        rulebooks built inside the convs' first use, nn.CrossEntropyLoss and torch.optim.SGD as tool/train.py builds them —
        on the same batch, timed by the same clock.  Nothing of doda_amd.model's extensions (fused residual, one-call
        residual blocks / coarse levels, fused loss, deferred weight gradients, prefetched rulebooks, one-launch SGD)."""
        from doda_amd.refgraph import RefSparseConvNet, run_reference_route
        Fsp.set_deferred_wgrad(False)
        try:
            torch.manual_seed(0)
            rnet = RefSparseConvNet(cfg).to(dev).train()
            ropt = torch.optim.SGD(rnet.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
            crit = torch.nn.CrossEntropyLoss(ignore_index=255)
            fdt = torch.float32 if dtype_name == "f32" else torch.bfloat16

            def rstep():
                ropt.zero_grad()
                loss = crit(run_reference_route(cfg, rnet, batch_dev, dev, feature_dtype=fdt).float(), batch_dev["labels"])
                loss.backward()
                ropt.step()
                return loss

            for _ in range(warmup):
                rstep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = rstep()
            torch.cuda.synchronize()
            return time.perf_counter() - t0, float(loss.detach())
        finally:
            Fsp.set_deferred_wgrad(deferred)

    elapsed, final_loss, net = run_training(args.dtype, args.steps, args.warmup)
    elapsed, (m_total, n_total) = ddist.reduce_step_stats(elapsed, [m_local, n_local], dev)
    # the reference computes in fp32 end to end (lib/pointgroup_ops/src/cuda.cu:11-13): the same step in
    # fp32 rides along as a sub-record, timed by the same clock (N = 1 only, fewer steps)
    fp32 = None
    if world == 1 and args.dtype != "f32" and args.fp32_steps > 0:
        e32, l32, _ = run_training("f32", args.fp32_steps, max(5, args.warmup // 3))
        fp32 = {"ms_per_step": e32 / args.fp32_steps * 1e3, "value": m_local * args.fp32_steps / e32,
                "unit": "voxels/s", "steps": args.fp32_steps, "final_loss": l32}

    if rank == 0:
        gate_scene = None
        if world == 1:   # the north-star gate is stated on ONE ~150k-voxel scene
            one = reorder_voxels(make_batch(1, 150000, 1000, args.voxel_scale), voxel_order)
            gate_scene = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in one.items()}
        roof, ppv = kernel_roofline(batch_dev, args.dtype, args.kernel_reps, gate_scene)
        step_bytes, n_layers = step_algorithmic_bytes(net, batch_dev, args.dtype)
        step_gbs = step_bytes / (elapsed / args.steps) / 1e9
        roof["step"] = {"algorithmic_bytes": step_bytes, "conv_layers": n_layers, "GBs": step_gbs,
                        "frac_of_hbm_peak": step_gbs / HBM_PEAK_GBS,
                        "note": "all 71 sparse convs fwd+bwd (SURVEY 8d formula) over the whole step time, which "
                                "also contains BN, rulebooks, loss and optimizer"}
        line = {
            "metric": "active-voxels/sec fwd+bwd SparseConv U-Net, ScanNet 2cm",
            "value": m_total * args.steps / elapsed, "unit": "voxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cfgs/scannet SparseConv U-Net training step (voxel pooling + fwd "
                                   "+ CE + bwd + SGD), %g cm voxels, %d scenes/GPU x ~%d active voxels, "
                                   "random-init weights" % (100.0 / args.voxel_scale, args.scenes, args.voxels),
                       "global_batch": args.scenes * world, "voxels_per_gpu": m_local,
                       "points_per_gpu": n_local, "pairs_per_voxel_subm1": round(ppv, 2),
                       "parallelism": "dp%d" % world, "n_classes": 20, "final_loss": final_loss,
                       "voxel_order": ("Z-order renumbering by the loader (doda_amd.collate.reorder_voxels; per-point outputs unchanged)"
                                       if voxel_order == "morton" else "first appearance (the reference's numbering)")
                                      + (" [chosen from the batch's tile overflow]" if args.voxel_order == "auto" else ""),
                       "host_pinning": ("NUMA node %d (%d CPUs)" % (pinned["node"], pinned["cpus"])) if pinned else "none",
                       "host_priority": prio or "unchanged",
                       "head": ("[points, classes] score matrix + fused cross-entropy" if args.matrix_head else
                                "Linear head + CrossEntropyLoss at voxel level (csrc/head.hip): no score matrix; same loss and gradients"),
                       "grad_sync": "deferred multi-layer wgrad + bucketed all-reduce" if deferred else "torch DDP",
                       "collectives": ("none (single process)" if not dist.is_initialized() else
                                       "%s (forced, 1 rank)" % dist.get_backend() if world == 1 else
                                       "%s, %d ranks" % (dist.get_backend(), world)),
                       "rulebooks": "13 per step, built for the next batch on a helper thread + side stream "
                                    "during the step" if args.prefetch else "13 per step, built in line",
                       "prefetch_gate": gate_info.get(args.dtype, "off"),
                       "grad_exchange": sync_info.get(args.dtype, "no process group"),
                       "rulebook_parity": "bit-exact vs this repo's restatement of spconv-1.2's CPU algorithm; "
                                          "spconv is not vendored by the reference: orderings unpinned"},
            "roofline": roof,
        }
        if world == 1 and args.refgraph_steps > 0:
            er, lr_ = run_reference_graph(args.dtype, args.refgraph_steps, max(3, args.warmup // 3))
            line["reference_graph"] = {
                "ms_per_step": er / args.refgraph_steps * 1e3, "value": m_local * args.refgraph_steps / er, "unit": "voxels/s",
                "steps": args.refgraph_steps, "final_loss": lr_, "vs_headline_step": (er / args.refgraph_steps) / (elapsed / args.steps),
                "note": "doda_amd.refgraph: the reference's module tree and call pattern (model/unet.py:15-99, model/unet_block.py:9-100) "
                        "over the drop-in spconv / pointgroup_ops surface, torch CrossEntropyLoss + torch.optim.SGD — what a DODA "
                        "checkout runs with zero changes; the headline runs doda_amd.model.SparseConvNet (same parameters, same "
                        "graph) with the fused residual, one-call blocks / coarse levels, deferred weight gradients, prefetched "
                        "rulebooks, fused loss and one-launch SGD"}
        if fp32 is not None:
            line["fp32_ms_per_step"] = fp32["ms_per_step"]      # (top level too: the driver's parser keeps flat keys)
            r32, _ = kernel_roofline(batch_dev, "f32", max(10, args.kernel_reps // 2), gate_scene)
            b32, _ = step_algorithmic_bytes(net, batch_dev, "f32")
            fp32["roofline"] = {"kernel": r32["kernel"], "achieved": r32["achieved"], "frac": r32["frac"],
                                "measured": r32["measured"], "avg_launch_us": r32["avg_launch_us"], "detail": r32["detail"],
                                "step_frac_of_hbm_peak": b32 / (fp32["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # the fp32 gather is bound by the fp32 matrix rate (157.3 TFLOP/s: 1/16 of bf16, MI355X_MICROARCH.md),
            # not by HBM: its tiles multiply all 27 offsets of every 16-row subtile with a present neighbour
            n_rows, pairs = r32["M"], r32["P"]
            t32 = r32["detail"]["subm16_fwd"]["plain_cold"]["us"] * 1e-6
            fp32["roofline"]["mfma_f32"] = {
                "peak_TFLOPs": 157.3, "algorithmic_flops": 2 * pairs * 256, "dense_tile_flops": 2 * n_rows * 27 * 256,
                "frac_algorithmic": 2 * pairs * 256 / t32 / 157.3e12, "frac_dense_tile_upper": 2 * n_rows * 27 * 256 / t32 / 157.3e12,
                "note": "algorithmic = 2 P Cin Cout (present pairs only); dense_tile = every offset of every row, an upper "
                        "bound on what the kernel issues (it skips offsets absent from a whole 32-row wave)"}
            line["fp32"] = fp32
        if world == 1 and args.config5_steps > 0 and args.voxel_scale != 100:
            line["config5"] = config5_record(args, dev, run_training, ordered, make_batch)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, batch, batch_dev)
            roof["detail"]["voxelize_idx_device"] = line["cpu_baseline"]["voxelize_idx"]["device"]
        if world == 1 and not args.no_train_entry and not dist.is_initialized():
            del batch_dev, net           # (the subprocesses share this GPU)
            torch.cuda.empty_cache()
            te = train_entry(args)
            for rec in te.values():
                if "ms_per_iter" in rec:
                    rec["voxels_per_s_whole_job"] = m_local / args.scenes * rec["scenes_per_rank"] * rec["world"] / (rec["ms_per_iter"] * 1e-3)
                    rec["vs_resident_batch_step"] = rec["ms_per_iter"] / (elapsed / args.steps * 1e3)
            line["train_entry"] = te
        print(json.dumps(line), flush=True)
    ddist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
