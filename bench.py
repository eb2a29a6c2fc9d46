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
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-voxels", type=int, default=100000, help="size of the CPU-baseline sample scene")
    p.add_argument("--kernel-reps", type=int, default=50)
    p.add_argument("--fp32-steps", type=int, default=40,
                   help="timed steps of the fp32 sub-record (reference precision); 0 = skip")
    p.add_argument("--prefetch", type=int, default=1,
                   help="1: rulebooks of the next batch are built on a helper thread during the step; 0: in line")
    return p.parse_args()


def _subm16_times(idx, shape, nb, dtype, reps):
    """SubMConv3d 16->16 on the level-1 rulebook of `idx`: forward gather, data-grad gather and weight
    gradient, timed with HIP events on the launch stream exactly as the training step issues them:
    weights fragment-packed beforehand (one launch per optimizer step in the model), the weight gradient
    through the multi-layer call with the level's 8 block convolutions in one call (pair lists exported
    once per rulebook for bf16).  Returns the per-launch / per-layer times and the algorithmic bytes
    (B_f = s(M Cin + M Cout) + 4 K Cin Cout + 8 P, B_b = s(2 M Cin + M Cout) + 2*4 K Cin Cout + 8 P)."""
    from doda_amd import ops, spconv
    dev = idx.device
    m = idx.shape[0]
    data = spconv.ops.build_subm(idx, nb, shape, 3)
    pairs_total = int((data.tbl >= 0).sum().item())
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    x = torch.randn(m, 16, device=dev).to(tdt)
    gy = torch.randn(m, 16, device=dev).to(tdt)
    w = torch.randn(27, 16, 16, device=dev) * 0.1
    s = 4 if dtype == "f32" else 2

    def timed(fn, per=1):
        """Average launch time over back-to-back launches, HIP events on the launch stream.  The reps are
        timed in five event-bracketed chunks and the MEDIAN chunk average is returned: one host stall
        (allocator growth, a descheduled issue thread) inside a single bracket once produced a 5x outlier
        for a kernel whose rocprofv3 average in the same run was unchanged."""
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

    b_f = s * (m * 16 + m * 16) + 4 * 27 * 16 * 16 + 8 * pairs_total
    b_b = s * (2 * m * 16 + m * 16) + 2 * 4 * 27 * 16 * 16 + 8 * pairs_total
    plan = ops.PackPlan([(w, 27, 16, 16, 0, s), (w, 27, 16, 16, 2, s)], dev)
    plan.run()
    pk_f, pk_d = plan.outputs
    # bf16: the LDS-staged tile kernel over the rulebook's tilebook (built once per rulebook, as the model does);
    # fp32 is bound by the fp32 matrix rate either way (tile kernel 83.7 us, dense 86.3 us) and stays on conv_fast
    tb = ops.tilebook_build(data.tbl) if (dtype == "bf16" and spconv.ops.TILE_KERNEL) else None
    t_f = timed(lambda: ops.spconv_gather(x, None, data.tbl, m, 0, 16, packed=pk_f, tilebook=tb))
    t_d = timed(lambda: ops.spconv_gather(gy, None, data.tbl, m, 2, 16, packed=pk_d, tilebook=tb))
    n_layers = 8   # the 16 -> 16 block convolutions of level 1 share the rulebook and one multi-layer call
    wg_kernel = "wgrad_multi_kernel (gather table)"
    pairs = None
    if dtype == "bf16" and spconv.functional.WGRAD_PAIRS:
        pairs = data.wgrad_lists()
        wg_kernel = "wgrad_pairs_kernel<1,1> (pair lists)"
    jobs = [(x, gy, data.tbl, m, pairs) if pairs is not None else (x, gy, data.tbl, m) for _ in range(n_layers)]
    t_w = timed(lambda: ops.spconv_wgrad_multi(jobs), per=n_layers)
    t_all = t_f + t_d + t_w
    return {"M": m, "P": pairs_total, "tile_kernel": tb is not None,
            "fwd": {"us": t_f * 1e6, "GBs": b_f / t_f / 1e9},
            "dgrad": {"us": t_d * 1e6, "GBs": b_f / t_d / 1e9},
            "wgrad": {"us": t_w * 1e6, "GBs": b_f / t_w / 1e9, "kernel": wg_kernel,
                      "note": "%d layers per multi-layer call, time per layer incl. the partial reduce" % n_layers},
            "fwd_bwd": {"us": t_all * 1e6, "GBs": (b_f + b_b) / t_all / 1e9,
                        "frac_of_hbm_peak": (b_f + b_b) / t_all / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes": b_f + b_b},
            "b_f": b_f}


def kernel_roofline(batch_dev, dtype, reps, gate_scene=None):
    """Dominant kernel live: SubMConv3d 16->16 forward gather on the batch's level-1 rulebook; plus the
    north-star gate (whole 16->16 fwd+bwd) at the batch size and on ONE ~150 k-voxel scene."""
    idx = batch_dev["voxel_locs"].int()
    nb = int(batch_dev["offsets"].numel() - 1)
    big = _subm16_times(idx, batch_dev["spatial_shape"], nb, dtype, reps)
    out = {"subm16_fwd": big["fwd"], "subm16_dgrad": big["dgrad"], "subm16_wgrad": big["wgrad"],
           "subm16_fwd_bwd": big["fwd_bwd"]}
    if gate_scene is not None:
        one = _subm16_times(gate_scene["voxel_locs"].int(), gate_scene["spatial_shape"], 1, dtype, reps)
        out["gate_150k"] = {"M": one["M"], "P": one["P"], "fwd_us": one["fwd"]["us"], "dgrad_us": one["dgrad"]["us"],
                            "wgrad_us": one["wgrad"]["us"], **one["fwd_bwd"],
                            "note": "north_star gate: SubMConv3d 16->16 fwd+bwd on one ~150k-voxel scene; its 45 MB "
                                    "working set sits in the 256 MB Infinity Cache between back-to-back launches"}
    m, pairs_total, b_f = big["M"], big["P"], big["b_f"]
    kname = "conv_tile (LDS-staged, tilebook)" if big["tile_kernel"] else (
        "conv_fast<PF32,1,2,3>" if dtype == "f32" else "conv_fast<PBF16P,1,2,3>")
    traffic, traffic_src = pmc_traffic(dtype)
    roof = {"kernel": "%s (SubMConv3d 16->16 fwd gather, M=%d, P=%d)" % (kname, m, pairs_total),
            "bound": "hbm", "achieved": out["subm16_fwd"]["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": out["subm16_fwd"]["GBs"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": b_f, "avg_launch_us": out["subm16_fwd"]["us"], "detail": out}
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
    (tools/profile_round.sh: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of the same kernel
    on the same 4 x 150k-voxel batch).  Counters are in KiB; FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads on gfx950 (check: the doubled value,
    86.5 MB, sits 2.5 % above the compulsory x + table bytes, 84.4 MB; WRITE_SIZE equals the output
    bytes exactly).  A counter pass cannot run inside this process, hence the file."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = os.path.join(prof, "r02_pmc_traffic_raw.json")
    if not os.path.exists(path):
        path = os.path.join(prof, "r01_pmc_traffic_raw.json")
    try:
        with open(path) as f:
            d = json.load(f)[dtype]
        fetch, write = d["FETCH_SIZE_KB_mean"], d["WRITE_SIZE_KB_mean"]
        if fetch is None or write is None:
            return None, None
        return (2.0 * fetch + write) * 1024.0, "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE, KiB)" % os.path.basename(path)
    except (OSError, KeyError, ValueError):
        return None, None


def cpu_baseline(args):
    """Oracle U-Net fwd+bwd (fp32, torch-CPU threads = all host cores) on ONE scene."""
    from doda_amd.scene import make_batch
    from oracle.unet_cpu import OracleUNet, forward_backward
    batch = make_batch(1, args.cpu_voxels, 1000, args.voxel_scale)
    torch.manual_seed(0)
    net = OracleUNet().train()
    t0 = time.time()
    forward_backward(net, batch)
    dt = time.time() - t0
    m = batch["voxel_locs"].shape[0]
    return {"value": m / dt, "unit": "voxels/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 scene (%d active voxels), 1 U-Net fwd+bwd incl. serial rulebook build, fp32, "
                      "oracle/unet_cpu.py (spconv CPU path cannot be built: restatement stands in)" % m,
            "seconds": dt}


def main():
    args = parse()
    from doda_amd import dist as ddist
    world, rank, local_rank = ddist.setup()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local_rank %= torch.cuda.device_count()   # (several ranks on one GPU only happens in gloo smoke tests)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from doda_amd.build import build_native
    from doda_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and rank == 0:
        build_native(verbose=False)
    ddist.barrier()
    _lib.lib()
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.scene import make_batch

    batch = make_batch(args.scenes, args.voxels, ddist.seed_for_rank(1000, rank), args.voxel_scale)
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

    def run_training(dtype_name, steps, warmup):
        """`warmup` untimed + `steps` timed training steps with a fresh network; returns
        (max-over-ranks seconds, final loss, network)."""
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
        prefetch = PyramidPrefetcher(dev, len(net.unet.nPlanes)) if args.prefetch else None
        from doda_amd.model import tile_levels_for
        with_tiles = tile_levels_for(fdt)
        pending = [prefetch.submit(batch_dev, with_pairs, with_tiles)] if prefetch else None

        def step():
            opt.zero_grad(set_to_none=True)
            pyramid = None
            if prefetch is not None:
                pyramid = PyramidPrefetcher.take(pending[0], dev)
                pending[0] = prefetch.submit(batch_dev, with_pairs, with_tiles)
            scores = voxelize_and_run(cfg, model, batch_dev, dev, feature_dtype=fdt, inputs_ready=True,
                                      pyramid=pyramid)
            loss = cross_entropy(scores, labels, ignore_index=255)
            loss.backward()
            if reducer is not None:
                reducer.reduce()
            opt.step()
            return loss

        for _ in range(warmup):
            step()
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
        return dt, float(loss.detach()), net

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
            one = make_batch(1, 150000, 1000, args.voxel_scale)
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
                       "grad_sync": "deferred multi-layer wgrad + bucketed all-reduce" if deferred else "torch DDP",
                       "rulebooks": "13 per step, built for the next batch on a helper thread + side stream "
                                    "during the step" if args.prefetch else "13 per step, built in line",
                       "rulebook_parity": "bit-exact vs this repo's restatement of spconv-1.2's CPU algorithm; "
                                          "spconv is not vendored by the reference: orderings unpinned"},
            "roofline": roof,
        }
        if fp32 is not None:
            r32, _ = kernel_roofline(batch_dev, "f32", max(10, args.kernel_reps // 2), gate_scene)
            b32, _ = step_algorithmic_bytes(net, batch_dev, "f32")
            fp32["roofline"] = {"kernel": r32["kernel"], "achieved": r32["achieved"], "frac": r32["frac"],
                                "avg_launch_us": r32["avg_launch_us"], "detail": r32["detail"],
                                "step_frac_of_hbm_peak": b32 / (fp32["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # the fp32 gather is bound by the fp32 matrix rate (157.3 TFLOP/s: 1/16 of bf16, MI355X_MICROARCH.md),
            # not by HBM: its tiles multiply all 27 offsets of every 16-row subtile with a present neighbour
            m32, p32 = r32["detail"]["subm16_fwd"], r32["kernel"]
            n_rows = int(p32.split("M=")[1].split(",")[0])
            pairs = int(p32.split("P=")[1].split(")")[0])
            t32 = r32["avg_launch_us"] * 1e-6
            fp32["roofline"]["mfma_f32"] = {
                "peak_TFLOPs": 157.3, "algorithmic_flops": 2 * pairs * 256, "dense_tile_flops": 2 * n_rows * 27 * 256,
                "frac_algorithmic": 2 * pairs * 256 / t32 / 157.3e12, "frac_dense_tile_upper": 2 * n_rows * 27 * 256 / t32 / 157.3e12,
                "note": "algorithmic = 2 P Cin Cout (present pairs only); dense_tile = every offset of every row, an upper "
                        "bound on what the kernel issues (it skips offsets absent from a whole 32-row wave)"}
            line["fp32"] = fp32
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    ddist.barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
