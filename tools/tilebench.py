"""Tile kernel (conv_tile over a tilebook) against the dense-table kernel (conv_fast) on the level-1
rulebook of the benchmark batch: tilebook build time, forward / data-grad times, HIP events on the launch
stream.  python tools/tilebench.py [--scenes 4] [--voxels 150000] [--reps 50]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--voxels", type=int, default=150000)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--nc", type=int, default=16)
    ap.add_argument("--kc", type=int, default=16)
    ap.add_argument("--level", type=int, default=1, help="1: finest rulebook, 2: the rulebook after one k2s2 downsampling")
    a = ap.parse_args()
    from doda_amd import ops, spconv
    from doda_amd.scene import make_batch
    d = torch.device("cuda:0")
    b = make_batch(a.scenes, a.voxels, 1000)
    idx = b["voxel_locs"].int().to(d)
    shape = b["spatial_shape"]
    for _ in range(a.level - 1):
        down = spconv.ops.build_down2(idx, a.scenes, shape, 2, 2, 0, 1)
        idx, shape = down.outids, down.out_spatial_shape
    data = spconv.ops.build_subm(idx, a.scenes, shape, 3)
    m = idx.shape[0]
    x = torch.randn(m, a.kc, device=d).bfloat16()
    w = torch.randn(27, a.kc, a.nc, device=d) * 0.1
    plan = ops.PackPlan([(w, 27, a.kc, a.nc, 0, 2)], d)
    plan.run()
    pk = plan.outputs[0]

    def timed(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.reps

    tb = ops.tilebook_build(data.tbl)
    out = {"M": m, "P": int((data.tbl >= 0).sum().item()),
           "tilebook_build_us": timed(lambda: ops.tilebook_build(data.tbl)),
           "tilebook_bytes": tb.numel(), "table_bytes": data.tbl.numel() * 4,
           "dense_us": timed(lambda: ops.spconv_gather(x, None, data.tbl, m, 0, a.nc, packed=pk)),
           "tile_us": timed(lambda: ops.spconv_gather(x, None, data.tbl, m, 0, a.nc, packed=pk, tilebook=tb))}
    if a.nc == 16 and a.kc == 16:
        gy = torch.randn(m, 16, device=d).bfloat16()
        plan2 = ops.PackPlan([(w, 27, 16, 16, 2, 2)], d)
        plan2.run()
        out["dgrad_tile_us"] = timed(lambda: ops.spconv_gather(gy, None, data.tbl, m, 2, 16, packed=plan2.outputs[0], tilebook=tb))
        pairs = data.wgrad_lists()
        jobs = [(x, gy, data.tbl, m, pairs)] * 8
        out["wgrad_pairs_us_per_layer"] = timed(lambda: ops.spconv_wgrad_multi(jobs)) / 8
    # cold variant: cycle through inputs / tables so that consecutive launches do not find their operands in the
    # 256 MB Infinity Cache (what a kernel meets inside the training step)
    xs = [torch.randn(m, a.kc, device=d).bfloat16() for _ in range(6)]
    tbls = [data.tbl.clone() for _ in range(6)]
    tbs = [ops.tilebook_build(t) for t in tbls]
    k = [0]

    def cold(tiled):
        j = k[0] = (k[0] + 1) % 6
        ops.spconv_gather(xs[j], None, tbls[j], m, 0, a.nc, packed=pk, tilebook=tbs[j] if tiled else None)
    out["dense_cold_us"] = timed(lambda: cold(False))
    out["tile_cold_us"] = timed(lambda: cold(True))
    from doda_amd._ext import ext
    if ext is not None:
        wt = torch.nn.Parameter(w.view(3, 3, 3, a.kc, a.nc).clone())
        res = torch.randn(m, a.nc, device=d).bfloat16()
        tt = [ext.with_tilebook(t) for t in tbls]

        def stats(tiled):
            j = k[0] = (k[0] + 1) % 6
            with torch.no_grad():
                ext.indice_conv_stats(xs[j], wt, tt[j] if tiled else tbls[j], tt[j] if tiled else tbls[j], m, 2, pk, None, res)
        out["dense_stats_res_cold_us"] = timed(lambda: stats(False))
        out["tile_stats_res_cold_us"] = timed(lambda: stats(True))
    if a.kc == 48 and a.nc == 48:
        from doda_amd._lib import lib as _l
        _l().doda_set_option(2, 0)
        out["stream_weights_us"] = timed(lambda: ops.spconv_gather(x, None, data.tbl, m, 0, 48, packed=pk))
        out["stream_weights_cold_us"] = timed(lambda: cold(False))
        _l().doda_set_option(2, 1)
        out["weights_in_lds_us"] = timed(lambda: ops.spconv_gather(x, None, data.tbl, m, 0, 48, packed=pk))
        out["weights_in_lds_cold_us"] = timed(lambda: cold(False))
    y0 = ops.spconv_gather(x, None, data.tbl, m, 0, a.nc, packed=pk, out_f32=True)
    y1 = ops.spconv_gather(x, None, data.tbl, m, 0, a.nc, packed=pk, out_f32=True, tilebook=tb)
    out["max_rel_diff"] = ((y0 - y1).abs().max() / y0.abs().max()).item()
    nt = (m + 255) // 256
    from doda_amd._lib import lib
    uc = tb[nt * (lib().doda_tilebook_umax() * 4 + 27 * 512):].view(torch.int32)[:nt]
    out["distinct_rows_per_tile"] = {"mean": uc.float().mean().item(), "max": int(uc.max().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
