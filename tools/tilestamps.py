#!/usr/bin/env python
"""Where a tile's time goes inside conv_tile: wall-clock stamps (100 MHz) from wave 0 of every persistent workgroup.

Builds ITS OWN copy of the library (csrc/spconv_tile.hip recompiled with -DDODA_TILE_STAMPS, the other objects reused)
under tools/_stamps/, loads it through doda_amd._lib, launches the step-form level-1 layer (statistics + residual) on
rotating buffer sets (cold) and prints, per round of tiles, the mean / p90 length of every phase:
  rows    stamp 0 -> 1   row / strip loads issued .. rows parked in LDS
  bar1    1 -> 2         barrier before the multiply phase
  units   2 -> 3         the unit loop
  epi     3 -> 4         epilogue
  bar2    4 -> 5         barrier at the end of the tile
usage: tilestamps.py [n_scenes=4] [reps=6] [l2]      (l2: the level-2 rulebook of the batch, 32 -> 32 channels: conv_tile<1>)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_stamps")


def build():
    from doda_amd import build as B
    lib = os.path.join(OUT, "libdoda_hip.so")
    src = os.path.join(B.HERE, "csrc", "spconv_tile.hip")
    if os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(src) and "--build" not in sys.argv:
        return lib            # built in the container; the .so travels to the GPU box, the objects do not
    B.build_native(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "spconv_tile.o")
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-DDODA_TILE_STAMPS", "-c", src, "-o", obj])
    objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith(".o") and f != "spconv_tile.o"]
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj, *objs, "-o", lib])
    return lib


def main():
    import ctypes as C
    import numpy as np
    import torch
    lib_path = build()
    from doda_amd import _lib
    _lib.LIB_PATH = lib_path
    from doda_amd import ops, spconv
    from doda_amd.scene import make_batch
    nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    batch = make_batch(nsc, 150000, 1000)
    idx = batch["voxel_locs"].int().to(dev)
    shape = [int(s) for s in batch["spatial_shape"]]
    l2 = "l2" in sys.argv[3:]
    c = 32 if l2 else 16
    if l2:
        t = spconv.SparseConvTensor(None, idx, shape, nsc)
        sub = spconv.ops.build_pyramid(t, 3, with_pairs=False, with_tiles=2)["subm2"]
    else:
        sub = spconv.ops.build_subm(idx, nsc, shape, 3)
    m = sub.tbl.shape[1]
    w = torch.randn(27, c, c, device=dev) * 0.05
    plan = ops.PackPlan([(w, 27, c, c, 0, 2)], dev)
    plan.run()
    n_sets = 8
    sets = []
    for j in range(n_sets):
        tbl = sub.tbl.clone()
        sets.append((torch.randn(m, c, device=dev).bfloat16(), torch.randn(m, c, device=dev).bfloat16(), tbl,
                     ops.tilebook_build(tbl)))
    h = _lib.lib()
    fn = C.CDLL(lib_path).doda_debug_tile_stamps   # (same path: the handle doda_amd._lib holds)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t]
    buf = np.zeros(768 * 8 * 8, dtype=np.uint64)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps):
        x, res, tbl, tb = sets[r % n_sets]
        torch.cuda.synchronize()
        ev[0].record()
        ops.spconv_gather(x, None, tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb, residual=res, want_stats=True)
        ev[1].record()
        torch.cuda.synchronize()
        if r < reps - 2:
            continue
        assert fn(buf.ctypes.data, buf.nbytes) == 0
        st = buf.reshape(768, 8, 8).astype(np.int64)
        nt = (m + 255) // 256
        groups = min(512 if l2 else 768, (nt + 7) // 8 * 8)
        if not l2 and h.doda_get_option(1) and nt >= int(os.environ.get("DODA_TILE16_MIN_TILES", "769")):
            groups = 512                                  # conv_tile16
        st = st[:groups]
        st = st[st[:, 0, 0] > 0]                          # (workgroups without a tile leave no stamps)
        t0 = st[:, 0, 0].min()
        print("rep %d: m %d tiles %d groups %d  event time %.1f us  stamp span %.1f us  start skew p50 %.2f p99 %.2f us" % (
            r, m, nt, groups, ev[0].elapsed_time(ev[1]) * 1e3, (st[:, :, 5].max() - t0) / 100.0,
            np.percentile(st[:, 0, 0] - t0, 50) / 100.0, np.percentile(st[:, 0, 0] - t0, 99) / 100.0))
        for it in range(8):
            have = st[:, it, 5] > st[:, it, 0]
            have &= st[:, it, 0] >= t0
            if it > 0:
                have &= st[:, it, 0] >= st[:, it - 1, 5]
            if not have.any():
                continue
            s = st[have, it, :6]
            d = np.diff(s, axis=1) / 100.0
            names = ("rows", "bar1", "units", "epi", "bar2")
            print("  round %d: %4d workgroups  start %6.2f us (p90 %6.2f)  " % (
                it, int(have.sum()), (s[:, 0] - t0).mean() / 100.0, np.percentile(s[:, 0] - t0, 90) / 100.0) + "  ".join(
                "%s %5.2f/%5.2f" % (n, d[:, k].mean(), np.percentile(d[:, k], 90)) for k, n in enumerate(names))
                + "   tile %5.2f" % ((s[:, 5] - s[:, 0]).mean() / 100.0))
        # the buffer keeps old stamps of rounds a workgroup does not run this time: zero it for the next read
        buf[:] = 0


if __name__ == "__main__":
    main()
