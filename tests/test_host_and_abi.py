"""CPU tests of the product side that need no GPU: the C-ABI library builds, loads and exports
every symbol include/doda_hip.h declares; the host voxeliser (fork-safe CPU entry point) matches the
oracle; argument errors are reported, not crashed on; the Python surface refuses CPU tensors."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(native_lib):
    from doda_amd import _lib
    header = open(os.path.join(ROOT, "include", "doda_hip.h")).read()
    declared = set(re.findall(r"\b(doda_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(native_lib, name), name
    assert native_lib.doda_abi_version() == 12
    assert len(declared) <= 70      # (the boundary a maintainer carries; ABI 8: four executor entry points, ABI 9: BatchNorm over totals, ABI 11: doda_layers_run)
    assert native_lib.doda_strerror(-3).decode().startswith("cell id")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "doda_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "liboracle" not in text, f


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_voxelize_idx_host_matches_oracle(native_lib, oracle, mode):
    from doda_amd import ops
    rng = np.random.default_rng(mode)
    if mode == 0:
        from tests.util import random_voxels
        coords = random_voxels(3, 4000, 3, [20, 20, 20]).astype(np.int64)
    else:
        coords = np.concatenate([np.sort(rng.integers(0, 3, (30000, 1)), 0),
                                 rng.integers(0, 25, (30000, 3))], 1).astype(np.int64)
    ref = oracle.voxelize_idx(coords, mode)
    got = ops.voxelize_idx_host(torch.from_numpy(coords), 3, mode)
    for r, g in zip(ref, got):
        assert np.array_equal(r, g.numpy())


def test_voxelize_idx_host_errors_and_wrapper(native_lib):
    from doda_amd import ops, pointgroup_ops
    from doda_amd._lib import DodaNativeError
    dup = torch.tensor([[0, 1, 1, 1], [0, 1, 1, 1]], dtype=torch.int64)
    with pytest.raises(DodaNativeError):
        ops.voxelize_idx_host(dup, 1, 0)          # mode 0 asserts uniqueness in the reference
    with pytest.raises(RuntimeError):
        ops.voxelize_idx_host(dup.int(), 1, 4)    # wrong dtype
    oc, im, om = pointgroup_ops.voxelization_idx(dup, 1, 4)
    assert oc.tolist() == [[0, 1, 1, 1]] and im.tolist() == [0, 0] and om.tolist() == [[2, 0, 1]]
    assert im.dtype == torch.int32 and om.dtype == torch.int32 and oc.dtype == torch.int64


def test_voxelize_idx_host_is_fork_safe(native_lib):
    """DODA calls voxelization_idx inside forked DataLoader workers (dataset/dataset.py:182)."""
    import multiprocessing as mp
    from doda_amd import ops
    coords = torch.tensor([[0, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1]], dtype=torch.int64)
    ops.voxelize_idx_host(coords, 1, 4)  # library loaded in the parent first
    ctx = mp.get_context("fork")
    q = ctx.Queue()

    def child():
        q.put(ops.voxelize_idx_host(coords, 1, 4)[1].tolist())
    p = ctx.Process(target=child)
    p.start()
    p.join(30)
    assert p.exitcode == 0 and q.get(timeout=5) == [0, 1, 0]


def test_native_ops_refuse_cpu_tensors(native_lib):
    from doda_amd import spconv
    conv = spconv.SubMConv3d(3, 16, 3, padding=1, bias=False, indice_key="a")
    t = spconv.SparseConvTensor(torch.zeros(4, 3), torch.zeros((4, 4), dtype=torch.int32), [8, 8, 8], 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        conv(t)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from doda_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "absent.so"))
    with pytest.raises(_lib.DodaNativeError, match="not built"):
        _lib.lib()


def test_scene_generator_statistics(native_lib, oracle):
    from doda_amd.scene import make_batch
    b = make_batch(2, 8000, 77)
    m = b["voxel_locs"].shape[0]
    assert abs(m - 16000) < 0.04 * 16000
    assert 1.1 < b["locs"].shape[0] / m < 1.6
    assert (b["spatial_shape"] >= 128).all()
    idx = b["voxel_locs"].int().numpy()
    pairs, pn = oracle.indice_pairs_subm(idx, 2, [int(s) for s in b["spatial_shape"]], 3)
    assert 7.0 < pn.sum() / m < 13.0          # ScanNet-like 27-neighbourhood occupancy
    b2 = make_batch(2, 8000, 77)
    assert torch.equal(b["locs"], b2["locs"])  # deterministic


def test_spconv_surface_and_state_dict_layout():
    import json
    from doda_amd import model, spconv
    from doda_amd.spconv.modules import SparseModule
    net = model.SparseConvNet(model.default_cfg())
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "unet_state_keys.json")))
    assert keys == gold   # gold = state_dict of the reference's own SparseConvNet
    assert net.input_conv[0].weight.shape == (3, 3, 3, 3, 16)
    assert isinstance(net.unet.blocks, spconv.SparseSequential) and issubclass(model.ResidualBlock, SparseModule)
    assert all("BatchNorm" not in type(m).__name__ for m in net.modules() if isinstance(m, spconv.SparseConvolution))
    with pytest.raises(NotImplementedError):   # kernel volume > 27 has no native path
        spconv.SparseConv3d(4, 4, kernel_size=4, stride=2)(spconv.SparseConvTensor(
            torch.zeros(1, 4), torch.zeros((1, 4), dtype=torch.int32), [8, 8, 8], 1))
    with pytest.raises(RuntimeError):          # and there is no CPU fallback for the supported ones
        spconv.SparseConv3d(4, 4, kernel_size=3, stride=2)(spconv.SparseConvTensor(
            torch.zeros(1, 4), torch.zeros((1, 4), dtype=torch.int32), [8, 8, 8], 1))


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="reference checkout not present")
def test_reference_model_files_run_on_doda_spconv_surface():
    """The reference's model/unet.py + unet_block.py import and construct unchanged on top of the
    drop-in shims (doda_amd/shims on sys.path), and produce the same parameter layout."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0]=['%s/doda_amd/shims','/root/reference','%s'];"
        "import spconv, PG_OP, pointops2_cuda; import model.unet as ru;"
        "from doda_amd.model import default_cfg, SparseConvNet;"
        "a=ru.SparseConvNet(default_cfg()); b=SparseConvNet(default_cfg());"
        "ka={k:tuple(v.shape) for k,v in a.state_dict().items()};"
        "kb={k:tuple(v.shape) for k,v in b.state_dict().items()};"
        "assert ka==kb and type(a.input_conv).__module__.startswith('doda_amd'); print('OK')" % (ROOT, ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_numa_pinning_helper_is_safe_without_a_gpu():
    """doda_amd.host: cpulist parsing, and pin_to_device_numa() is a no-op (None) when the topology is unknown."""
    from doda_amd import host
    assert host._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    import os
    before = os.sched_getaffinity(0)
    res = host.pin_to_device_numa(0)
    assert res is None or (isinstance(res, dict) and res["cpus"] >= 1)
    assert os.sched_getaffinity(0) <= before


def test_every_environment_switch_is_documented():
    """dozens of DODA_* switches, the default path one point among them.  INTEGRATION.md §2c lists every variable the
    code reads (default, class, meaning); a new getenv / os.environ read without a row fails here."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r'(?:getenv\(|env_ll\(|environ(?:\.get)?[\(\[])\s*"(DODA_[A-Z0-9_]+)"')
    read = set()
    files = [os.path.join(root, "bench.py")]
    for ext in ("py", "hip", "hpp", "cpp"):
        files += glob.glob(os.path.join(root, "doda_amd", "**", "*." + ext), recursive=True)
    for f in files:
        with open(f, errors="replace") as fh:
            read.update(pat.findall(fh.read()))
    assert len(read) > 40, sorted(read)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    section = doc[doc.index("## 2c. Environment switches"):doc.index("## 2d. Training entry point")]
    missing = sorted(v for v in read if "`" + v + "`" not in section)
    assert not missing, missing
