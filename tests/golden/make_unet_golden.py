"""Generates tests/golden/unet_golden.npz (+ unet_state_keys.json) and tests/golden/unet_golden_120k.npz.

Run in the BUILD container only (needs /root/reference):
    python tests/golden/make_unet_golden.py            # the 2 x 10k-voxel golden (logits, loss; BASELINE config 1 size)
    python tests/golden/make_unet_golden.py 60000      # the 2 x 60k-voxel golden (gradient norms)
Why two: the U-Net's deepest levels shrink the voxel set by ~4.3 per level, so the 20k-voxel batch leaves 5 rows at level 7
and a BatchNorm over 5 rows (1/sqrt(var + 1e-4) up to 100) makes every parameter gradient ill-conditioned — fp32 and fp64
runs of the SAME CPU oracle differ by 14 % (median) there, by 1e-4 (median) / 1.4e-3 (worst) on the 120k-voxel batch.
Gradient parity is therefore asserted on the larger golden, at a tolerance that catches a 1 % error.

It imports the REFERENCE's own model/unet.py + model/unet_block.py (unmodified, from
/root/reference) on top of the CPU oracle's spconv surface (oracle/spconv_cpu.py), builds the
reference SparseConvNet for cfgs/scannet/spconv.yaml's backbone settings, and records its output
on a seeded synthetic batch: logits, loss, per-parameter gradient norms.  The GPU tests then
require doda_amd.model.SparseConvNet over the HIP path to reproduce these numbers.
Only data is stored (inputs are regenerated from the seed; weights from tests.util.deterministic_init).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

GOLDEN_SCENES, GOLDEN_VOXELS, GOLDEN_SEED = 2, 10000, 4242
GRAD_SAMPLE = 4096


def grad_sample_index(name, numel):
    """Positions of the elementwise gradient sample of parameter `name`: a fixed multiplicative sequence
    through the flattened tensor — reproduced by the test, so only the VALUES are stored.  Small tensors are taken whole."""
    import zlib
    if numel <= GRAD_SAMPLE:
        return np.arange(numel, dtype=np.int64)
    start = zlib.crc32(name.encode()) % numel
    return (start + np.arange(GRAD_SAMPLE, dtype=np.int64) * 2654435761) % numel


def import_reference_model():
    from oracle import spconv_cpu
    spconv_cpu.install("spconv")
    # extension modules the reference wrappers import at module scope; the model class itself
    # never calls them (SparseConvNet.forward uses spconv only)
    for name in ("PG_OP", "pointops2_cuda"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import model.unet as ref_unet  # noqa: E402  (the reference's file, imported in place)
    return ref_unet


def golden_batch(voxels=GOLDEN_VOXELS):
    from doda_amd.scene import make_batch
    return make_batch(GOLDEN_SCENES, voxels, GOLDEN_SEED)


def main(voxels=GOLDEN_VOXELS):
    from doda_amd.model import default_cfg
    from oracle import oracle as orc
    from oracle import spconv_cpu
    from tests.util import deterministic_init
    ref_unet = import_reference_model()
    cfg = default_cfg()
    torch.manual_seed(0)
    net = deterministic_init(ref_unet.SparseConvNet(cfg), seed=0).double()
    net.train()
    batch = golden_batch(voxels)
    vf = orc.voxelize_fp(batch["feats"].numpy(), batch["v2p_map"].numpy(), True)
    inp = spconv_cpu.SparseConvTensor(torch.from_numpy(vf).double(), batch["voxel_locs"].int(),
                                      batch["spatial_shape"], GOLDEN_SCENES)
    scores = net(inp, batch["p2v_map"])
    loss = torch.nn.functional.cross_entropy(scores, batch["labels"], ignore_index=255)
    loss.backward()
    grads = {k: float(p.grad.norm()) for k, p in net.named_parameters()}
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    small = voxels == GOLDEN_VOXELS
    if small:
        with open(os.path.join(HERE, "unet_state_keys.json"), "w") as f:
            json.dump(keys, f, indent=0, sort_keys=True)
    names = sorted(grads)
    params = dict(net.named_parameters())
    # elementwise sample of every parameter gradient (a permuted or sign-flipped block inside a weight gradient keeps its norm)
    sample = [params[n].grad.reshape(-1)[torch.from_numpy(grad_sample_index(n, params[n].numel()))].numpy() for n in names]
    gs_off = np.cumsum([0] + [len(v) for v in sample]).astype(np.int64)
    np.savez_compressed(
        os.path.join(HERE, "unet_golden.npz" if small else "unet_golden_%dk.npz" % (GOLDEN_SCENES * voxels // 1000)),
        scores_head=scores[:4096].detach().numpy().astype(np.float32),
        scores_colsum=scores.detach().sum(0).numpy(), scores_abssum=float(scores.detach().abs().sum()),
        loss=float(loss), grad_names=np.array(names), grad_norms=np.array([grads[n] for n in names]),
        grad_sample_offsets=gs_off, grad_sample_values=np.concatenate(sample).astype(np.float32),
        grad_absmax=np.array([float(params[n].grad.abs().max()) for n in names]),
        n_points=batch["locs"].shape[0], n_voxels=batch["voxel_locs"].shape[0],
        voxel_checksum=int(batch["voxel_locs"].numpy().astype(np.int64).sum()),
        running_mean_l1=float(net.state_dict()["output_layer.0.running_mean"].abs().sum()))
    print("golden written: N=%d M=%d loss=%.6f" % (batch["locs"].shape[0], batch["voxel_locs"].shape[0], float(loss)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else GOLDEN_VOXELS)
