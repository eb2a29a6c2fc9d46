"""The oracle's own U-Net restatement (used for the CPU baseline) against the golden output of the
REFERENCE's model files: same state-dict layout, same logits / loss / gradient norms."""
import json
import os

import numpy as np
import torch

from tests.util import deterministic_init

G = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_unet_reproduces_reference_golden(native_lib, oracle):
    from doda_amd.scene import make_batch
    from oracle.unet_cpu import OracleUNet, forward_backward
    g = np.load(os.path.join(G, "unet_golden.npz"))
    net = deterministic_init(OracleUNet(), seed=0).double().train()
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert keys == json.load(open(os.path.join(G, "unet_state_keys.json")))
    batch = make_batch(2, 10000, 4242)
    assert int(batch["voxel_locs"].numpy().astype(np.int64).sum()) == int(g["voxel_checksum"])
    scores, loss = forward_backward(net, batch)
    assert np.abs(scores[:4096].detach().numpy() - g["scores_head"]).max() < 1e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-9
    grads = {k: float(p.grad.norm()) for k, p in net.named_parameters()}
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        assert abs(grads[str(name)] - ref) <= 1e-8 * max(ref, 1.0)
