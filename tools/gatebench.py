"""The bench line's kernel section alone (roofline kernel + north-star gate, cold and warm), without the training
runs: python tools/gatebench.py [--dtype bf16] [--reps 50] [--no-150k]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--no-150k", action="store_true")
    a = ap.parse_args()
    from doda_amd.scene import make_batch
    d = torch.device("cuda:0")
    b = make_batch(a.scenes, 150000, 1000)
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in b.items()}
    one = None
    if not a.no_150k:
        o = make_batch(1, 150000, 1000)
        one = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in o.items()}
    roof, ppv = bench.kernel_roofline(bd, a.dtype, a.reps, one)
    det = roof.pop("detail")
    out = {"roofline": roof, "pairs_per_voxel": ppv}
    for k in ("subm16_fwd", "subm16_dgrad"):
        out[k] = {kk: (round(v["us"], 2), round(v["frac_of_hbm_peak"], 3)) for kk, v in det[k].items() if kk != "algorithmic_bytes"}
    out["wgrad"] = {kk: (round(v["us"], 2), round(v["frac_of_hbm_peak"], 3)) for kk, v in det["subm16_wgrad"].items() if isinstance(v, dict)}
    out["gate_B4"] = det["subm16_fwd_bwd"]
    out["gate_150k"] = det.get("gate_150k")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
