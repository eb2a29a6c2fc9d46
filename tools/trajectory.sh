#!/bin/bash
# bf16 against fp32 training trajectory through `python -m doda_amd.train`: the same HBM-resident 32-scene
# dataset, the same seed, EPOCHS x 8 optimizer steps each (default 40 = 320 steps); per epoch the mean training loss, every 5th epoch
# the held-out loss / mIoU / per-class IoU.  Writes gpurun_out/traj_{f32,bf16}.json and the summary gpurun_out/trajectory.json.
cd ${GRAFT_REPO_ROOT:-/root/repo}
EPOCHS=${EPOCHS:-40}; VOX=${VOX:-150000}; SEED=${SEED:-0}
# third run: fp32 again with another weight-initialisation seed — the run-to-run spread the bf16 difference is judged against
for run in f32:$SEED bf16:$SEED f32b:$((SEED + 1)); do
  dt=${run%%:*}; seed=${run##*:}; rm -rf /tmp/traj_$dt
  python -m doda_amd.train --cfg_file doda_amd/cfgs/synthetic/spconv.yaml --dtype ${dt%b} --batch_size 4 --synthetic_scenes 32 \
      --synthetic_voxels $VOX --synthetic_base 16 --epochs $EPOCHS --manual_seed $seed --print_freq 100000 --ckpt_save_freq 100000 \
      --output_root /tmp/traj_$dt --curve_json gpurun_out/traj_$dt.json --set EVALUATION.eval_freq 5 OPTIMIZATION.NUM_EPOCHS $EPOCHS \
      > gpurun_out/traj_$dt.log 2>&1 || { tail -20 gpurun_out/traj_$dt.log; exit 1; }
done
python - <<'PY'
import json
a, b, c = (json.load(open("gpurun_out/traj_%s.json" % d)) for d in ("f32", "bf16", "f32b"))
ca, cb, cc = a["curve"], b["curve"], c["curve"]
assert len(ca) == len(cb) and len(ca) >= 10
w = max(5, len(ca) // 8)                                 # final window: the last eighth of the run
la = sum(e["train_loss"] for e in ca[-w:]) / w
lb = sum(e["train_loss"] for e in cb[-w:]) / w
va = [e for e in ca if "val_iou" in e][-1]
vb = [e for e in cb if "val_iou" in e][-1]
d_iou = [abs(x - y) for x, y in zip(va["val_iou"], vb["val_iou"])]
vc = [e for e in cc if "val_iou" in e][-1]
lc = sum(e["train_loss"] for e in cc[-w:]) / w
d_seed = [abs(x - y) for x, y in zip(va["val_iou"], vc["val_iou"])]
out = {"steps": ca[-1]["iterations"], "final_window_epochs": w, "final_window_loss": {"f32": la, "bf16": lb, "rel_diff": abs(la - lb) / la},
       "held_out": {"f32": {k: va[k] for k in ("val_loss", "val_miou", "val_allacc")}, "bf16": {k: vb[k] for k in ("val_loss", "val_miou", "val_allacc")},
                    "miou_abs_diff": abs(va["val_miou"] - vb["val_miou"]), "per_class_iou_max_abs_diff": max(d_iou), "per_class_iou_abs_diff": d_iou},
       "fp32_other_seed": {"final_window_loss": lc, "loss_rel_diff": abs(la - lc) / la, "val_miou": vc["val_miou"],
                           "miou_abs_diff": abs(va["val_miou"] - vc["val_miou"]), "per_class_iou_max_abs_diff": max(d_seed), "per_class_iou_abs_diff": d_seed},
       "val_iou": {"f32": va["val_iou"], "bf16": vb["val_iou"], "f32_other_seed": vc["val_iou"]},
       "train_loss_curve": {"f32": [e["train_loss"] for e in ca], "bf16": [e["train_loss"] for e in cb], "f32_other_seed": [e["train_loss"] for e in cc]},
       "val_miou_curve": {"f32": [(e["epoch"], e["val_miou"]) for e in ca if "val_miou" in e], "bf16": [(e["epoch"], e["val_miou"]) for e in cb if "val_miou" in e]},
       "config": {k: a[k] for k in ("seed", "batch_size_per_gpu", "scenes_per_epoch", "voxels_per_scene")}}
json.dump(out, open("gpurun_out/trajectory.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("steps", "final_window_loss")}))
print(json.dumps(out["held_out"]))
print(json.dumps(out["fp32_other_seed"]))
print("loss f32 ", " ".join("%.3f" % v for v in out["train_loss_curve"]["f32"]))
print("loss bf16", " ".join("%.3f" % v for v in out["train_loss_curve"]["bf16"]))
PY
