#!/usr/bin/env python
"""One-screen summary of a bench.py JSON line: step time, roofline record, gate (B4 / one scene).  usage: benchsum.py file..."""
import json, sys
for path in sys.argv[1:]:
    line = [l for l in open(path) if l.startswith("{")][-1]
    d = json.loads(line)
    r = d["roofline"]; det = r["detail"]
    print("== %s: %.3f ms/step  %.1f M voxels/s  loss %.3f" % (path, d["ms_per_step"], d["value"] / 1e6, d["config"]["final_loss"]))
    print("   roofline %s..." % r["kernel"][:40])
    print("   step-form cold %.1f us frac %.3f (8d %.3f) | warm frac %.3f | plain cold %.3f warm %.3f | in-step %s" % (
        r["avg_launch_us"], r["frac"], r.get("frac_8d", float("nan")), r["frac_warm"], r["frac_plain_cold"], r["frac_plain_warm"],
        r.get("in_step_rocprof_avg_us")))
    for key in ("subm16_fwd_bwd", "gate_150k"):
        if key in det:
            g = det[key]
            print("   %-15s cold fwd %.1f dgrad %.1f wgrad %.1f us -> %.3f | warm fwd %.1f dgrad %.1f wgrad %.1f -> %.3f" % (
                key, g["fwd_us"], g["dgrad_us"], g["wgrad_us"], g["frac_of_hbm_peak"], g["warm"]["fwd_us"], g["warm"]["dgrad_us"],
                g["warm"]["wgrad_us"], g["warm"]["frac_of_hbm_peak"]))
    sd = det["subm16_dgrad"]
    print("   dgrad step-form cold %.1f us warm %.1f us" % (sd["step_cold"]["us"], sd["step_warm"]["us"]))
    if "fp32" in d:
        f = d["fp32"]
        print("   fp32: %.3f ms/step; gate cold %.3f" % (f["ms_per_step"], f["roofline"]["detail"]["subm16_fwd_bwd"]["frac_of_hbm_peak"]))
    print("   step frac of HBM %.4f" % r["step"]["frac_of_hbm_peak"])
