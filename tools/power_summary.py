#!/usr/bin/env python3
"""profiles/<tag>/power.txt from a bench_extra.json: roofline.power in words, with ONE set of Joules-per-unit figures -- the calibrated one (every
probe figure of profiles/energy_model.json scaled by the one factor that makes components + idle equal the step's MEASURED Joules) -- in every
sentence (round 5's text mixed raw and calibrated figures).  usage: power_summary.py <bench_extra.json> <out.txt>"""
import json
import sys

d = json.load(open(sys.argv[1]))
pw = d["roofline"]["power"]
sc = pw["calibration_scale"]
comp = pw["joules_by_component"]
cnt = pw["counts_per_step"]
lines = []
w = lines.append
w("roofline.power of the GQI step (140^3 x 270, fp16 pieces), calibrated figures throughout")
w("=" * 96)
w("measured, live: %.3f J per step at %.0f W (board energy counter around %d back-to-back steps), step %.3f ms, contraction kernel %.3f ms;"
  % (pw["measured_joules_per_step"], pw["board_watts_while_stepping"], pw["steps_measured"], pw["step_ms"], pw["kernel_ms"]))
w("idle board %.0f W -> %.3f J of the step; cap %.0f W -> %.0f W for everything that switches." % (pw["idle_w"], pw["idle_joules_per_step"], pw["cap_w"], pw["budget_w"]))
w("stored (profiles/energy_model.json: each ingredient alone under the same counter), scaled by %.3f so that components + idle = the measured Joules" % sc)
w("(the probes for HBM, LDS and the vector ALU ran at 2.4 GHz and its voltage, the kernel runs at 1.7-1.8 GHz where every operation costs less):")
w("")
w("  %-26s %12s %14s %10s %8s" % ("component", "count/step", "pJ per unit", "J/step", "share"))
tot = pw["measured_joules_per_step"]
for k in ("hbm_bytes", "mfma_flops", "lds_fragment_bytes", "l2_to_lds_bytes", "lds_other_bytes", "valu_wave_instructions"):
    pj = pw["joules_per_unit"][k] * sc * 1e12
    w("  %-26s %12.4g %14.4g %10.3f %7.0f %%" % (k, cnt[k], pj, comp[k], 100 * comp[k] / tot))
w("  %-26s %12s %14s %10.3f %7.0f %%" % ("idle board", "", "", pw["idle_joules_per_step"], 100 * pw["idle_joules_per_step"] / tot))
un = sum(comp[k] for k in pw["unavoidable_components"])
w("")
w("the algorithm's own part = its HBM bytes + its executed MFMA flops = %.3f J; at %.0f W that is %.3f ms -> roofline.power.frac = %.2f of the kernel's %.3f ms."
  % (un, pw["budget_w"], pw["floor_ms"], pw["frac"], pw["kernel_ms"]))
w("In bytes: a kernel that spent nothing else would move the step's %.2f GB in %.3f ms = %.2f TB/s under this cap; the kernel moves them at %.2f TB/s."
  % (cnt["hbm_bytes"] / 1e9, pw["floor_ms"], cnt["hbm_bytes"] / pw["floor_ms"] / 1e9, cnt["hbm_bytes"] / pw["kernel_ms"] / 1e9))
w("What the kernel adds on top -- vector ALU %.2f J, LDS fragment re-reads %.2f, other LDS %.2f, L2 -> LDS %.2f = %.2f J -- is %.0f %% of the step."
  % (comp["valu_wave_instructions"], comp["lds_fragment_bytes"], comp["lds_other_bytes"], comp["l2_to_lds_bytes"],
     sum(comp[k] for k in comp if k not in pw["unavoidable_components"]), 100 * sum(comp[k] for k in comp if k not in pw["unavoidable_components"]) / tot))
w("(for comparison with earlier rounds only: the same arithmetic WITHOUT the scaling, each ingredient at what it costs alone at its own clock,")
w(" over-counts the step by %.0f %% and gives frac_raw = %.2f.)" % (100 * (pw["model_over_measured"] - 1), pw["frac_raw"]))
ck = (d.get("extra") or {}).get("in_kernel_clock") or {}
if "gqi_fused" in ck:
    g = ck["gqi_fused"]
    w("clocks in the same seconds (diagnostic build, one stamp pair per workgroup): in-kernel %.3f GHz (min %.3f, max %.3f), SMU %.0f MHz; %.0f W."
      % (g["clock_ghz_median"], g["clock_ghz_min"], g["clock_ghz_max"], g["smu_sclk_mhz_mean"], g["board_watts"]))
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
