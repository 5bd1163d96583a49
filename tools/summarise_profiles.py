#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag> (tools/collect_profiles.sh) into profiles/<tag>/: per-kernel stats tables (library
kernels only), the bench line, PMC sums per kernel of the LAST dispatch, and HBM traffic with the gfx950 FETCH_SIZE
correction (x2 for wide coalesced reads; MI355X_MICROARCH.md, HBM).  usage: summarise_profiles.py <tag>"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
LIB = re.compile(r"anonymous namespace")


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n.split("(")[0]


lines = []
for what in ("bench", "benchfull", "gqi", "dti", "stream", "dsi", "c5"):
    fs = glob.glob(os.path.join(src, what, "**", "*kernel_stats.csv"), recursive=True)
    if not fs:
        continue
    shutil.copy(fs[0], os.path.join(dst, what + "_kernel_stats.csv"))
    lines.append("== rocprofv3 --kernel-trace --stats -- python3 %s  (library kernels) ==" % ("bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 5" if what == "bench" else "bench.py --no-cpu-baseline" if what == "benchfull" else "tools/prof_step.py %s 5" % what))
    for r in csv.DictReader(open(fs[0])):
        if LIB.search(r["Name"]):
            lines.append("  %-60s calls=%4d avg_us=%10.1f min_us=%10.1f" % (short(r["Name"])[:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
for bj, name in (("bench.json", "bench_under_rocprof.json"), ("benchfull.json", "bench_full_under_rocprof.json")):
    if os.path.exists(os.path.join(src, bj)):                 # (bench.py's stdout: the result line is the last one that starts with `{`)
        res = [ln for ln in open(os.path.join(src, bj)).read().splitlines() if ln.startswith("{")]
        if res:
            open(os.path.join(dst, name), "w").write(res[-1] + "\n")


def pmc(sub):
    fs = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(dict))          # kernel -> dispatch -> counter -> value
    for f in fs:
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][int(r["Dispatch_Id"])][r["Counter_Name"]] = acc[short(r["Kernel_Name"])][int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return {k: v[max(v)] for k, v in acc.items()}


traffic = {}
c5_bytes = None
for what in ("gqi", "dti", "dsi", "stream", "c5"):
    fe, wr = pmc(what + "_fetch"), pmc(what + "_write")
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, {}).get("FETCH_SIZE"); w = wr.get(k, {}).get("WRITE_SIZE")
        if f is None and w is None:
            continue
        if what == "c5":                                     # the fused tracer on 3 peaks x ~10 M lines: one launch = C5's tracking step
            if k.startswith("stream_trace_kernel<3") and (f or 0) + (w or 0) > 1e6:
                c5_bytes = (2.0 * (f or 0) + (w or 0)) * 1024.0
                traffic["c5: " + k] = dict(FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, hbm_bytes_per_launch=c5_bytes)
            continue
        if (f or 0) + (w or 0) < 1000 or not re.match(r"(odf_|fit_|stream_|dsi_|mask_|zero_|qa_|scan_)", k) or k in traffic:
            continue
        traffic[k] = dict(FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, hbm_bytes_per_launch=(2.0 * (f or 0) + (w or 0)) * 1024.0)
lines.append("== HBM traffic per launch (last dispatch; FETCH_SIZE x2 = gfx950 correction for wide coalesced reads) ==")
for k, v in traffic.items():
    lines.append("  %-44s fetch(x2)=%8.3f GB  write=%8.3f GB  total=%8.3f GB" % (k[:44], 2 * (v["FETCH_SIZE_KB"] or 0) * 1024 / 1e9, (v["WRITE_SIZE_KB"] or 0) * 1024 / 1e9, v["hbm_bytes_per_launch"] / 1e9))
for sub in ("gqi_sq", "gqi_sq2", "dsi_sq", "dsi_sq2"):
    for k, v in pmc(sub).items():
        if "gemm" in k or "peaks" in k or "dsi2" in k:
            lines.append("  PMC %-40s %s" % (k[:40], ", ".join("%s=%.4g" % kv for kv in sorted(v.items()))))
# the tracer's issue / wait / occupancy counters (one pass per counter set; the last dispatch of each kernel)
st = defaultdict(dict)
for sub in sorted(os.path.basename(d) for d in glob.glob(os.path.join(src, "stream_SQ_*")) if os.path.isdir(d)):
    for k, v in pmc(sub).items():
        if k.startswith("stream_"):
            st[k].update(v)
if st:
    lines.append("== tracer counters (rocprofv3 --pmc, tools/prof_step.py stream; per launch) ==")
    for k, v in st.items():
        lines.append("  PMC %-44s %s" % (k[:44], ", ".join("%s=%.4g" % kv for kv in sorted(v.items()))))
        if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]:
            # SQ_ACTIVE_INST_VALU counts quad-cycles over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            busy = v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)
            lines.append("      -> vector-ALU issue busy %.0f %% of the kernel's cycles on the average SIMD; waiting on an instruction %.0f %% of the wave-cycles"
                         % (100 * busy, 100 * v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)))
for t in ("gqi_timeline.txt", "gqi_timeline_ball.txt"):
    if os.path.exists(os.path.join(src, t)):
        shutil.copy(os.path.join(src, t), os.path.join(dst, t))
        lines.append("== %s ==" % t)
        lines += ["  " + x.rstrip() for x in open(os.path.join(src, t))]
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
gem = [v for k, v in traffic.items() if k.startswith("odf_gemm3_kernel<10, 1, 8, false, true")] or [v for k, v in traffic.items() if k.startswith("odf_gemm")]
if gem:   # what bench.py reports as roofline.traffic (per launch of the dominant kernel)
    json.dump({"odf_gemm_bytes_per_launch": gem[0]["hbm_bytes_per_launch"], "stream_c5_bytes_per_step": c5_bytes, "source": "profiles/%s/traffic.json" % tag,
               "note": "FETCH_SIZE*1024*2 (gfx950 correction for wide coalesced reads) + WRITE_SIZE*1024, separate --pmc passes; algorithmic = 6.63e9 (DWI + mask in, ODF + peaks + qa out)",
               "kernels": traffic}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
open(os.path.join(dst, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
