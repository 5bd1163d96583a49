#!/bin/bash
# PMC view of the streamline trace kernel (run on the GPU box): instruction counts and wave-cycle breakdown
set -u
OUT=gpurun_out/prof_trace
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/a -o a -- python3 tools/prof_step.py stream 2 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv -d $OUT/b -o b -- python3 tools/prof_step.py stream 2 > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_WAVES_EQ_64 -f csv -d $OUT/c -o c -- python3 tools/prof_step.py stream 2 > $OUT/c.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof_trace/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "stream_trace" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(k, "%.4g" % v[-1], "n", len(v))
PY
find $OUT -type f -size +4M -delete
