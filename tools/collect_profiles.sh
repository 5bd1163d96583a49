#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats of bench.py and of the per-stage drivers, then the HBM-traffic
# counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never together with tracing).
# Output: gpurun_out/prof_<tag>/...; summarise with tools/summarise_profiles.py and copy into profiles/.
set -u
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the headline command alone (the extras launch the same kernels at other sizes: the host-tier call runs the contraction kernel on
# 21 chunks), so that the kernel's average here is comparable with the hipEvent average in the JSON line
# (200 timed steps: the kernel's average over the whole run is then the steady state's, comparable with the line's hipEvent average -- the
# preconditioning and warm-up steps in front of the timed region run while the chip still leaves its idle power state)
rocprofv3 --kernel-trace --stats -f csv -d $OUT/bench -o bench -- python3 bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/benchfull -o benchfull -- python3 bench.py --no-cpu-baseline > $OUT/benchfull.json 2> $OUT/benchfull.err
for what in gqi dti stream dsi c5; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/$what -o $what -- python3 tools/prof_step.py $what 5 > $OUT/$what.log 2>&1
done
for what in gqi dti dsi stream c5; do
  rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/${what}_fetch -o fetch -- python3 tools/prof_step.py $what 2 > $OUT/${what}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/${what}_write -o write -- python3 tools/prof_step.py $what 2 > $OUT/${what}_write.log 2>&1
done
# the tracer: occupancy / issue / wait counters ("VALU-bound" has to be shown, not asserted)
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -f csv -d $OUT/stream_$tag -o pmc -- python3 tools/prof_step.py stream 2 > $OUT/stream_$tag.log 2>&1
done
for what in gqi dsi; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -f csv -d $OUT/${what}_sq -o sq -- python3 tools/prof_step.py $what 2 > $OUT/${what}_sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -f csv -d $OUT/${what}_sq2 -o sq2 -- python3 tools/prof_step.py $what 2 > $OUT/${what}_sq2.log 2>&1
done
# one steady-state step as a timeline (start / duration / gap per kernel), all-ones and ball mask
for m in "" 1; do
  tag=gqi_timeline${m:+_ball}
  PROF_BALL_MASK=$m rocprofv3 --kernel-trace -f csv -d $OUT/$tag -o tl -- python3 tools/prof_step.py gqi 40 > $OUT/$tag.log 2>&1
  T=$(find $OUT/$tag -name "*kernel_trace.csv" | head -1)
  python3 tools/step_timeline.py $T mask_compact 20 > $OUT/$tag.txt 2>&1
done
# the raw traces are large (gpurun copies back at most 64 MiB): keep the stats and counter tables only
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
find $OUT -type f -size +2M -exec ls -la {} ;
find $OUT -type f -size +8M -delete
du -sh $OUT
find $OUT -name "*.csv" | head -40
