export TMPDIR=/tmp
OUT=gpurun_out/pmc_stream
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TA_BUSY_avr" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"; do
  i=$((i+1))
  PROF_STREAM_PREALLOC=1 rocprofv3 --pmc $set -f csv -d $OUT/s$i -o pmc -- python3 tools/prof_step.py stream 2 > $OUT/s$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
def short(n):
    n = re.sub(r'^void ', '', n); n = re.sub(r'\(anonymous namespace\)::', '', n); return n.split('(')[0]
for f in sorted(glob.glob('gpurun_out/pmc_stream/s*/**/*counter_collection.csv', recursive=True)):
    acc = collections.OrderedDict()
    rows = list(csv.DictReader(open(f)))
    # last dispatch of each kernel
    last = {}
    for r in rows:
        k = short(r['Kernel_Name'])
        if 'stream_trace' in k or 'stream_pack' in k:
            last[k] = r['Dispatch_Id']
    for r in rows:
        k = short(r['Kernel_Name'])
        if k in last and r['Dispatch_Id'] == last[k]:
            acc.setdefault(k, {})[r['Counter_Name']] = acc.setdefault(k, {}).get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for k, v in acc.items():
        print(k, ' '.join('%s=%.4g' % kv for kv in v.items()))
PY
rm -rf $OUT/s*/
