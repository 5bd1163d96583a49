#!/bin/bash
# samples the shader clock / power while the GQI step runs in a loop (FIBERS_ODF_PIPE selects the kernel)
python tools/pipe_time.py > /tmp/pt_$1.log 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4 | tr "\n" " "; echo
  sleep 0.5
done
wait $PID
grep "kernels" /tmp/pt_$1.log
