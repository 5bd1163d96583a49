tools/probes/mfma_power_probe 5 > /tmp/pp.log 2>&1 &
PID=$!
for i in $(seq 1 36); do sleep 0.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*(\([0-9]*\)Mhz).*/\1 MHz/; s/.*Power (W): \([0-9.]*\).*/\1 W/' | tr "\n" " "; echo; done
wait $PID
cat /tmp/pp.log
