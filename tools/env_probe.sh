set -x
mkdir -p gpurun_out/r05
{
nproc; numactl -H 2>/dev/null | head -20; lscpu | head -30
ls /sys/class/drm/ ; for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h; ls $h; for f in power1_average power1_input power1_cap freq1_input energy1_input; do [ -r $h/$f ] && echo "$f: $(cat $h/$f)"; done; done
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head; cat /sys/class/drm/card*/device/numa_node 2>&1
python3 -c "import amdsmi; print('amdsmi ok')" 2>&1 | tail -1
ls /opt/rocm/lib | grep -i smi
rocm-smi --showpower --showclocks --showtoponuma 2>&1 | tail -30
rocm-smi --showenergycounter 2>&1 | tail -8
cat /proc/meminfo | head -5
ulimit -a | head -20
} > gpurun_out/r05/envprobe.txt 2>&1
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r05/bench_base.json 2> gpurun_out/r05/bench_base.err
tail -c 600 gpurun_out/r05/bench_base.json
