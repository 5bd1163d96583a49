#!/usr/bin/env python3
"""Board power and shader clock of plain streaming kernels (torch): a linear device-to-device copy, a linear read (sum), a linear fill --
the baseline for the hot path's kernels, which all sit at the 1 400 W cap (tools/clock_under_load.py).  usage: power_probe.py [seconds]"""
import os, subprocess, sys, threading, time
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def smi(tag):
    o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    sclk = [l.split("(")[-1].split(")")[0] for l in o.split("\n") if "sclk" in l]
    pw = [l.split(":")[-1].strip() for l in o.split("\n") if "Power (W)" in l]
    print("[%s] sclk %s  power %s W" % (tag, sclk[:1], pw[:1]), flush=True)


import torch
dev = torch.device("cuda", 0)
n = 1 << 29                                       # 2 GiB of float32
a = torch.ones(n, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
cases = {
    "copy (2 GiB read + 2 GiB write)": (lambda: b.copy_(a), 2 * 4.0 * n),
    "sum (2 GiB read)": (lambda: a.sum(), 4.0 * n),
    "fill (2 GiB write)": (lambda: b.fill_(1.5), 4.0 * n),
}
for name, (fn, nbytes) in cases.items():
    fn(); torch.cuda.synchronize()
    stop = False

    def sampler():
        k = 0
        while not stop:
            time.sleep(1.0)
            if not stop:
                smi("%s, %d s" % (name, k + 1))
            k += 1
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); it += 20
    el = time.perf_counter() - t0
    stop = True; th.join()
    print("%s: %.3f ms per pass, %.2f TB/s" % (name, el / it * 1e3, nbytes * it / el / 1e12), flush=True)
