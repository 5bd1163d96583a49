#!/usr/bin/env python3
"""How much of the streamline tracer's lane time is idle because the lines of a wave end at different steps?
From the per-line point counts (all_npts): a lane runs npts + (1 or 2) iterations, a wave as long as its slowest lane.
Two phantoms: the smooth field of the headline benchmark (lines use the whole len_max budget) and the bundle phantom
(broad length distribution).  Prints the length statistics, the idle fraction and the measured kernel rates of the
one-lane-per-line kernel and of the self-compacting persistent-wave kernel at several hand-over thresholds; checks that all
give the same bytes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
import ctypes as C

dev = torch.device("cuda", 0)
SHAPE = (140, 140, 140)
L = fj.lib()


def prof(name):
    ms, n = C.c_double(0), C.c_int64(0)
    L.fib_profile_get(name.encode(), C.byref(ms), C.byref(n))
    return ms.value / max(n.value, 1)


def run(label, ovec, mask, nsub):
    field, mout = fj.stream_field_device([ovec], mask=mask)
    seeds = torch.nonzero(mout).flatten()
    sub = torch.from_numpy(fj.make_sublist(nsub, np.random.default_rng(5))).to(dev) if nsub > 1 else torch.tensor([[0.1, -0.2, 0.3]], device=dev)
    buf = {}

    def xyz_out(n):
        if buf.get("t") is None or buf["t"].numel() < 3 * n:
            buf["t"] = torch.empty(3 * n + 16, dtype=torch.float32, device=dev)
        return buf["t"]
    ref = None
    for thr in CONFIGS:
        os.environ["FIBERS_STREAM_COMPACT"] = str(thr)
        r = fj.stream_device(field, SHAPE, seeds, sub, want_all_npts=True, xyz_out=xyz_out)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            r = fj.stream_device(field, SHAPE, seeds, sub, want_all_npts=True, xyz_out=xyz_out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        L.fib_profile_enable(0)
        n = r["all_npts"].cpu().numpy().astype(np.int64)
        it = n + 2
        pad = (-len(it)) % 64
        w = np.concatenate([it, np.zeros(pad, np.int64)]).reshape(-1, 64)
        idle = 1.0 - it.sum() / (w.max(1).sum() * 64.0)
        npnt = int(r["xyz"].shape[0])
        cur = (r["npts"].clone(), r["seed_index"].clone(), r["xyz"][:npnt].clone())
        same = "" if ref is None else (" identical" if all(torch.equal(x, y) for x, y in zip(cur, ref)) else " DIFFERENT")
        if ref is None:
            ref = cur
        print("%-8s nsub %2d %-22s lines %9d points %11d | npts mean %.1f median %.0f p90 %.0f max %d | static lane-idle %.1f %% | trace %.3f ms pack %.3f ms wall %.3f ms -> %.0f Mpoints/s%s"
              % (label, nsub, "one lane per line" if thr == 0 else "compacting, thresh %d" % thr, len(n), npnt, n.mean(), np.median(n), np.percentile(n, 90), n.max(), 100 * idle,
                 prof("stream_trace"), prof("stream_pack"), dt * 1e3, npnt / dt / 1e6, same), flush=True)
    os.environ.pop("FIBERS_STREAM_COMPACT", None)


# 0 = stream_trace_kernel (one lane per line).  The compacting / refilling kernels this tool compared it with (hand-over threshold
# n > 0 through FIBERS_STREAM_COMPACT) were removed after they lost on every workload: profiles/r03/trace_compaction.log holds
# this tool's output for them, the repository's history their source.
CONFIGS = [0]
axes = torch.from_numpy(np.ascontiguousarray(np.moveaxis(phantom.fibre_field(*SHAPE).astype(np.float32), -1, 0).reshape(3, -1, order="F"))).to(dev)
run("smooth", axes, phantom.ball_mask_torch(SHAPE, dev), 1)
ov, m = phantom.bundle_field_torch(SHAPE, dev)
run("bundles", ov, m, 1)
run("bundles", ov, m, 10)
