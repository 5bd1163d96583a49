#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X back end (contract: the task description / DESIGN.md §3).

Metric (BASELINE.json): Mvoxels/s (fit) + Mpoints/s (streamline) on a synthetic 140^3 x 270-direction HCP-like volume
at 1/2/4/8 GPUs.  A "step" = one pass of the GQI hot path (gqi.jl:109-171: ODF contraction on the matrix cores with
find_peaks! fused into its epilogue, exact odfmax, QA normalisation) over ONE resident 140^3 x 270 volume.

N GPUs (one process per GPU, torch.distributed, backend nccl == RCCL over xGMI): the volume is cut into contiguous
z-slabs as the reference threads its z loop (gqi.jl:132); every rank reconstructs its slab; the path's only exchange
step, odfmax = maximum(mean(odf, dims=4)) (gqi.jl:164), is a 2-float all-reduce(MAX) inside the timed region, then qa ./=
odfmax on every rank.  Strong scaling: the total work is fixed.  `value` is whole-job Mvoxels/s with inputs in HBM.

Output: ONE final stdout line, JSON, <= 6 KB (compact_line below: the contract's keys, `roofline`, `cpu_baseline` and a
number-only `extra` per leg).  Everything else the run measured (tools/bench_legs.py: the other BASELINE configs, host-tier
stages, energy components, clocks) goes to bench_extra.json next to this file and to an earlier stdout line that starts
with `bench_extra ` -- never into the result line (round 5's 25 KB line could not be parsed by the driver)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

SHAPE = (140, 140, 140)
PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 MFMA (v_mfma_f32_32x32x2_f32) dense peak
PEAK_BF16_TFLOPS = 2500.0    # MI355X_MICROARCH.md: BF16/F16 MFMA dense peak (v_mfma_f32_32x32x16_*, 32 cycles each)
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
MAX_LINE = 6144              # bytes of the result line (tests/test_bench_cli.py asserts it)


# ---- CPU baselines: the oracle (C/OpenMP restatement of the reference's CPU path with its z-slice / seed-chunk threading) on
# the box's host cores, bounded samples of the same workloads.  Every figure is the MEDIAN of NRUNS timed runs of >= RUN_S seconds
# each (a run = one or more calls on the same resident sample) after an untimed warm-up call on that sample (thread pool, input
# pages); the sample is sized from short probe calls: the whole 140^3 volume when a call on it stays within a few seconds, else
# whole slices in multiples of the thread count (one z-slice per thread and trip: the reference's static z-slice threading,
# dti.jl:258 / gqi.jl:132 / dsi.jl:197, stays balanced), else `cores` slices of fewer rows.  Outputs are allocated zero-filled
# inside every call, as the reference's entry points do (`MRI(mask, n, Float32)` -> zeros, mri.jl:251-255).  This is the ONLY
# part of this file (and of tools/bench_legs.py) that touches oracle/. ---------------------------------------------------------
RUN_S = 2.5
NRUNS = 3


def _host_volume(shape3, bval, bvec, seed, **kw):
    """[nx, ny, nz, nvol] float32, Fortran order (== MRI.vol memory), generated on the GPU when there is one (NumPy takes
    minutes for 3 GB) and copied to pageable host memory"""
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    d, _ = phantom.make_dwi_torch(shape3, bval, bvec, seed=seed, device=dev, **kw)
    h = d.cpu().numpy()
    del d
    return h.T.reshape(tuple(shape3) + (len(bval),), order="F")


def _sample_shape(target_vox, cores):
    nx, ny, nz = SHAPE
    if target_vox >= nx * ny * nz:
        return (nx, ny, nz)
    nzc = min(nz, cores)
    if target_vox >= nx * ny * nzc:
        return (nx, ny, int(min(nz, target_vox // (nx * ny)) // nzc) * nzc)
    return (nx, int(max(1, target_vox // (nx * nzc))), nzc)


def _median_runs(call):
    """NRUNS runs, each = calls on the same sample until RUN_S seconds have passed -> (median seconds per call, the runs' seconds
    per call, calls in the median run)"""
    per, reps = [], []
    for _ in range(NRUNS):
        n, t0 = 0, time.perf_counter()
        while True:
            call()
            n += 1
            el = time.perf_counter() - t0
            if el >= RUN_S:
                break
        per.append(el / n)
        reps.append(n)
    k = int(np.argsort(per)[len(per) // 2])
    return float(per[k]), per, reps[k]


def _fit_baseline(fit, label, bval, bvec, seed, cores, **gen_kw):
    nx, ny, nz = SHAPE
    full = nx * ny * nz
    shp = _sample_shape(nx * 2 * min(nz, cores), cores)
    while True:                                                   # probe calls: grow the sample until a call takes >= 0.25 s (or it is the whole volume)
        dwi = _host_volume(shp, bval, bvec, seed, **gen_kw)
        mask = np.ones(shp, np.uint8)
        fit(dwi, mask)                                            # (untimed: thread pool, first touch of the input)
        t0 = time.perf_counter()
        fit(dwi, mask)
        t = time.perf_counter() - t0
        nv = shp[0] * shp[1] * shp[2]
        if t >= 0.25 or nv >= full:
            break
        shp = _sample_shape(nv * min(16.0, 0.6 / max(t, 1e-4)), cores)
        del dwi
    want = _sample_shape(nv / t * RUN_S, cores)
    if want != shp:
        del dwi
        shp = want
        dwi = _host_volume(shp, bval, bvec, seed, **gen_kw)
        mask = np.ones(shp, np.uint8)
        fit(dwi, mask)                                            # warm-up on the final sample
    nv = shp[0] * shp[1] * shp[2]
    med, per, reps = _median_runs(lambda: fit(dwi, mask))
    return dict(value=nv / med / 1e6, unit="Mvoxels/s", cores=cores, kind="port", spread=(max(per) - min(per)) / med,
                sample="%s, %dx%dx%d sample, all-ones mask: median of %d runs x %d call(s), %.1f s/run" % (label, shp[0], shp[1], shp[2], NRUNS, reps, med * reps))


def cpu_baseline_gqi(bval, bvec, sph, seed):
    from oracle import oracle as orc
    cores = orc.max_threads()
    return _fit_baseline(lambda d, m: orc.gqi_rec(d, m, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=cores),
                         "oracle gqi_rec (C/OpenMP restatement of gqi.jl:109-201, z-slice threads), %d frames" % len(bval), bval, bvec, seed, cores)


def cpu_baseline_dti(bval, bvec):
    from oracle import oracle as orc
    cores = orc.max_threads()
    return _fit_baseline(lambda d, m: orc.dti_fit(d, m, bval, bvec, nthreads=cores),
                         "oracle dti_fit (dti.jl:221-335), %d frames" % len(bval), bval, bvec, 2, cores, nfib=1)


def cpu_baseline_dsi(bval, bvec, sph):
    from oracle import oracle as orc
    cores = orc.max_threads()
    return _fit_baseline(lambda d, m: orc.dsi_rec(d, m, bval, bvec, sph.vertices, sph.faces, 32, nthreads=cores),
                         "oracle dsi_rec (dsi.jl:171-270: 16^3 FFT + trilinear radial integration), %d frames" % len(bval), bval, bvec, 5, cores)


def cpu_baseline_stream():
    """tracking on the analytic fibre field, ball mask, one sub-voxel offset: a z-range of seeds of the full 140^3 field"""
    from oracle import oracle as orc
    from fibers_jl_amd import phantom
    cores = orc.max_threads()
    ov = np.asfortranarray(phantom.fibre_field(*SHAPE).astype(np.float32))
    mask = phantom.ball_mask(*SHAPE)
    sub = np.array([[0.1, -0.2, 0.3]], np.float32)
    res = {}

    def seeds(nzs):
        seed = np.zeros(SHAPE, np.uint8, order="F")
        z0 = SHAPE[2] // 2 - nzs // 2
        seed[:, :, z0:z0 + nzs] = mask[:, :, z0:z0 + nzs]
        return seed

    def run(seed):
        t0 = time.perf_counter()
        r = orc.stream(ov, sub, mask=mask, seed=seed, nthreads=cores)
        res["points"], res["seeds"] = int(r["xyz"].shape[0]), int(seed.sum())
        return time.perf_counter() - t0
    nzs = 4
    while True:
        sd = seeds(nzs)
        run(sd)
        t = run(sd)
        if t >= 0.25 or nzs >= SHAPE[2]:
            break
        nzs = int(min(SHAPE[2], max(nzs + 1, nzs * min(16.0, 0.6 / max(t, 1e-4)))))
    want = int(min(SHAPE[2], max(2, round(nzs * RUN_S / t))))
    if want != nzs:
        nzs = want
        sd = seeds(nzs)
        run(sd)
    med, per, reps = _median_runs(lambda: run(sd))
    return dict(value=res["points"] / med / 1e6, unit="Mpoints/s", cores=cores, kind="port", spread=(max(per) - min(per)) / med,
                sample="oracle stream (stream.jl:625-790, seed chunks per thread), %d seeds (%d z-slices of the ball), %d points/call: median of %d runs x %d call(s)"
                       % (res["seeds"], nzs, res["points"], NRUNS, reps))


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) the way the driver does --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>` --
    as a child process, pass rank 0's JSON line through and return the launcher's exit code (non-zero if any rank failed or no line
    came back).  Runs before torch is imported: the parent never initialises the GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith("{"):
            line = ln.rstrip("\n")
        elif ln.startswith("bench_extra "):
            sys.stdout.write(ln)
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank launch failed (exit code %d)\n" % (n, rc))
        return rc
    if line is None:
        sys.stderr.write("bench.py: the %d-rank launch produced no result line\n" % n)
        return 1
    try:
        got = json.loads(line).get("n_gpus")
    except ValueError:
        got = None
    if got != n:
        sys.stderr.write("bench.py: asked for %d ranks, the result line says n_gpus = %r\n" % (n, got))
        return 1
    print(line, flush=True)
    return 0


# ---- the result line ---------------------------------------------------------------------------------------------------------------
# what the line keeps of each leg of `extra` (tools/bench_legs.py returns more; all of it is in bench_extra.json): numbers only
KEEP = {
    "host_tier": None,                                     # (handled below: per entry point, median ms + fraction of the PCIe roof at the median)
    "pipeline": ("read_ms", "fit_ms", "track_ms", "write_ms", "total_ms", "host_path_total_ms", "lines", "points"),
    "gqi_ball_mask_nonpositive": ("ms_per_step", "mvoxels_in_mask_per_s", "gemm_kernel_ms"),
    "gqi_slab_1of8": ("ms_per_step", "ideal_ms", "efficiency_before_collectives"),
    "gqi_other_format": ("ms_per_step", "mvoxels_per_s", "gemm_kernel_ms"),
    "gqi_weak_scaling": ("mvoxels_per_s", "ms_per_step"),
    "dti_fit_140x64": ("mvoxels_per_s", "ms_per_step", "kernel_ms", "hbm_frac"),
    "stream_dti_ball": ("mpoints_per_s", "ms_per_step", "kernel_sum_ms", "lines", "points", "frac", "traffic_frac"),
    "stream_dti_ball_trilinear": ("mpoints_per_s", "ms_per_step"),
    "stream_micro_ball": ("mpoints_per_s", "ms_per_step"),
    "stream_lcm_2d": ("mpoints_per_s", "ms_per_step"),
    "rumba_140_ball": ("ms_per_iteration", "gemm_tflops"),
    "dsi_rec_140x515": ("mvoxels_per_s", "ms_per_step", "gemm_kernel_ms", "frac"),
    "stream_dsi_3peaks_10M": ("mpoints_per_s", "ms_per_step", "kernel_sum_ms", "lines", "points", "frac", "traffic_frac"),
}


def _r(v, sig=5):
    """numbers to `sig` significant digits (the line is for reading; bench_extra.json keeps every digit)"""
    if isinstance(v, bool) or v is None or isinstance(v, (str, int)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, v))
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def compact_line(full):
    """the contract's keys + roofline + cpu_baseline + number-only extras; strings clipped (dtype 120, workload 200, kernel 100)"""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    line["dtype"] = full["dtype"][:120]
    line["data"] = full["data"]
    cfg = dict(full["config"])
    cfg["workload"] = cfg["workload"][:200]
    line["config"] = cfg
    rf = full["roofline"]
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "algorithmic_bytes", "avg_kernel_ms", "launches", "traffic", "mfma_frac", "step_launches_ms")
    roof = {k: rf.get(k) for k in keep if k in rf}
    roof["kernel"] = str(roof.get("kernel", ""))[:100]
    pw = rf.get("power")
    if isinstance(pw, dict) and "error" not in pw:
        roof["power"] = dict(frac=pw.get("frac"), frac_raw=pw.get("frac_raw"), measured_joules_per_step=pw.get("measured_joules_per_step"),
                             board_watts=pw.get("board_watts_while_stepping"))
    line["roofline"] = roof
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = dict(value=cb["value"], unit=cb["unit"], cores=cb["cores"], kind=cb["kind"], spread=cb.get("spread"), sample=cb["sample"][:160])
    else:
        line["cpu_baseline"] = None
    ex = {}
    for leg, rec in (full.get("extra") or {}).items():
        if not isinstance(rec, dict):
            continue
        if "error" in rec:
            ex[leg] = dict(error=1)
        elif leg == "host_tier":
            ex[leg] = {k: (dict(ms_median=v["e2e_pcie_ms_median"], pcie_roof_frac=v["pcie_floor_ms"] / v["e2e_pcie_ms_median"]) if "e2e_pcie_ms_median" in v else dict(error=1))
                       for k, v in rec.items() if isinstance(v, dict)}
        elif leg == "cpu_baselines":
            ex[leg] = {k: v["value"] for k, v in rec.items() if isinstance(v, dict) and "value" in v}
        elif leg in KEEP:
            ex[leg] = {k: rec[k] for k in KEEP[leg] if rec.get(k) is not None}
    line["extra"] = ex
    line["extra_file"] = "bench_extra.json"
    return _r(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()

    # ---- `--gpus N` means N ranks.  Under a launcher (WORLD_SIZE set: the driver's torch.distributed.run) this process is one of
    # them.  Without one and N > 1 the N ranks are started here as CHILD processes of torch.distributed.run, before this process has
    # imported torch or made any HIP call (a process that has touched the GPU must never exec another program on this pool), rank 0's
    # JSON line is forwarded and a failing rank makes this process fail.  In every case world == --gpus or the run stops: a
    # `--gpus 8` run can never print an `n_gpus: 1` line.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree "
                         "(run `python bench.py --gpus N` without a launcher, or torch.distributed.run --nproc-per-node N ... --gpus N)\n"
                         % (args.gpus, world))
        sys.exit(2)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (1-GPU box): FIBERS_BENCH_BACKEND=gloo FIBERS_BENCH_ONE_DEVICE=1 runs N ranks on cuda:0 over gloo (RCCL refuses two
    # ranks on one device); FIBERS_BENCH_SHAPE shrinks the volume; FIBERS_BENCH_FORCE_PG=1 creates the process group (nccl = RCCL)
    # with ONE rank and takes every multi-rank branch, so that this file's RCCL code has run on hardware before an 8-GPU launch
    backend = os.environ.get("FIBERS_BENCH_BACKEND", "nccl")
    shape = tuple(int(v) for v in os.environ.get("FIBERS_BENCH_SHAPE", "140,140,140").split(","))
    if os.environ.get("FIBERS_BENCH_ONE_DEVICE"):
        local = 0
    force_pg = os.environ.get("FIBERS_BENCH_FORCE_PG", "0") not in ("", "0")
    multi = world > 1 or force_pg
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if multi else 0)

    import fibers_jl_amd as fj
    from fibers_jl_amd import dist as fd, phantom
    import bench_legs as legs
    L = fj.lib()
    nx, ny, nz = shape
    nxy = nx * ny
    nvox = nxy * nz
    sph = fj.sphere_642
    z0, z1 = fd.slab_bounds(nz, world, rank, nxy)
    v0, v1 = z0 * nxy, z1 * nxy
    nloc = v1 - v0
    counts = [(b - a) * nxy for a, b in (fd.slab_bounds(nz, world, r, nxy) for r in range(world))]

    def prof_get(name):
        import ctypes as C
        ms, n = C.c_double(0), C.c_int64(0)
        L.fib_profile_get(name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    PRECOND_S = 0.15

    def timed(fn, steps, warmup, only=None):
        """W untimed steps, then K steps bracketed by barrier + synchronize; max over ranks.  Before the W warm-up steps the same step
        runs untimed for about PRECOND_S seconds (a step count derived from the first step's duration, identical on every rank): after
        the idle stretch that precedes every section (plan set-up, input generation) the chip needs ~50 ms of load to leave its idle
        power state (tools/step_evolution.py, profiles/r03/step_evolution.txt) and W = 2..5 steps end inside that ramp.  What is timed
        is the steady state a stream of volumes sees."""
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        npre = max(5.0, min(2000.0, PRECOND_S / max(time.perf_counter() - t0, 1e-5)))   # (>= 5: the first call may carry one-off costs)
        if multi:                                        # a step may hold a collective: the SAME number of steps on every rank
            tn = torch.tensor([npre], device=dev, dtype=torch.float64)
            dist.all_reduce(tn, op=dist.ReduceOp.MAX)
            npre = float(tn.item())
        for _ in range(int(npre)):
            fn()
        torch.cuda.synchronize()
        for _ in range(warmup):
            fn()
        sync()
        L.fib_profile_filter(only.encode() if only else None)   # (`only`: bracket just these kernels -- every event pair is two packets in the queue)
        L.fib_profile_enable(1)
        L.fib_profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        dt = time.perf_counter() - t0
        L.fib_profile_enable(0)
        L.fib_profile_filter(None)
        if multi:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    class Ctx:
        pass
    ctx = Ctx()
    ctx.args, ctx.dev, ctx.rank, ctx.world, ctx.multi, ctx.force_pg, ctx.shape, ctx.L, ctx.sph = args, dev, rank, world, multi, force_pg, shape, L, sph
    ctx.counts, ctx.v0, ctx.v1, ctx.nloc, ctx.nvox, ctx.timed, ctx.prof_get = counts, v0, v1, nloc, nvox, timed, prof_get

    # ---- headline: GQI + peaks, ONE 140^3 x 270 volume, z-slabs over the ranks --------------------------------------------
    bval, bvec = phantom.scheme_gqi()
    nvol, nvert = len(bval), sph.nvert
    dwi_full, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3, device=dev)      # the same volume on every rank
    dwi = dwi_full[:, v0:v1].contiguous() if world > 1 else dwi_full
    if world > 1:
        del dwi_full
        torch.cuda.empty_cache()
    mask = torch.ones(nloc, dtype=torch.uint8, device=dev)
    plan = fj.OdfPlan("gqi", bval, bvec, sph, sigma=1.25, device=dev.index)
    out = fj.odf_rec_device(plan, dwi, mask, normalize=False)

    def gqi_step():
        if not multi:                                                      # one GPU: qa ./= odfmax inside the library call
            fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
        else:                                                              # slab + ONE all-reduce(MAX) of {odfmax, NaN flag} + qa ./= odfmax
            fd.odf_rec_sharded(plan, dwi, mask, out=out, counts=counts, always=force_pg)

    dt = timed(gqi_step, args.steps, args.warmup, only="odf_gemm")      # the timed steps bracket the roofline's kernel only
    gemm_ms, gemm_n = prof_get("odf_gemm")
    # the step's other launches, from a few more steps with every kernel bracketed (outside the timed region)
    L.fib_profile_enable(1); L.fib_profile_reset()
    for _ in range(5):
        gqi_step()
    sync()
    L.fib_profile_enable(0)
    other_ms = sum(prof_get(k)[0] for k in ("odf_peaks", "odf_post", "mask_compact", "qa_normalize")) / 5.0
    value = nvox * args.steps / dt / 1e6
    step_ms = dt / args.steps * 1e3

    fmt = plan.format                                         # what the plan's kernels run, read back from the library (not from the environment)
    split = fmt != "f32"                                      # fp16x2 (default) / bf16x3: piece products on the f16/bf16 matrix cores, peak scan fused
    nprod = 6 if fmt == "bf16x3" else 3
    gemm_avg_ms = gemm_ms / max(gemm_n, 1)
    flops = 2.0 * nvert * nvol * nloc                         # algorithmic: 173 340 flop/voxel, this rank's voxels per launch
    gemm_bytes = (4.0 * nvol + 1 + 4.0 * nvert + (48 if split else 0)) * nloc   # DWI + mask in, ODF (+ peaks and qa when fused) out
    gbs = gemm_bytes / (gemm_avg_ms * 1e-3) / 1e9 if gemm_n else 0.0
    tfs = flops / (gemm_avg_ms * 1e-3) / 1e12 if gemm_n else 0.0
    if split:
        # two floors: executed piece-product flops (K padded 270 -> 272, 320 of the 321 rows) / 2500 TFLOP/s and algorithmic bytes /
        # 8 TB/s; the roofline is the larger one (3 products: HBM, 0.83 ms against 0.57 ms), the other fraction rides along
        exec_flops = nprod * 2.0 * 320 * 272 * nloc
        t_mfma, t_hbm = exec_flops / (PEAK_BF16_TFLOPS * 1e12), gemm_bytes / (PEAK_HBM_GBS * 1e9)
        kname = "odf_gemm3_kernel<10,1,8,FUSE,%s>: %d MFMA piece products/f32 product + find_peaks! on the accumulators" % ("bf16x3" if nprod == 6 else "fp16x2", nprod)
        mfma_frac = exec_flops / (gemm_avg_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if gemm_n else 0.0
        if t_hbm >= t_mfma:
            roofline = dict(bound="hbm", kernel=kname, achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS, mfma_frac=mfma_frac)
        else:
            roofline = dict(bound="mfma", kernel=kname, achieved=tfs, peak=PEAK_BF16_TFLOPS / nprod, unit="TFLOP/s", frac=tfs / (PEAK_BF16_TFLOPS / nprod),
                            hbm_frac=gbs / PEAK_HBM_GBS)
    else:
        roofline = dict(bound="mfma", kernel="odf_gemm_kernel<10,1> (v_mfma_f32_32x32x2_f32) + separate peak kernel", achieved=tfs, peak=PEAK_F32_TFLOPS,
                        unit="TFLOP/s", frac=tfs / PEAK_F32_TFLOPS, hbm_frac=gbs / PEAK_HBM_GBS)
    roofline.update(algorithmic_bytes=gemm_bytes, avg_kernel_ms=gemm_avg_ms, launches=gemm_n, traffic=None, step_launches_ms=gemm_avg_ms + other_ms)
    if world == 1 and shape == SHAPE:
        tj = legs.stored_traffic()                            # STORED: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes (tools/collect_profiles.sh)
        roofline["traffic"] = tj.get("odf_gemm_bytes_per_launch")
        roofline["traffic_source"] = tj.get("source")

    extra = {}

    def leg(name, fn, into=None):
        try:
            r = fn()
            if into is None:
                extra[name] = r
            else:
                extra.update(r)
        except Exception as e:                                                       # noqa: BLE001
            if multi:                                     # a leg with collectives: one rank skipping it would leave the others waiting -- fail the run
                raise
            extra[name] = dict(error="%s: %s" % (type(e).__name__, e))
            sys.stderr.write("bench.py: leg %s failed: %s\n" % (name, e))

    if not args.no_extra and world == 1:
        leg("host_tier", lambda: legs.host_tier(ctx))                                 # (first: see its docstring)
        if hasattr(legs, "pipeline"):
            leg("pipeline", lambda: legs.pipeline(ctx))
        leg("in_kernel_clock", lambda: legs.in_kernel_clock(ctx))
        try:
            roofline["power"] = legs.power(ctx, gqi_step, gemm_avg_ms, nloc)
        except Exception as e:                                                       # noqa: BLE001
            roofline["power"] = dict(error=str(e))
        leg("gqi_variants", lambda: legs.gqi_variants(ctx, plan, dwi, out, step_ms), into=extra)
    if not args.no_extra:
        if multi:
            del out, dwi
            torch.cuda.empty_cache()
            leg("gqi_weak_scaling", lambda: legs.gqi_weak(ctx, plan, bval, bvec))
        else:
            del out, dwi
        torch.cuda.empty_cache()
        leg("dti_and_c4", lambda: legs.dti_and_c4(ctx), into=extra)
        torch.cuda.empty_cache()
        leg("c5", lambda: legs.c5(ctx), into=extra)
        torch.cuda.empty_cache()
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline_gqi(bval, bvec, sph, seed=3)    # (rank 0 only; at N > 1 the other ranks wait at the barrier below)
        if not args.no_extra and world == 1:
            b2, g2 = phantom.scheme_dti(60, 4, 1000.0, seed=2)
            b5, g5 = phantom.scheme_dsi()
            extra["cpu_baselines"] = dict(dti_fit=cpu_baseline_dti(b2, g2), dsi_rec=cpu_baseline_dsi(b5, g5, sph), stream=cpu_baseline_stream())

    if rank == 0:
        full = dict(metric="Mvoxels/s fit (GQI ODF + peaks, 140^3 x 270-dir); Mpoints/s streamline in extra",
                    value=value, unit="Mvoxels/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=step_ms, higher_is_better=True, scaling="strong", vs_baseline=None,
                    dtype="f32" if not split else ("f32 (operands as 3 exact bf16 pieces on MFMA, f32 accumulate)" if nprod == 6 else
                                                   "f32 (operands as 2 fp16 pieces = 23 bits on MFMA, 3 piece products, f32 accumulate)"),
                    data="synthetic",
                    config=dict(workload="gqi_rec + find_peaks + qa normalisation (gqi.jl:109-201), ONE %dx%dx%d x 270-frame volume "
                                         "(18 x b=5 + 84 dirs x {1000,2000,3000}), sphere_642, all-ones mask" % shape,
                                voxels=nvox, voxels_per_gpu=nloc, frames=nvol, odf_vertices=nvert,
                                parallelism="z-slabs over the ranks (gqi.jl:132) + 2-float all-reduce(MAX) of odfmax (gqi.jl:164)" if world > 1 else "single GPU"),
                    preconditioning_s=PRECOND_S, roofline=roofline, cpu_baseline=cpu, extra=extra)
        try:
            with open(os.path.join(ROOT, "bench_extra.json"), "w") as f:
                json.dump(full, f, indent=1)
        except OSError as e:
            sys.stderr.write("bench.py: cannot write bench_extra.json: %s\n" % e)
        print("bench_extra " + json.dumps(full), flush=True)      # (an EARLIER line that does not start with `{`)
        text = json.dumps(compact_line(full))
        if len(text) > MAX_LINE:                                  # never again a line the driver cannot parse: drop the extras before the contract's keys
            small = compact_line(dict(full, extra={}))
            small["extra_dropped"] = "line was %d bytes" % len(text)
            text = json.dumps(small)
        print(text, flush=True)
    if multi:
        dist.barrier()                                   # (rank 0 may still have been timing the CPU baseline)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
