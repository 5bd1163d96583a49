#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X back end (contract: the task description / DESIGN.md §3).

Metric (BASELINE.json): Mvoxels/s (fit) + Mpoints/s (streamline) on a synthetic 140^3 x 270-direction HCP-like volume
at 1/2/4/8 GPUs.  A "step" = one pass of the GQI hot path (gqi.jl:109-171: ODF contraction on the matrix cores with
find_peaks! fused into its epilogue, exact odfmax, QA normalisation) over ONE resident 140^3 x 270 volume.

N GPUs (one process per GPU, torch.distributed, backend nccl == RCCL over xGMI): the volume is cut into contiguous
z-slabs as the reference threads its z loop (gqi.jl:132); every rank reconstructs its slab; the path's only exchange
step, odfmax = maximum(mean(odf, dims=4)) (gqi.jl:164), is a 2-float all-reduce(MAX) inside the timed region, then qa ./=
odfmax on every rank.  Strong scaling: the total work is fixed.  `value` is whole-job Mvoxels/s with inputs in HBM.
`extra` carries, at every N: the DTI fit (140^3 x 64, slabs), streamline tracking (DTI field all-gathered from the slabs
over RCCL inside the timed step, seeds round-robin), BASELINE config 5 (DSI-515 in slabs with the global odfmax all-reduced,
then the 3-peak field all-gathered and ~10 M seeds x offsets round-robin) and the weak-scaling GQI figure (one whole volume
per rank); at N = 1 also the in-kernel clock (diagnostic build, child process), the microscopy / LCM modes, RUMBA-SD, the
PCIe-inclusive host-tier call and the other CPU baselines."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPE = (140, 140, 140)
PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 MFMA (v_mfma_f32_32x32x2_f32) dense peak
PEAK_BF16_TFLOPS = 2500.0    # MI355X_MICROARCH.md: BF16 MFMA dense peak (v_mfma_f32_32x32x16_bf16, 32 cycles each)
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)


def prof_get(L, name):
    import ctypes as C
    ms, n = C.c_double(0), C.c_int64(0)
    L.fib_profile_get(name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value


# ---- CPU baselines: the oracle (C/OpenMP restatement of the reference's CPU path with its z-slice / seed-chunk threading) on
# the box's host cores, bounded samples of the same workloads.  Every figure is the MEDIAN of NRUNS timed runs of >= RUN_S seconds
# each (a run = one or more calls on the same resident sample) after an untimed warm-up call on that sample (thread pool, input
# pages); the sample is sized from short probe calls: the whole 140^3 volume when a call on it stays within a few seconds, else
# whole slices in multiples of the thread count (one z-slice per thread and trip: the reference's static z-slice threading,
# dti.jl:258 / gqi.jl:132 / dsi.jl:197, stays balanced), else `cores` slices of fewer rows.  Outputs are allocated zero-filled
# inside every call, as the reference's entry points do (`MRI(mask, n, Float32)` -> zeros, mri.jl:251-255). -------------------------
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


def _median_runs(call, t_call):
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
        t0 = time.perf_counter()
        fit(dwi, mask)                                            # warm-up on the final sample
        t = time.perf_counter() - t0
    nv = shp[0] * shp[1] * shp[2]
    med, per, reps = _median_runs(lambda: fit(dwi, mask), t)
    return dict(value=nv / med / 1e6, unit="Mvoxels/s", cores=cores, kind="port",
                runs_mvoxels_per_s=[nv / p / 1e6 for p in per], spread=(max(per) - min(per)) / med,
                sample="%s on a %dx%dx%d sample of the volume, all-ones mask: median of %d runs of %d call(s), %.1f s per run, after a warm-up call"
                       % (label, shp[0], shp[1], shp[2], NRUNS, reps, med * reps))


def cpu_baseline_gqi(bval, bvec, sph, seed):
    from oracle import oracle as orc
    cores = orc.max_threads()
    return _fit_baseline(lambda d, m: orc.gqi_rec(d, m, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=cores),
                         "gqi_rec oracle (C/OpenMP restatement of gqi.jl:109-171, z-slice threads), %d frames" % len(bval),
                         bval, bvec, seed, cores)


def cpu_baseline_dti(bval, bvec):
    from oracle import oracle as orc
    cores = orc.max_threads()
    return _fit_baseline(lambda d, m: orc.dti_fit(d, m, bval, bvec, nthreads=cores),
                         "dti_fit oracle (dti.jl:221-335), %d frames" % len(bval), bval, bvec, 2, cores, nfib=1)


def cpu_baseline_dsi(bval, bvec, sph):
    from oracle import oracle as orc
    cores = orc.max_threads()
    return _fit_baseline(lambda d, m: orc.dsi_rec(d, m, bval, bvec, sph.vertices, sph.faces, 32, nthreads=cores),
                         "dsi_rec oracle (dsi.jl:171-270: 16^3 FFT + trilinear radial integration per voxel), %d frames" % len(bval),
                         bval, bvec, 5, cores)


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
        t = run(sd)
    med, per, reps = _median_runs(lambda: run(sd), t)
    return dict(value=res["points"] / med / 1e6, unit="Mpoints/s", cores=cores, kind="port",
                runs_mpoints_per_s=[res["points"] / p / 1e6 for p in per], spread=(max(per) - min(per)) / med,
                sample="stream oracle (stream.jl:625-790, contiguous seed chunks per thread) from %d seeds (%d z-slices of the ball mask) "
                       "of the 140^3 field, %d points per call: median of %d runs of %d call(s), %.1f s per run, after a warm-up call"
                       % (res["seeds"], nzs, res["points"], NRUNS, reps, med * reps))


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
    # test hook (1-GPU box): FIBERS_BENCH_BACKEND=gloo FIBERS_BENCH_ONE_DEVICE=1 runs N ranks on cuda:0 over gloo, to
    # exercise the multi-rank control flow where RCCL cannot be used (it refuses two ranks on one device)
    backend = os.environ.get("FIBERS_BENCH_BACKEND", "nccl")
    shape = tuple(int(v) for v in os.environ.get("FIBERS_BENCH_SHAPE", "140,140,140").split(","))   # (tests shrink the volume)
    if os.environ.get("FIBERS_BENCH_ONE_DEVICE"):
        local = 0
    # FIBERS_BENCH_FORCE_PG=1 (test hook, 1-GPU box): with ONE rank still create the process group (backend nccl = RCCL) and take
    # every multi-rank branch below -- barrier, all-reduces, the sharded drivers, the field all-gather, the object gather -- so that
    # this file's RCCL code has run on hardware before its first 8-GPU launch
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
    L = fj.lib()
    nx, ny, nz = shape
    nxy = nx * ny
    nvox = nxy * nz
    sph = fj.sphere_642
    z0, z1 = fd.slab_bounds(nz, world, rank, nxy)
    v0, v1 = z0 * nxy, z1 * nxy
    nloc = v1 - v0
    counts = [(b - a) * nxy for a, b in (fd.slab_bounds(nz, world, r, nxy) for r in range(world))]

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    PRECOND_S = 0.15

    def timed(fn, steps, warmup):
        """W untimed steps, then K steps bracketed by barrier + synchronize; max over ranks.
        Before the W warm-up steps the same step runs untimed for about PRECOND_S seconds (a step count derived from the first step's
        duration, identical on every rank): after the idle stretch that precedes every
        section (plan set-up, input generation) the chip needs ~50 ms of load to leave its idle power state -- the step takes 3.1,
        2.70, 2.63, 2.59 ms in its first four blocks of five steps and 2.58 ms from then on (tools/step_evolution.py,
        profiles/r03/step_evolution.txt; extra.gqi_cold_start has this run's ramp) -- and W = 2..5 steps end inside that ramp.
        What is timed is the steady state a stream of volumes sees; the line's `preconditioning` field says so."""
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
        L.fib_profile_enable(1)
        L.fib_profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        dt = time.perf_counter() - t0
        L.fib_profile_enable(0)
        if multi:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

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

    # the ramp out of the idle power state, for the record (untimed as far as `value` goes): ms per step in blocks of five
    cold = []
    torch.cuda.synchronize()
    time.sleep(0.5)
    for _ in range(6):
        t0 = time.perf_counter()
        for _ in range(5):
            gqi_step()
        torch.cuda.synchronize()
        cold.append((time.perf_counter() - t0) / 5 * 1e3)
    dt = timed(gqi_step, args.steps, args.warmup)
    gemm_ms, gemm_n = prof_get(L, "odf_gemm")
    peaks_ms, peaks_n = prof_get(L, "odf_peaks")              # (the separate peak kernel: not launched by the fused path)
    post_ms, post_n = prof_get(L, "odf_post")                 # redo list + exact odfmax + its two floats: one launch
    mc_ms, mc_n = prof_get(L, "mask_compact")
    qn_ms, qn_n = prof_get(L, "qa_normalize")
    value = nvox * args.steps / dt / 1e6

    fmt = plan.format                                         # what the plan's kernels run, read back from the library (not from the environment)
    fused = fmt != "f32"                                      # (sphere_642, aligned volume, a split format: the fused peak scan runs)
    flops = 2.0 * nvert * nvol * nloc                      # algorithmic: 173 340 flop/voxel, this rank's voxels per launch
    gemm_avg_ms = gemm_ms / max(gemm_n, 1)
    achieved = flops / (gemm_avg_ms * 1e-3) / 1e12 if gemm_n else 0.0
    gemm_bytes = (4.0 * nvol + 1 + 4.0 * nvert + (48 if fused else 0)) * nloc   # read DWI + mask, write ODF (+ peaks and qa when fused)
    split = fmt != "f32"
    hbm2 = dict(achieved=gemm_bytes / (gemm_avg_ms * 1e-3) / 1e9 if gemm_n else 0.0, peak=PEAK_HBM_GBS, unit="GB/s",
                algorithmic_bytes=gemm_bytes, frac=gemm_bytes / (gemm_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if gemm_n else 0.0)
    pk = dict(avg_kernel_ms=peaks_ms / max(peaks_n, 1), post_ms=post_ms / max(post_n, 1), mask_compact_ms=mc_ms / max(mc_n, 1),
              qa_normalize_ms=qn_ms / max(qn_n, 1), launches_per_step=4 if fused else 6,
              note=("fused: a step is 4 launches -- mask_compact (voxel list by decoupled look-back + outputs outside the mask), the contraction "
                    "kernel, odf_post (redo list + exact odfmax), qa_normalize") if fused else "separate peak kernel (ODF re-read)")
    exact = fmt == "bf16x3"
    nprod = 6 if exact else 3
    if split:
        # every f32 product = 3 piece products of two fp16 pieces per operand (default; format bf16x3: 6 products of three
        # bf16 pieces): the matrix cores execute nprod x the algorithmic flops of 320 of the 321 rows (K padded 270 -> 272).  Two
        # floors: executed flops / 2500 TFLOP/s and algorithmic bytes / 8 TB/s; the line's roofline is the larger one (with
        # 3 products: HBM, 0.83 ms against 0.57 ms), the other is reported beside it
        exec_flops = nprod * 2.0 * 320 * 272 * nloc
        t_mfma, t_hbm = exec_flops / (PEAK_BF16_TFLOPS * 1e12), gemm_bytes / (PEAK_HBM_GBS * 1e9)
        mfma2 = dict(achieved=achieved, peak=PEAK_BF16_TFLOPS / nprod, unit="TFLOP/s", frac=achieved / (PEAK_BF16_TFLOPS / nprod),
                     executed_tflops=exec_flops / (gemm_avg_ms * 1e-3) / 1e12 if gemm_n else 0.0, floor_ms=t_mfma * 1e3,
                     note="algorithmic f32 flops (2*321*270 per voxel) / kernel time against 2500 TFLOP/s dense / %d piece products per f32 product" % nprod)
        kname = ("odf_gemm3_kernel<MB=10,NX=1,NW=8%s,%s> (%s; 320 rows on MFMA + 1 row on VALU%s)"
                 % (",FUSE" if fused else "", "bf16x3" if exact else "H2",
                    "v_mfma_f32_32x32x16_bf16 on exact 3-way bf16 splits of both f32 operands: 6 piece products per f32 product" if exact else
                    "v_mfma_f32_32x32x16_f16 on two fp16 pieces per f32 operand (23 significant bits, per-voxel power-of-two sample scale): 3 piece products per f32 product",
                    "; find_peaks! + peak/qa extraction on the accumulators" if fused else ""))
        if t_hbm >= t_mfma:
            roofline = dict(bound="hbm", kernel=kname, achieved=hbm2["achieved"], peak=PEAK_HBM_GBS, unit="GB/s", frac=hbm2["frac"],
                            algorithmic_bytes=gemm_bytes, floor_ms=t_hbm * 1e3,
                            note="achieved = algorithmic bytes per launch (4*270 + 1 in, 4*321 + 48 out per voxel: SURVEY 8d) / the kernel's mean duration "
                                 "(hipEvents on the launch stream); the kernel's time includes the peak finder.  Which roof binds: bytes / 8 TB/s = %.2f ms "
                                 "against executed MFMA flops / 2500 TFLOP/s = %.2f ms -> HBM" % (t_hbm * 1e3, t_mfma * 1e3),
                            avg_kernel_ms=gemm_avg_ms, launches=gemm_n, traffic=None, mfma_secondary=mfma2, peaks_kernel=pk)
        else:
            roofline = dict(bound="mfma", kernel=kname, achieved=achieved, peak=PEAK_BF16_TFLOPS / nprod, unit="TFLOP/s", frac=achieved / (PEAK_BF16_TFLOPS / nprod),
                            note="achieved = algorithmic f32 flops (2*321*270 per voxel) / kernel time (the fused kernel's time includes the peak finder); "
                                 "peak = 2500 TFLOP/s dense / %d piece products; executed MFMA rate = %.0f TFLOP/s of 2500 (floors: MFMA %.2f ms, HBM %.2f ms)"
                                 % (nprod, mfma2["executed_tflops"], t_mfma * 1e3, t_hbm * 1e3),
                            avg_kernel_ms=gemm_avg_ms, launches=gemm_n, traffic=None, hbm_secondary=hbm2, peaks_kernel=pk)
    else:
        roofline = dict(bound="mfma", kernel="odf_gemm_kernel<MB=10,NX=1> (v_mfma_f32_32x32x2_f32; 320 rows on MFMA + 1 row on VALU)", achieved=achieved,
                        peak=PEAK_F32_TFLOPS, unit="TFLOP/s", frac=achieved / PEAK_F32_TFLOPS,
                        avg_kernel_ms=gemm_avg_ms, launches=gemm_n, traffic=None, hbm_secondary=hbm2, peaks_kernel=pk)
    tr_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr_file) and world == 1:
        try:
            tj = json.load(open(tr_file))
            roofline["traffic"] = tj.get("odf_gemm_bytes_per_launch")
            roofline["traffic_source"] = ("STORED figure, not measured in this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE in separate passes over "
                                          "tools/prof_step.py gqi (tools/collect_profiles.sh), last collected into %s" % tj.get("source", "profiles/traffic.json"))
        except Exception:
            pass

    extra = {}
    extra["gqi_cold_start"] = dict(ms_per_step_blocks_of_5=cold, note="the headline step right after 0.5 s of idle, six blocks of five steps, before any "
                                                                        "preconditioning: the ramp the `preconditioning` field refers to")
    if not args.no_extra and rank == 0 and world == 1:
        # ---- the boundary a Julia caller pays for (SURVEY 8d "report both"): the fib_* entry points on pageable host arrays, PCIe both ways,
        # every stage of the host tier timed apart (tools/host_tier_probe.py).  This process is the caller: its arrays are first touched by
        # this thread wherever the scheduler put it; the library binds its pinned ring and copy threads to the GPU's NUMA node itself.
        # FIRST among the extras: a process that has just released tens of GB of device memory (torch.cuda.empty_cache() between the legs
        # below) sees its downloads run at 39 instead of 50 GB/s for a few seconds -- the driver is still busy with the released memory
        # (tools/host_tier_state_check.py: 80.8 ms fresh, 101.7 ms right after 120 GB of allocations were released, 81 ms again later).
        # A caller of fib_gqi_rec has not just done that; the legs run before this process has.
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import host_tier_probe as htp
            ht = {}
            ht["gqi_rec"] = htp.leg_odf("gqi", shape=shape, reps=4, dev=dev)
            try:
                mb = np.ascontiguousarray(phantom.ball_mask_torch(shape, dev).reshape(-1).cpu().numpy().astype(np.uint8))
                r_b = htp.leg_odf("gqi", shape=shape, reps=3, mask=mb, dev=dev)
                r_b["voxels_in_mask"] = int(mb.sum())
                r_b["note"] = ("fib_gqi_rec, ball mask (36 % inside): the host tier packs the runs of the voxels inside the mask into the pinned ring and "
                               "zero-fills the gaps on the way back; bytes_in / bytes_out count the voxels that travel")
                ht["gqi_rec_ball_mask"] = r_b
            except Exception as e:                                                          # noqa: BLE001
                ht["gqi_rec_ball_mask"] = dict(error=str(e))
            for name, fn in (("dti_fit", lambda: htp.leg_dti(shape=shape, reps=4, dev=dev)), ("dsi_rec", lambda: htp.leg_odf("dsi", shape=shape, reps=3, dev=dev)),
                             ("stream_c4", lambda: htp.leg_stream(shape=shape, reps=3, dev=dev))):
                try:
                    ht[name] = fn()
                except Exception as e:                                                      # noqa: BLE001
                    ht[name] = dict(error=str(e))
            ht["note"] = ("pcie_floor_ms = max(bytes in, bytes out) / 63 GB/s (Gen5 x16, one direction); with both directions busy this box's link "
                          "moves ~97 GB/s in all (tools/probes/host_probe.hip), i.e. ~49 GB/s each way: the downloads are the pipeline's long pole")
            extra["host_tier"] = ht
            extra["gqi_host_tier"] = {k: v for k, v in ht["gqi_rec"].items() if not isinstance(v, dict)}   # (the key earlier rounds' lines carried)
            if isinstance(ht.get("gqi_rec_ball_mask"), dict):
                extra["gqi_host_tier"]["ball_mask"] = {k: v for k, v in ht["gqi_rec_ball_mask"].items() if not isinstance(v, (dict, list))}
        except Exception as e:                                                              # noqa: BLE001
            extra["host_tier"] = dict(error=str(e))

    if not args.no_extra and world == 1:
        # ---- in-kernel clock of the contraction kernels (MI355X_MICROARCH.md "DVFS give-back" item 6): a child process loads the
        # DIAGNOSTIC build (libfibers_hip_stamp.so: one s_memtime / s_memrealtime pair around each workgroup's work loop) and runs the
        # GQI and DSI steps back to back for 2 s each on the same random phantoms; the product library never executes a stamp ------
        try:
            import subprocess
            if os.path.exists(os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so")) and shape == SHAPE:
                o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_clock.py"), "--seconds", "2.0", "--kernels", "fused,dsi"],
                                   capture_output=True, text=True, timeout=300)
                kc = json.loads([ln for ln in o.stdout.splitlines() if ln.startswith("{")][-1])
                extra["in_kernel_clock"] = kc
                if "gqi_fused" in kc and kc["gqi_fused"].get("clock_ghz_median"):
                    ghz = kc["gqi_fused"]["clock_ghz_median"]
                    roofline["in_kernel_clock_ghz"] = ghz
                    if roofline.get("bound") == "mfma" and split:
                        roofline["frac_of_clock_adjusted_peak"] = roofline["frac"] * 2.4 / ghz
        except Exception as e:                                                      # noqa: BLE001
            extra["in_kernel_clock"] = dict(error=str(e))
        # ---- roofline.power: is "the 1 400 W cap binds" a measured bound?  The step's Joules, LIVE from the board's
        # energy counter around ~2 s of back-to-back steps; the Joules per byte / flop / instruction of the step's ingredients, STORED
        # (tools/energy_model.py runs tools/probes/energy_probe.hip, each ingredient alone under the same counter: profiles/energy_model.json)
        try:
            from fibers_jl_amd import energy as en
            em = en.measure(gqi_step, torch.cuda.synchronize, seconds=2.0)
            if em is None:
                roofline["power"] = dict(error="no board energy counter (librocm_smi64 / rsmi_dev_energy_count_get)")
            else:
                torch.cuda.synchronize()
                idle_w = en.idle_watts(1.2)
                stored = json.load(open(os.path.join(ROOT, "profiles", "energy_model.json")))
                gm = stored.get("gqi_model") or {}
                pw = en.gqi_power_roofline(gm["joules_per_unit"], nloc, gemm_avg_ms, em["ms_per_step"], em["joules_per_step"], idle_w)
                pw.update(board_watts_while_stepping=em["watts"], smu_sclk_mhz_while_stepping=em["sclk_mhz_mean"], steps_measured=em["steps"],
                          source=("measured_joules_per_step, board_watts, idle_w: LIVE (energy counter, this run).  joules_per_unit: STORED, from %s.  "
                                  "floor_ms = (algorithmic HBM bytes + executed MFMA flops, in Joules) / (cap - idle), frac = floor_ms / the kernel's hipEvent "
                                  "time, with every component scaled by calibration_scale so that components + idle = the MEASURED Joules (the probes ran at "
                                  "2.4 GHz and its voltage, the kernel at 1.8-1.9 GHz); floor_raw_ms / frac_raw: the same at what each ingredient costs ALONE at "
                                  "its own clock -- an over-count by ~25 %%, an upper estimate that can exceed 1" % stored.get("source", "profiles/energy_model.json")))
                roofline["power"] = pw
                kc = extra.get("in_kernel_clock", {}).get("gqi_fused") if isinstance(extra.get("in_kernel_clock"), dict) else None
                roofline["clock"] = dict(
                    smu_sclk_mhz_product_kernel=em["sclk_mhz_mean"],
                    in_kernel_ghz_diagnostic_build=kc.get("clock_ghz_median") if kc else None,
                    smu_sclk_mhz_diagnostic_build_same_seconds=kc.get("smu_sclk_mhz_mean") if kc else None,
                    diagnostic_kernel_ms=kc.get("kernel_ms_hipevent") if kc else None,
                    note="the in-kernel clock (s_memtime / s_memrealtime around each workgroup's loop) and the SMU's reported shader clock agree within ~1 % "
                         "when read in the SAME seconds on the SAME build; round 4's 1.875 GHz (in-kernel) against 1.62 GHz (smi) compared the diagnostic build -- "
                         "whose phase marks made it 14 % slower and cooler -- with the product on another box.  The diagnostic build now carries one stamp "
                         "pair per workgroup and runs at the product's speed")
        except Exception as e:                                                      # noqa: BLE001
            roofline["power"] = dict(error=str(e))
        # ---- the same step on the less flattering inputs of SURVEY §8d: ball mask (36 % of the volume inside) and ~1 % of the
        # samples non-positive (exercises the clamp and the mask compaction; the headline uses an all-ones mask, all positive) ----
        bm_h = phantom.ball_mask_torch(shape, dev)
        g = torch.Generator(device=dev); g.manual_seed(17)
        dwi_np = dwi.clone()
        hit = torch.rand(dwi_np.shape, generator=g, device=dev) < 0.01
        dwi_np[hit] = torch.where(torch.rand(int(hit.sum()), generator=g, device=dev) < 0.5, 0.0, -3.0)
        del hit
        out_b = fj.odf_rec_device(plan, dwi_np, bm_h, normalize=True)
        t_b = timed(lambda: fj.odf_rec_device(plan, dwi_np, bm_h, out=out_b, normalize=True), max(2, args.steps // 2), 1) / max(2, args.steps // 2)
        gb_ms, gb_n = prof_get(L, "odf_gemm")
        nin = int(bm_h.sum())
        extra["gqi_ball_mask_nonpositive"] = dict(voxels_in_mask=nin, ms_per_step=t_b * 1e3, mvoxels_in_mask_per_s=nin / t_b / 1e6,
                                                  mvoxels_of_volume_per_s=nvox / t_b / 1e6, gemm_kernel_ms=gb_ms / max(gb_n, 1),
                                                  note="ball mask r = 62 (998 592 voxels), 1 % of the samples set to 0 or -3; cost scales with the mask")
        del dwi_np, out_b, bm_h
        # ---- what ONE of eight ranks would run per step, timed alone on this GPU: rank 0's z-slab of the same volume (nz = 18 of 140),
        # the step of odf_rec_sharded without its collective (mask_compact, contraction, odf_post with the raw {max, flag} pair,
        # qa_normalize from the pair).  No 8-GPU node has run this bench: this bounds the strong-scaling efficiency from the fixed
        # per-step cost alone (ideal = the N = 1 step / 8) ------------------------------------------------------------------------
        try:
            zs0, zs1 = fd.slab_bounds(nz, 8, 0, nxy)
            ns = (zs1 - zs0) * nxy
            counts8 = [(b - a) * nxy for a, b in (fd.slab_bounds(nz, 8, r, nxy) for r in range(8))]
            dwi_s = dwi[:, :ns].contiguous()
            mask_s = torch.ones(ns, dtype=torch.uint8, device=dev)
            out_s = fj.odf_rec_device(plan, dwi_s, mask_s, normalize=False)
            nss = max(4, args.steps)
            t_s = timed(lambda: fd.odf_rec_sharded(plan, dwi_s, mask_s, out=out_s, counts=counts8), nss, 2) / nss
            gs_ms, gs_n = prof_get(L, "odf_gemm")
            extra["gqi_slab_1of8"] = dict(voxels=ns, nz=zs1 - zs0, ms_per_step=t_s * 1e3, gemm_kernel_ms=gs_ms / max(gs_n, 1),
                                          ideal_ms=dt / args.steps * 1e3 * ns / nvox,
                                          efficiency_before_collectives=(dt / args.steps * ns / nvox) / t_s,
                                          note="rank 0 of 8's slab timed alone on one GPU (no collective): ideal = N=1 ms_per_step x slab share")
            del dwi_s, mask_s, out_s
        except Exception as e:                                                      # noqa: BLE001
            extra["gqi_slab_1of8"] = dict(error=str(e))
        # ---- the headline step with the other operand format (the format is a plan parameter): the exact 3 x bf16 split
        # when the line runs the default, the two-piece fp16 form when the line itself was run with FIBERS_ODF_FORMAT=bf16x3 ------------------
        try:
            plan_x = fj.OdfPlan("gqi", bval, bvec, sph, sigma=1.25, device=dev.index, format="fp16x2" if exact else "bf16x3")
            assert plan_x.format == ("fp16x2" if exact else "bf16x3")
            out_x = fj.odf_rec_device(plan_x, dwi, mask, normalize=True)
            t_x = timed(lambda: fj.odf_rec_device(plan_x, dwi, mask, out=out_x, normalize=True), args.steps, 1) / args.steps
            gx_ms, gx_n = prof_get(L, "odf_gemm")
            den = out["odf"].abs().amax(dim=0).clamp_min(1e-30)
            dmax = float(((out_x["odf"] - out["odf"]).abs().amax(dim=0) / den).max())
            same_pk = float((out_x["peak"][0] == out["peak"][0]).all(dim=0).float().mean())
            extra["gqi_exact_split" if not exact else "gqi_fp16_pieces"] = dict(
                ms_per_step=t_x * 1e3, mvoxels_per_s=nvox / t_x / 1e6, gemm_kernel_ms=gx_ms / max(gx_n, 1),
                odf_max_difference_of_voxel_max=dmax, first_peak_identical_fraction=same_pk,
                note="the same step, same inputs, with %s; differences between the two formats' outputs relative to each voxel's ODF maximum "
                     "(tests: <= 3e-6, peaks identical except ties; both are ~1e-6 from a float64 contraction, tools/gemm_accuracy.py)"
                     % ("three exact bf16 pieces per operand, 6 MFMAs per block and 16 frames (format bf16x3)" if not exact else "two fp16 pieces per operand, 3 MFMAs per block and 16 frames (the default)"))
            del out_x
            plan_x.close()
        except Exception as e:                                                      # noqa: BLE001
            extra["gqi_exact_split"] = dict(error=str(e))
    if not args.no_extra:
        # ---- weak-scaling figure of the same step: one whole volume per rank, odfmax all-reduced -----------------------------
        if multi:
            del out, dwi
            torch.cuda.empty_cache()
            dwi_w, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3 + rank, device=dev)
            mask_w = torch.ones(nvox, dtype=torch.uint8, device=dev)
            out_w = fj.odf_rec_device(plan, dwi_w, mask_w, normalize=False)
            nst = max(2, args.steps // 2)
            t_w = timed(lambda: fd.odf_rec_sharded(plan, dwi_w, mask_w, out=out_w, counts=[nvox] * world, always=force_pg), nst, 1)
            extra["gqi_weak_scaling"] = dict(mvoxels_per_s=world * nvox * nst / t_w / 1e6, ms_per_step=t_w / nst * 1e3,
                                             note="one whole 140^3 x 270 volume per rank, 2-float all-reduce(MAX) inside the step")
            del dwi_w, out_w, mask_w
        else:
            del out, dwi
        torch.cuda.empty_cache()
        # ---- C2: DTI fit, 140^3 x 64, z-slabs (no exchange step) --------------------------------------------------------------
        b2, g2 = phantom.scheme_dti(60, 4, 1000.0, seed=2)
        d2f, _ = phantom.make_dwi_torch(shape, b2, g2, seed=2, device=dev, nfib=1)
        d2 = d2f[:, v0:v1].contiguous() if world > 1 else d2f
        del d2f
        p2 = fj.DtiPlan(b2, g2, device=dev.index)
        o2 = fj.dti_fit_device(p2, d2, mask)
        t_dti = timed(lambda: fj.dti_fit_device(p2, d2, mask, out=o2), args.steps, 1) / args.steps
        k_ms, k_n = prof_get(L, "dti_fit")
        dbytes = (4.0 * len(b2) + 1 + 64) * nloc
        extra["dti_fit_140x64"] = dict(mvoxels_per_s=nvox / t_dti / 1e6, ms_per_step=t_dti * 1e3,
                                       kernel_ms=k_ms / max(k_n, 1), algorithmic_bytes=dbytes,
                                       hbm_gbs=dbytes / (k_ms / max(k_n, 1) * 1e-3) / 1e9 if k_n else 0.0,
                                       hbm_frac=dbytes / (k_ms / max(k_n, 1) * 1e-3) / 1e9 / PEAK_HBM_GBS if k_n else 0.0,
                                       note="one volume in z-slabs over the ranks; per-kernel figures are rank 0's slab")
        # ---- C4: streamlines from the DTI principal eigenvector, ball mask: the slab's field is all-gathered over RCCL inside
        # the timed step (the path's only bulk collective: 16 B/voxel), seeds round-robin, no collective after ----------------------
        bm_full = phantom.ball_mask_torch(shape, dev)
        field_loc, mout_loc = fj.stream_field_device([o2["eigvec1"]], fa=o2["fa"], fa_thresh=0.1, mask=bm_full[v0:v1].contiguous())
        mout = fd.allgather_slabs(mout_loc, counts, always=force_pg)
        seeds_all = torch.nonzero(mout).flatten()
        sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
        xyz_buf = {}

        def xyz_out(npnt):                                                   # steady-state output buffer (no per-call allocation)
            if xyz_buf.get("t") is None or xyz_buf["t"].numel() < 3 * npnt:
                xyz_buf["t"] = torch.empty(int(3 * npnt * 1.05) + 16, dtype=torch.float32, device=dev)
            return xyz_buf["t"]
        res = {}

        sbuf4 = fj.StreamBuffers(dev) if not multi else None

        def stream_step():
            field = fd.allgather_slabs(field_loc, counts, always=force_pg)   # shared peak field over xGMI
            if multi:
                res["r"] = fd.stream_sharded(field, shape, seeds_all, sub, xyz_out=xyz_out)
            else:                                                            # one GPU: the one-call form into kept buffers (as the C5 section below)
                res["r"] = fj.stream_device_run(field, shape, seeds_all, sub, buffers=sbuf4)
        nst = max(2, args.steps // 2)
        t_st = timed(stream_step, nst, 2)
        r = res["r"]
        cnt = torch.tensor([float(r["xyz"].shape[0]), float(r["npts"].numel())], device=dev, dtype=torch.float64)
        if multi:
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        npoints, nlines, t_st = int(cnt[0].item()), int(cnt[1].item()), t_st / nst
        tr_ms, tr_n = prof_get(L, "stream_trace")
        pk_ms, pk_n = prof_get(L, "stream_pack")
        sc_ms, sc_n = prof_get(L, "stream_scan")
        extra["stream_dti_ball"] = dict(seeds=int(seeds_all.numel()), lines=nlines, points=npoints,
                                        mpoints_per_s=npoints / t_st / 1e6, ms_per_step=t_st * 1e3,
                                        trace_kernel_ms=tr_ms / max(tr_n, 1), pack_kernel_ms=pk_ms / max(pk_n, 1),
                                        kernel_sum_ms=(tr_ms + pk_ms + sc_ms) / max(tr_n, 1),
                                        algorithmic_bytes=25.0 * npoints,
                                        hbm_gbs_trace=25.0 * (npoints / world) / (tr_ms / max(tr_n, 1) * 1e-3) / 1e9 if tr_n else 0.0,
                                        roofline=dict(bound="hbm", achieved=25.0 * (npoints / world) / ((tr_ms + pk_ms + sc_ms) / max(tr_n, 1) * 1e-3) / 1e9 if tr_n else 0.0,
                                                      peak=PEAK_HBM_GBS, unit="GB/s",
                                                      frac=25.0 * (npoints / world) / ((tr_ms + pk_ms + sc_ms) / max(tr_n, 1) * 1e-3) / 1e9 / PEAK_HBM_GBS if tr_n else 0.0,
                                                      note="25 B per emitted point (SURVEY 8d, nvec = 1) x rank 0's points / device time of trace + scan + pack; the "
                                                           "trace bound by its point stores and vector-ALU issue, pack by HBM (DESIGN.md K6 [r5])"),
                                        note="one volume; wall = field all-gather + trace + scan + pack into kept buffers (+ one host sync for the counts): one GPU "
                                             "fibd_stream_run, N > 1 trace + pack per rank with the seeds dealt round-robin")
        # the one-call form (fibd_stream_run: straight into buffers kept between calls) on the same field and seeds, for the record
        if rank == 0 and world == 1:
            try:
                sbuf = fj.StreamBuffers(dev)
                f_once = fd.allgather_slabs(field_loc, counts)
                fj.stream_device_run(f_once, shape, seeds_all, sub, buffers=sbuf)
                t_run = timed(lambda: fj.stream_device_run(f_once, shape, seeds_all, sub, buffers=sbuf), nst, 1) / nst
                extra["stream_dti_ball"]["one_call_form"] = dict(ms_per_step=t_run * 1e3, mpoints_per_s=npoints / t_run / 1e6,
                                                                 note="fibd_stream_run into kept buffers: trace + scan + pack in one call (1 M lines: below the 2^21 "
                                                                      "lines from which the fused kernel is used -- tools/stream_fused_ab.py, profiles/r05/negative_results.txt)")
                # .. and without the host round trip at the end of every call (fibd_stream_run_enqueue: the counts stay on the device)
                cnt2 = torch.zeros(2, dtype=torch.int64, device=dev)
                t_enq = timed(lambda: fj.stream_device_run_enqueue(f_once, shape, seeds_all, sub, sbuf, counts=cnt2), nst, 1) / nst
                torch.cuda.synchronize()
                ke_ms, ke_n = prof_get(L, "stream_trace")
                kp_ms, kp_n = prof_get(L, "stream_pack")
                ks_ms, _ = prof_get(L, "stream_scan")
                extra["stream_dti_ball"]["enqueue_form"] = dict(ms_per_step=t_enq * 1e3, mpoints_per_s=int(cnt2[1]) / t_enq / 1e6, lines=int(cnt2[0]), points=int(cnt2[1]),
                                                                kernel_sum_ms=(ke_ms + kp_ms + ks_ms) / max(ke_n, 1),
                                                                note="fibd_stream_run_enqueue: the same launches, {lines, points} written by the stream to device memory, no "
                                                                     "synchronisation inside the call -- back-to-back calls keep the GPU busy (the synchronising forms idle it for "
                                                                     "the 16-byte download of the counts and the next call's launch latency)")
                del sbuf, f_once
            except Exception as e:                                                  # noqa: BLE001
                extra["stream_dti_ball"]["one_call_form"] = dict(error=str(e))
        # the trilinear option (fib_stream_params.interp = 1; not in the reference) on the same field and seeds, rank 0's share
        if rank == 0:
            field_all = fd.allgather_slabs(field_loc, counts) if world == 1 else None
            if field_all is not None:
                rt = {}

                def tri_step():
                    rt["r"] = fj.stream_device(field_all, shape, seeds_all, sub, xyz_out=xyz_out, interp="trilinear")
                L.fib_profile_reset()
                t_tri = timed(tri_step, nst, 1) / nst
                tt_ms, tt_n = prof_get(L, "stream_trace")
                npt = int(rt["r"]["xyz"].shape[0])
                extra["stream_dti_ball_trilinear"] = dict(lines=int(rt["r"]["npts"].numel()), points=npt, mpoints_per_s=npt / t_tri / 1e6,
                                                          ms_per_step=t_tri * 1e3, trace_kernel_ms=tt_ms / max(tt_n, 1),
                                                          note="interp = trilinear: 8 x the field reads per step, same integrator")
                del rt, field_all
            # ---- divergent termination: the bundle phantom (Voronoi bundles, lines end where bundles meet at > 45 degrees: broad
            # length distribution); the share of lane-steps that idle because a wave runs as long as its longest line -----------------
            if world == 1:
                ovb, mb = phantom.bundle_field_torch(shape, dev)
                fb, mob = fj.stream_field_device([ovb], mask=mb)
                sb = torch.nonzero(mob).flatten()
                div = {}
                for nsub_b in (1, 10):
                    subb = torch.from_numpy(fj.make_sublist(nsub_b, np.random.default_rng(5))).to(dev) if nsub_b > 1 else sub
                    rb = {}

                    def bstep():
                        rb["r"] = fj.stream_device(fb, shape, sb, subb, want_all_npts=True, xyz_out=xyz_out)
                    L.fib_profile_reset()
                    t_b = timed(bstep, 3, 1) / 3
                    tb_ms, tb_n = prof_get(L, "stream_trace")
                    nall = rb["r"]["all_npts"].cpu().numpy().astype(np.int64)
                    it = nall + 2                                            # loop trips of a lane: its points + the two failed steps
                    w = np.concatenate([it, np.zeros((-len(it)) % 64, np.int64)]).reshape(-1, 64)
                    div["nsub%d" % nsub_b] = dict(
                        lines=int(len(nall)), points=int(rb["r"]["xyz"].shape[0]), npts_median=float(np.median(nall)), npts_max=int(nall.max()),
                        static_lane_idle_frac=float(1.0 - it.sum() / (w.max(1).sum() * 64.0)),
                        trace_kernel_ms=tb_ms / max(tb_n, 1), ms_per_step=t_b * 1e3, mpoints_per_s=int(rb["r"]["xyz"].shape[0]) / t_b / 1e6)
                    del rb
                div["note"] = ("static_lane_idle_frac: share of lane-steps idle when a wave runs as long as its longest line.  Four compaction / refill "
                               "forms of the tracer were built, bit-identical, and measured slower (profiles/r03/trace_compaction.log, DESIGN.md K6): "
                               "one lane per line is final")
                extra["stream_bundle_divergent"] = div
                del ovb, mb, fb, mob, sb
        del res, r
        if world == 1:
            field = field_loc
            # ---- microscopy regime (stream.jl:547-619) on the same field: every 8th seed, reference defaults ----------
            sm = seeds_all[::8].contiguous()
            z1_ = torch.zeros((1, 3), dtype=torch.float32, device=dev)
            kw = dict(ang_thresh=20, step_size=1.0, smooth_coeff=0.0, search_dist=15, search_ang=10, xyz_out=xyz_out)
            rm = fj.stream_device(field, shape, sm, z1_, **kw)
            torch.cuda.synchronize()
            L.fib_profile_enable(1); L.fib_profile_reset()
            t0 = time.perf_counter()
            rm = fj.stream_device(field, shape, sm, z1_, **kw)
            torch.cuda.synchronize()
            t_m = time.perf_counter() - t0
            L.fib_profile_enable(0)
            mk_ms, mk_n = prof_get(L, "stream_trace_micro")
            npm = int(rm["xyz"].shape[0])
            # per emitted point the reference visits the 31^3 search cube; 15 939 of its cells lie in the search ball
            extra["stream_micro_ball"] = dict(seeds=int(sm.numel()), lines=int(rm["npts"].numel()), points=npm,
                                              mpoints_per_s=npm / t_m / 1e6, ms_per_step=t_m * 1e3,
                                              trace_kernel_ms=mk_ms / max(mk_n, 1),
                                              search_cells_per_s=npm * 29791.0 / (mk_ms / max(mk_n, 1) * 1e-3) if mk_n else 0.0,
                                              note="search_dist 15, search_ang 10, ang_thresh 20, step 1 (reference defaults of the regime)")
            del rm
            # ---- LCM-guided tracking (stream.jl:380-495) on a synthetic 2-D section: 2048^2 pixels, 3 orientations each ----
            n2 = 2048
            g = torch.Generator(device=dev); g.manual_seed(11)
            # 2-D orientation ANGLES, as the reference's microscopy data come (stream.jl:147-172): three per pixel, radians in
            # [-pi/2, pi/2]; the through-plane dimension is the one with the largest voxel size (z)
            ang = [((torch.rand(n2 * n2, device=dev, generator=g) - 0.5 + k * 3.14159265 / 3 + 1.5707963) % 3.14159265) - 1.5707963 for k in range(3)]
            ov2 = [fj.angles_to_vectors_device(a_.clamp(-1.5707963, 1.5707963), volres=(0.5, 0.5, 2.0))[0] for a_ in ang]
            lc = torch.rand((10, n2 * n2), device=dev, generator=g)
            fld, mo = fj.stream_field_device(ov2, mask=torch.ones(n2 * n2, dtype=torch.uint8, device=dev))
            sd2 = torch.nonzero(mo).flatten()
            s2 = torch.tensor([[0.1, -0.2, 0.0]], dtype=torch.float32, device=dev)
            kw = dict(lcms=lc, lcm_thresh=0.099, strdims=(0, 1), rng_seed=7, len_max=140, xyz_out=xyz_out)
            rl = fj.stream_device(fld, (n2, n2, 1), sd2, s2, **kw)
            torch.cuda.synchronize()
            L.fib_profile_enable(1); L.fib_profile_reset()
            t0 = time.perf_counter()
            rl = fj.stream_device(fld, (n2, n2, 1), sd2, s2, **kw)
            torch.cuda.synchronize()
            t_l = time.perf_counter() - t0
            L.fib_profile_enable(0)
            lk_ms, lk_n = prof_get(L, "stream_trace_lcm")
            npl = int(rl["xyz"].shape[0])
            extra["stream_lcm_2d"] = dict(seeds=int(sd2.numel()), lines=int(rl["npts"].numel()), points=npl,
                                          mpoints_per_s=npl / t_l / 1e6, ms_per_step=t_l * 1e3,
                                          trace_kernel_ms=lk_ms / max(lk_n, 1), flagged_fraction=float(rl["flags"].float().mean()),
                                          note="2048x2048x1 pixels, 3 orientation ANGLES (radians, expanded as stream.jl:147-172 does) + one 10-element LCM per pixel, len_max 140")
            del rl, fld, lc, ov2, ang
            # ---- RUMBA-SD (rusd.jl, row N4): 140^3 x 270 frames, ball mask, sphere_724 (364 compartments), 10 iterations ----
            torch.cuda.empty_cache()
            b4, g4 = phantom.scheme_gqi()
            d4, _ = phantom.make_dwi_torch(shape, b4, g4, seed=3, device=dev)
            rp = fj.RumbaPlan(b4, g4, fj.sphere_724, device=dev.index)
            fj.rumba_rec_device(rp, d4, bm_full, shape, niter=2)
            torch.cuda.synchronize()
            L.fib_profile_enable(1); L.fib_profile_reset()
            nit = 10
            t0 = time.perf_counter()
            rr = fj.rumba_rec_device(rp, d4, bm_full, shape, niter=nit)
            torch.cuda.synchronize()
            t_r = time.perf_counter() - t0
            L.fib_profile_enable(0)
            gm_ms, gm_n = prof_get(L, "matrix_gemm")
            tv_ms, tv_n = prof_get(L, "rumba_tv")
            el_ms, el_n = prof_get(L, "rumba_elementwise")
            nmask = int(bm_full.sum())
            kk, _nd = rp.kernel().shape[1], rp.kernel().shape[0]
            extra["rumba_140_ball"] = dict(voxels=nmask, compartments=kk, dirs=_nd, iterations=nit, ms_total=t_r * 1e3,
                                           ms_per_iteration=(gm_ms + tv_ms + el_ms) / nit,
                                           gemm_ms_per_iteration=gm_ms / nit, tv_ms_per_iteration=tv_ms / nit,
                                           elementwise_ms_per_iteration=el_ms / nit,
                                           gemm_tflops=3 * 2.0 * kk * _nd * nmask * nit / (gm_ms * 1e-3) / 1e12 if gm_n else 0.0,
                                           snr_mean=rr["snr_mean"],
                                           note="three [364 x 253] contractions per iteration on the split-bf16 MFMA kernel; 600 iterations in the reference's default")
            del rr, d4, rp
        del field_loc, mout_loc, seeds_all, o2, d2

    if not args.no_extra:
        # ---- C5 (BASELINE config 5) at every N: DSI 515-direction reconstruction in z-slabs (dsi.jl:197) with the global odfmax
        # all-reduced (dsi.jl:263); then the 3-peak field + mask all-gathered over RCCL inside the timed step and ~10 M seeds x
        # offsets round-robin over the ranks (stream.jl:757-761) --------------------------------------------------------------------
        torch.cuda.empty_cache()
        b5, g5 = phantom.scheme_dsi()
        d5f, _ = phantom.make_dwi_torch(shape, b5, g5, seed=5, device=dev)
        d5 = d5f[:, v0:v1].contiguous() if world > 1 else d5f
        del d5f
        torch.cuda.empty_cache()
        p5 = fj.OdfPlan("dsi", b5, g5, sph, hann_width=32, device=dev.index)
        o5 = fj.odf_rec_device(p5, d5, mask, normalize=False)
        nd = max(2, args.steps // 2)

        def dsi_step():
            if not multi:
                fj.odf_rec_device(p5, d5, mask, out=o5, normalize=True)
            else:
                fd.odf_rec_sharded(p5, d5, mask, out=o5, counts=counts, always=force_pg)
        t_dsi = timed(dsi_step, nd, 1) / nd
        g_ms, g_n = prof_get(L, "odf_gemm")
        f_ms, f_n = prof_get(L, "dsi_fold")
        q_ms, q_n = prof_get(L, "odf_post")
        n5 = len(b5)
        dsi_bytes = (4.0 * n5 + 1 + 4.0 * n5 + 4.0 * nvert + 48) * nloc          # 5 456 B / voxel (DWI + mask in; pdf, odf, peaks, qa out)
        dsi_k_ms = g_ms / max(g_n, 1)
        nprod5 = 6 if p5.format == "bf16x3" else 3
        dsi_exec = nprod5 * 2.0 * (320 + 288) * 272 * nloc                       # executed MFMA flops: piece products x (10 + 9 blocks) x 32 rows x 17 stages x 16
        extra["dsi_rec_140x515"] = dict(mvoxels_per_s=nvox / t_dsi / 1e6, ms_per_step=t_dsi * 1e3,
                                        gemm_kernel_ms=dsi_k_ms, fold_kernel_ms=f_ms / max(f_n, 1), peaks_kernel_ms=q_ms / max(q_n, 1),
                                        roofline=dict(bound="hbm" if nprod5 == 3 else "mfma",
                                                      kernel="odf_dsi2_kernel<9>: fused ODF tile (10 blocks + pole row, find_peaks on the accumulators) + pdf tile "
                                                             "(9 blocks) per voxel group, antipodal fold inside the sample load, %d piece products per f32 product" % nprod5,
                                                      achieved=(dsi_bytes / (dsi_k_ms * 1e-3) / 1e9 if nprod5 == 3 else dsi_exec / (dsi_k_ms * 1e-3) / 1e12) if g_n else 0.0,
                                                      peak=PEAK_HBM_GBS if nprod5 == 3 else PEAK_BF16_TFLOPS, unit="GB/s" if nprod5 == 3 else "TFLOP/s",
                                                      frac=(dsi_bytes / (dsi_k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if nprod5 == 3 else dsi_exec / (dsi_k_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if g_n else 0.0,
                                                      note="floors: algorithmic bytes / 8 TB/s = %.2f ms, executed MFMA flops / 2500 TFLOP/s = %.2f ms; achieved = the binding "
                                                           "quantity / the kernel's hipEvent time" % (dsi_bytes / (PEAK_HBM_GBS * 1e9) * 1e3, dsi_exec / (PEAK_BF16_TFLOPS * 1e12) * 1e3),
                                                      mfma_secondary=dict(achieved=dsi_exec / (dsi_k_ms * 1e-3) / 1e12 if g_n else 0.0, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                                                                          frac=dsi_exec / (dsi_k_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if g_n else 0.0,
                                                                          note="executed dense MFMA flops of the kernel / its hipEvent time"),
                                                      hbm_secondary=dict(algorithmic_bytes=dsi_bytes, achieved=dsi_bytes / t_dsi / 1e9 if t_dsi else 0.0,
                                                                         peak=PEAK_HBM_GBS, unit="GB/s", frac=dsi_bytes / t_dsi / 1e9 / PEAK_HBM_GBS,
                                                                         note="algorithmic bytes of the whole step / step wall time")),
                                        note="ONE volume in z-slabs over the ranks; folded lattice: 258 folded samples x (258 pdf + 321 odf) rows; per-kernel figures are rank 0's slab")
        # ---- C5 tracking: 3 peaks per voxel (f = qa, f_thresh = .03), ball mask, nsub = 10 -> ~10 M lines ---------------------------
        del d5
        bm_full5 = phantom.ball_mask_torch(shape, dev)
        f3_loc, m3_loc = fj.stream_field_device(o5["peak"], f=o5["qa"], f_thresh=0.03, mask=bm_full5[v0:v1].contiguous())
        mout3 = fd.allgather_slabs(m3_loc, counts, always=force_pg)
        seeds3 = torch.nonzero(mout3).flatten()
        sub10 = torch.from_numpy(fj.make_sublist(10, np.random.default_rng(5))).to(dev)
        xyz5 = {}

        def xyz_out5(npnt):
            if xyz5.get("t") is None or xyz5["t"].numel() < 3 * npnt:
                xyz5["t"] = torch.empty(int(3 * npnt * 1.05) + 16, dtype=torch.float32, device=dev)
            return xyz5["t"]
        r3 = {}

        sbuf5 = fj.StreamBuffers(dev) if not multi else None

        def c5_step():
            field3 = fd.allgather_slabs(f3_loc, counts, always=force_pg)       # the shared 3-peak field over xGMI (48 B / voxel)
            if multi:
                r3["r"] = fd.stream_sharded(field3, shape, seeds3, sub10, xyz_out=xyz_out5)
            else:                                                              # one GPU: the one-call form into kept buffers (fibd_stream_run: 10 M lines ->
                r3["r"] = fj.stream_device_run(field3, shape, seeds3, sub10, buffers=sbuf5)   # the fused trace + look-back + pack kernel)
        t3 = timed(c5_step, 3, 2) / 3
        tr_ms, tr_n = prof_get(L, "stream_trace")
        pk_ms, pk_n = prof_get(L, "stream_pack")
        sc_ms, sc_n = prof_get(L, "stream_scan")
        cnt3 = torch.tensor([float(r3["r"]["xyz"].shape[0]), float(r3["r"]["npts"].numel())], device=dev, dtype=torch.float64)
        if multi:
            dist.all_reduce(cnt3, op=dist.ReduceOp.SUM)
        np3, nl3 = int(cnt3[0].item()), int(cnt3[1].item())
        ksum3 = (tr_ms + pk_ms + sc_ms) / max(tr_n, 1)
        extra["stream_dsi_3peaks_10M"] = dict(seeds=int(seeds3.numel()), nsub=10, lines=nl3, points=np3,
                                              mpoints_per_s=np3 / t3 / 1e6, ms_per_step=t3 * 1e3,
                                              trace_kernel_ms=tr_ms / max(tr_n, 1), pack_kernel_ms=pk_ms / max(pk_n, 1),
                                              kernel_sum_ms=ksum3, algorithmic_bytes=49.0 * np3,
                                              roofline=dict(bound="hbm", achieved=49.0 * (np3 / world) / (ksum3 * 1e-3) / 1e9 if tr_n else 0.0, peak=PEAK_HBM_GBS, unit="GB/s",
                                                            frac=49.0 * (np3 / world) / (ksum3 * 1e-3) / 1e9 / PEAK_HBM_GBS if tr_n else 0.0,
                                                            note="49 B per emitted point (SURVEY 8d, nvec = 3) x rank 0's points / device time of trace + scan + pack"),
                                              note="wall = field all-gather + tracking into kept buffers (+ one host sync for the counts); one GPU: fibd_stream_run, which from "
                                                   "2^21 lines on is ONE kernel (the workgroup that traced 512 lines packs them behind a decoupled look-back: no scan, no "
                                                   "pack launch -- pack_kernel_ms 0); N > 1: trace + scan + pack per rank, seeds x offsets round-robin; kernel_sum = device "
                                                   "time of the tracking kernels on rank 0")
        if not multi:                                                          # the same without the host round trip at the end of every call
            try:
                field3e = fd.allgather_slabs(f3_loc, counts)
                cnt5 = torch.zeros(2, dtype=torch.int64, device=dev)
                t5e = timed(lambda: fj.stream_device_run_enqueue(field3e, shape, seeds3, sub10, sbuf5, counts=cnt5), 3, 1) / 3
                torch.cuda.synchronize()
                extra["stream_dsi_3peaks_10M"]["enqueue_form"] = dict(ms_per_step=t5e * 1e3, mpoints_per_s=int(cnt5[1]) / t5e / 1e6, lines=int(cnt5[0]), points=int(cnt5[1]),
                                                                      note="fibd_stream_run_enqueue on the gathered field: {lines, points} stay on the device, no synchronisation inside the call")
                del field3e
            except Exception as e:                                                  # noqa: BLE001
                extra["stream_dsi_3peaks_10M"]["enqueue_form"] = dict(error=str(e))
        del o5, r3, f3_loc, m3_loc, mout3, seeds3, xyz5
        torch.cuda.empty_cache()
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline_gqi(bval, bvec, sph, seed=3)    # (rank 0 only; at N > 1 the other ranks wait at the barrier below)
        if not args.no_extra and world == 1:
            b2, g2 = phantom.scheme_dti(60, 4, 1000.0, seed=2)
            b5, g5 = phantom.scheme_dsi()
            extra["cpu_baselines"] = dict(dti_fit=cpu_baseline_dti(b2, g2), dsi_rec=cpu_baseline_dsi(b5, g5, sph), stream=cpu_baseline_stream(),
                                          note="the oracle (C/OpenMP restatement of the reference's CPU path, the reference's own decomposition) on this box's "
                                               "host cores; Julia itself cannot run here")

    if rank == 0:
        line = dict(metric="Mvoxels/s fit (GQI ODF + peaks, 140^3 x 270-dir); Mpoints/s streamline in extra",
                    value=value, unit="Mvoxels/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="strong", vs_baseline=None,
                    dtype="f32" if not split else ("f32 (exact 3xbf16 operand splits on the bf16 matrix cores, f32 accumulate)" if exact else
                                                   "f32 (f32 in, f32 accumulate, f32 out; inside the contraction every operand travels as two fp16 pieces = 23 significant "
                                                   "bits with a per-voxel power-of-two scale, 3 exact piece products per f32 product on the f16 matrix cores; measured "
                                                   "against a float64 contraction the result is closer than the exact 3xbf16 split's and than an f32 fma chain's: "
                                                   "profiles/r03/gemm_accuracy.txt; format bf16x3 (FIBERS_ODF_FORMAT) selects the exact split, timed in extra.gqi_exact_split)"), data="synthetic",
                    config=dict(workload="gqi_rec + find_peaks + qa normalisation, ONE %dx%dx%d x 270-frame volume "
                                         "(18 x b=5 + 84 dirs x {1000,2000,3000}), sphere_642, mask = all ones" % shape,
                                voxels=nvox, voxels_per_gpu=nloc, frames=nvol, odf_vertices=nvert,
                                parallelism="contiguous z-slabs over the ranks (gqi.jl:132), 2-float all-reduce(MAX) of odfmax (gqi.jl:164)" if world > 1 else "single GPU"),
                    preconditioning="every timed section is preceded by %.2f s of the same step, untimed, then the W warm-up steps, then exactly K timed steps "
                                    "(the chip leaves its idle power state over ~50 ms of load; extra.gqi_cold_start, tools/step_evolution.py)" % PRECOND_S,
                    roofline=roofline, cpu_baseline=cpu, extra=extra)
        print(json.dumps(line))
    if multi:
        dist.barrier()                                   # (rank 0 may still have been timing the CPU baseline)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
