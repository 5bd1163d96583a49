#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X back end (contract: see the task description / DESIGN.md §Measurement).

Metric (BASELINE.json): Mvoxels/s (fit) + Mpoints/s (streamline) on a synthetic 140^3 x 270-direction
HCP-like volume.  A "step" = one pass of the GQI reconstruction hot path (ODF GEMM on FP32 MFMA + ODF peak
finder + global QA normalisation; gqi.jl:109-171) over one resident 140^3 x 270 volume per rank; `value` is
whole-job Mvoxels/s with inputs already in HBM.  The same JSON line carries the DTI fit (140^3 x 64) and the
streamline tracker (DTI-like field, ball mask, ~1 M seeds) as `extra`, the roofline of the dominant kernel
and the CPU baseline (the oracle = C restatement of the reference CPU path, bounded sample).

N>1: one process per GPU (torch.distributed, backend nccl == RCCL); every rank reconstructs its own volume
(weak scaling; voxels are independent) and the only exchange step of the path — odfmax = max over all
voxels of mean(odf) (gqi.jl:164) — is a 1-float all-reduce(MAX) inside the timed region."""
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


def cpu_baseline_gqi(bval, bvec, sph, seed, target_s=15.0):
    """the oracle (C restatement of the reference CPU path, z-slice threading) on a bounded z-slab"""
    from oracle import oracle as orc
    from fibers_jl_amd import phantom
    cores = orc.max_threads()
    nx, ny = SHAPE[0], SHAPE[1]

    def run(nz):
        dwi, _, _ = phantom.make_volume((nx, ny, nz), bval, bvec, seed, noise_frac=0.02, crossing=True)
        mask = np.ones((nx, ny, nz), np.uint8)
        t0 = time.perf_counter()
        orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=cores)
        return time.perf_counter() - t0

    t_probe = run(cores)                                 # one slice per thread
    nz = int(max(cores, min(SHAPE[2], round(cores * target_s / max(t_probe, 1e-3) / cores) * cores)))
    t = run(nz)
    reps = 1
    while nz == SHAPE[2] and t * reps < target_s and reps < 8:      # the whole volume is short of the sample: repeat it
        t = min(t, run(nz)) if False else (t * reps + run(nz)) / (reps + 1)
        reps += 1
    nvox = nx * ny * nz
    return dict(value=nvox / t / 1e6, unit="Mvoxels/s", cores=cores, kind="port",
                sample="gqi_rec oracle (C/OpenMP restatement of gqi.jl:109-171, z-slice threads) on a %dx%dx%d x %d-frame slab, "
                       "%d pass(es), %.1f s each" % (nx, ny, nz, len(bval), reps, t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (1-GPU box): FIBERS_BENCH_BACKEND=gloo FIBERS_BENCH_ONE_DEVICE=1 runs N ranks on cuda:0 over gloo, to
    # exercise the multi-rank control flow where RCCL cannot be used (it refuses two ranks on one device)
    backend = os.environ.get("FIBERS_BENCH_BACKEND", "nccl")
    if os.environ.get("FIBERS_BENCH_ONE_DEVICE"):
        local = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom
    L = fj.lib()
    nvox = SHAPE[0] * SHAPE[1] * SHAPE[2]
    sph = fj.sphere_642

    # ---- headline: GQI + peaks, 140^3 x 270 ------------------------------------------------------
    bval, bvec = phantom.scheme_gqi()
    dwi, axes = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=3 + rank, device=dev)
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    plan = fj.OdfPlan("gqi", bval, bvec, sph, sigma=1.25, device=dev.index)
    out = fj.odf_rec_device(plan, dwi, mask, normalize=False)

    def gqi_step():
        if world == 1:                                                     # one volume, one GPU: qa ./= odfmax inside the library call
            fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
            return
        fj.odf_rec_device(plan, dwi, mask, out=out, normalize=False)
        if world > 1:
            dist.all_reduce(out["odfmax"][:1], op=dist.ReduceOp.MAX)      # gqi.jl:164 across ranks
        fj.qa_normalize_device(out["qa"], out["odfmax"])                   # qa ./= the all-reduced odfmax, read on the device

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        gqi_step()
    sync()
    L.fib_profile_enable(1)
    L.fib_profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gqi_step()
    sync()
    dt = time.perf_counter() - t0
    L.fib_profile_enable(0)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    gemm_ms, gemm_n = prof_get(L, "odf_gemm")
    peaks_ms, peaks_n = prof_get(L, "odf_peaks")
    value = world * nvox * args.steps / dt / 1e6

    nvert, nvol = sph.nvert, len(bval)
    flops = 2.0 * nvert * nvol * nvox                      # algorithmic: 173 340 flop/voxel
    gemm_avg_ms = gemm_ms / max(gemm_n, 1)
    achieved = flops / (gemm_avg_ms * 1e-3) / 1e12 if gemm_n else 0.0
    gemm_bytes = (4.0 * nvol + 1 + 4.0 * nvert) * nvox     # read DWI + mask, write ODF
    split = os.environ.get("FIBERS_ODF_GEMM", "bf16x3").lower() != "f32"
    hbm2 = dict(achieved=gemm_bytes / (gemm_avg_ms * 1e-3) / 1e9 if gemm_n else 0.0, peak=PEAK_HBM_GBS, unit="GB/s",
                algorithmic_bytes=gemm_bytes, frac=gemm_bytes / (gemm_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if gemm_n else 0.0)
    pk = dict(avg_kernel_ms=peaks_ms / max(peaks_n, 1),
              hbm_gbs=(4.0 * nvert + 48) * nvox / (peaks_ms / max(peaks_n, 1) * 1e-3) / 1e9 if peaks_n else 0.0)
    if split:
        # every f32 product = 6 exact bf16 piece products -> the matrix cores execute 6 x the algorithmic flops (320 of the
        # 321 rows; K padded 270 -> 272); the binding roof is the BF16 MFMA peak / 6 for the algorithmic f32 flops
        peak_eff = PEAK_BF16_TFLOPS / 6.0
        roofline = dict(bound="mfma", kernel="odf_gemm3_kernel<MB=10,NX=1,NW=8> (v_mfma_f32_32x32x16_bf16 on exact 3-way bf16 splits "
                                             "of both f32 operands: 6 piece products per f32 product; 320 rows on MFMA + 1 row on VALU)",
                        achieved=achieved, peak=peak_eff, unit="TFLOP/s", frac=achieved / peak_eff,
                        note="achieved = algorithmic f32 flops (2*321*270 per voxel) / kernel time; peak = 2500 TFLOP/s dense BF16 / 6 piece "
                             "products; executed BF16 MFMA rate = %.0f TFLOP/s of 2500" % (6.0 * 2.0 * 320 * 272 * nvox / (gemm_avg_ms * 1e-3) / 1e12 if gemm_n else 0.0),
                        avg_kernel_ms=gemm_avg_ms, launches=gemm_n, traffic=None, hbm_secondary=hbm2, peaks_kernel=pk)
    else:
        roofline = dict(bound="mfma", kernel="odf_gemm_kernel<MB=10,NX=1> (v_mfma_f32_32x32x2_f32; 320 rows on MFMA + 1 row on VALU)", achieved=achieved,
                        peak=PEAK_F32_TFLOPS, unit="TFLOP/s", frac=achieved / PEAK_F32_TFLOPS,
                        avg_kernel_ms=gemm_avg_ms, launches=gemm_n, traffic=None, hbm_secondary=hbm2, peaks_kernel=pk)
    tr_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr_file):
        try:
            roofline["traffic"] = json.load(open(tr_file)).get("odf_gemm_bytes_per_launch")
        except Exception:
            pass

    extra = {}
    if not args.no_extra:
        # Every rank takes part (the ranks leave together).  C2: DTI fit, one 140^3 x 64 volume per rank (weak scaling, no
        # exchange step).  C4: streamlines from ONE volume's principal eigenvector: rank 0's field is broadcast over
        # RCCL/xGMI (the path's only bulk collective: 16 B/voxel), seeds shard round-robin, no data-path collective after.
        del out, dwi
        torch.cuda.empty_cache()
        b2, g2 = phantom.scheme_dti(60, 4, 1000.0, seed=2)
        d2, ax2 = phantom.make_dwi_torch(SHAPE, b2, g2, seed=2, device=dev, nfib=1)
        p2 = fj.DtiPlan(b2, g2, device=dev.index)
        o2 = fj.dti_fit_device(p2, d2, mask)
        sync()
        L.fib_profile_enable(1); L.fib_profile_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fj.dti_fit_device(p2, d2, mask, out=o2)
        sync()
        t_dti = time.perf_counter() - t0
        L.fib_profile_enable(0)
        if world > 1:
            tt = torch.tensor([t_dti], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_dti = float(tt.item())
        t_dti /= args.steps
        k_ms, k_n = prof_get(L, "dti_fit")
        dbytes = (4.0 * len(b2) + 1 + 64) * nvox
        extra["dti_fit_140x64"] = dict(mvoxels_per_s=world * nvox / t_dti / 1e6, ms_per_step=t_dti * 1e3,
                                       kernel_ms=k_ms / max(k_n, 1), algorithmic_bytes=dbytes,
                                       hbm_gbs=dbytes / (k_ms / max(k_n, 1) * 1e-3) / 1e9 if k_n else 0.0,
                                       hbm_frac=dbytes / (k_ms / max(k_n, 1) * 1e-3) / 1e9 / PEAK_HBM_GBS if k_n else 0.0,
                                       note="one volume per rank; per-kernel figures are rank 0's")
        # ---- streamlines from the DTI principal eigenvector, ball mask (C4) -----------------------
        from fibers_jl_amd import dist as fd
        bm = phantom.ball_mask_torch(SHAPE, dev)
        field, mout = fj.stream_field_device([o2["eigvec1"]], fa=o2["fa"], fa_thresh=0.1, mask=bm)
        seeds_all = torch.nonzero(mout).flatten()
        sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)

        def stream_step():
            if world > 1:
                dist.broadcast(field, src=0)                               # shared peak field over xGMI
            seeds, _ = fd.shard_seeds(seeds_all, world, rank)
            return fj.stream_device(field, SHAPE, seeds.contiguous(), sub)

        res = stream_step()
        sync()
        L.fib_profile_enable(1); L.fib_profile_reset()
        nst = max(2, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(nst):
            res = stream_step()
        sync()
        t_st = time.perf_counter() - t0
        L.fib_profile_enable(0)
        cnt = torch.tensor([float(res["xyz"].shape[0]), float(res["npts"].numel()), t_st], device=dev, dtype=torch.float64)
        if world > 1:
            tmax = cnt[2:].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(cnt[:2], op=dist.ReduceOp.SUM)
            cnt[2] = tmax[0]
        npoints, nlines, t_st = int(cnt[0].item()), int(cnt[1].item()), float(cnt[2].item()) / nst
        tr_ms, tr_n = prof_get(L, "stream_trace")
        pk_ms, pk_n = prof_get(L, "stream_pack")
        extra["stream_dti_ball"] = dict(seeds=int(seeds_all.numel()), lines=nlines, points=npoints,
                                        mpoints_per_s=npoints / t_st / 1e6, ms_per_step=t_st * 1e3,
                                        trace_kernel_ms=tr_ms / max(tr_n, 1), pack_kernel_ms=pk_ms / max(pk_n, 1),
                                        algorithmic_bytes=25.0 * npoints,
                                        hbm_gbs_trace=25.0 * (npoints / world) / (tr_ms / max(tr_n, 1) * 1e-3) / 1e9 if tr_n else 0.0,
                                        note="one volume, seeds sharded round-robin over the ranks"
                                             + (", field broadcast from rank 0 inside the timed step" if world > 1 else ""))
        if world == 1:
            # ---- microscopy regime (stream.jl:547-619) on the same field: every 8th seed, reference defaults ----------
            sm = seeds_all[::8].contiguous()
            z1 = torch.zeros((1, 3), dtype=torch.float32, device=dev)
            kw = dict(ang_thresh=20, step_size=1.0, smooth_coeff=0.0, search_dist=15, search_ang=10)
            rm = fj.stream_device(field, SHAPE, sm, z1, **kw)
            torch.cuda.synchronize()
            L.fib_profile_enable(1); L.fib_profile_reset()
            t0 = time.perf_counter()
            rm = fj.stream_device(field, SHAPE, sm, z1, **kw)
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
            ang = [torch.rand(n2 * n2, device=dev, generator=g) - 0.5 + k * 3.14159265 / 3 for k in range(3)]
            ov2 = [torch.stack([torch.cos(a_), torch.sin(a_), torch.zeros_like(a_)]) for a_ in ang]
            lc = torch.rand((10, n2 * n2), device=dev, generator=g)
            fld, mo = fj.stream_field_device(ov2, mask=torch.ones(n2 * n2, dtype=torch.uint8, device=dev))
            sd2 = torch.nonzero(mo).flatten()
            s2 = torch.tensor([[0.1, -0.2, 0.0]], dtype=torch.float32, device=dev)
            kw = dict(lcms=lc, lcm_thresh=0.099, strdims=(0, 1), rng_seed=7, len_max=140)
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
                                          note="2048x2048x1 pixels, 3 orientations + one 10-element LCM per pixel, len_max 140")
            del rl, fld, lc, ov2, ang
        if world == 1:
            # ---- RUMBA-SD (rusd.jl, row N4): 140^3 x 270 frames, ball mask, sphere_724 (364 compartments), 10 iterations ----
            torch.cuda.empty_cache()
            b4, g4 = phantom.scheme_gqi()
            d4, _ = phantom.make_dwi_torch(SHAPE, b4, g4, seed=3, device=dev)
            rp = fj.RumbaPlan(b4, g4, fj.sphere_724, device=dev.index)
            fj.rumba_rec_device(rp, d4, bm, SHAPE, niter=2)
            torch.cuda.synchronize()
            L.fib_profile_enable(1); L.fib_profile_reset()
            nit = 10
            t0 = time.perf_counter()
            rr = fj.rumba_rec_device(rp, d4, bm, SHAPE, niter=nit)
            torch.cuda.synchronize()
            t_r = time.perf_counter() - t0
            L.fib_profile_enable(0)
            gm_ms, gm_n = prof_get(L, "matrix_gemm")
            tv_ms, tv_n = prof_get(L, "rumba_tv")
            el_ms, el_n = prof_get(L, "rumba_elementwise")
            nmask = int(bm.sum())
            kk, _nd = rp.kernel().shape[1], rp.kernel().shape[0]
            extra["rumba_140_ball"] = dict(voxels=nmask, compartments=kk, dirs=_nd, iterations=nit, ms_total=t_r * 1e3,
                                           ms_per_iteration=(gm_ms + tv_ms + el_ms) / nit,
                                           gemm_ms_per_iteration=gm_ms / nit, tv_ms_per_iteration=tv_ms / nit,
                                           elementwise_ms_per_iteration=el_ms / nit,
                                           gemm_tflops=3 * 2.0 * kk * _nd * nmask * nit / (gm_ms * 1e-3) / 1e12 if gm_n else 0.0,
                                           snr_mean=rr["snr_mean"],
                                           note="three [364 x 253] contractions per iteration on the split-bf16 MFMA kernel; 600 iterations in the reference's default")
            del rr, d4, rp
        del res, bm, seeds_all

    if not args.no_extra and rank == 0 and world == 1:
        # ---- DSI 515-direction reconstruction + peaks (C5 fit part) ------------------------------------
        del field, o2, d2
        torch.cuda.empty_cache()
        b5, g5 = phantom.scheme_dsi()
        d5, _ = phantom.make_dwi_torch(SHAPE, b5, g5, seed=5, device=dev)
        p5 = fj.OdfPlan("dsi", b5, g5, sph, hann_width=32, device=dev.index)
        o5 = fj.odf_rec_device(p5, d5, mask)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        nd = max(2, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(nd):
            fj.odf_rec_device(p5, d5, mask, out=o5)
        torch.cuda.synchronize()
        t_dsi = (time.perf_counter() - t0) / nd
        L.fib_profile_enable(0)
        g_ms, g_n = prof_get(L, "odf_gemm")
        f_ms, f_n = prof_get(L, "dsi_fold")
        extra["dsi_rec_140x515"] = dict(mvoxels_per_s=nvox / t_dsi / 1e6, ms_per_step=t_dsi * 1e3,
                                        gemm_kernel_ms=g_ms / max(g_n, 1), fold_kernel_ms=f_ms / max(f_n, 1),
                                        note="antipodal folding inside the contraction kernel: 258 folded samples x (258 pdf + 321 odf) rows")
        # ---- C5 tracking: 3 peaks per voxel (f = qa, f_thresh = .03), ball mask, nsub = 10 -> ~10 M lines -------------
        del d5
        bm = phantom.ball_mask_torch(SHAPE, dev)
        field3, mout3 = fj.stream_field_device(o5["peak"], f=o5["qa"], f_thresh=0.03, mask=bm)
        seeds3 = torch.nonzero(mout3).flatten()
        sub10 = torch.from_numpy(fj.make_sublist(10, np.random.default_rng(5))).to(dev)
        r3 = fj.stream_device(field3, SHAPE, seeds3, sub10)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        t0 = time.perf_counter()
        for _ in range(2):
            r3 = fj.stream_device(field3, SHAPE, seeds3, sub10)
        torch.cuda.synchronize()
        t3 = (time.perf_counter() - t0) / 2
        L.fib_profile_enable(0)
        tr_ms, tr_n = prof_get(L, "stream_trace")
        pk_ms, pk_n = prof_get(L, "stream_pack")
        np3 = int(r3["xyz"].shape[0])
        extra["stream_dsi_3peaks_10M"] = dict(seeds=int(seeds3.numel()), nsub=10, lines=int(r3["npts"].numel()), points=np3,
                                              mpoints_per_s=np3 / t3 / 1e6, ms_per_step=t3 * 1e3,
                                              trace_kernel_ms=tr_ms / max(tr_n, 1), pack_kernel_ms=pk_ms / max(pk_n, 1),
                                              algorithmic_bytes=49.0 * np3)
        del o5, r3, field3

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_gqi(bval, bvec, sph, seed=3)

    if rank == 0:
        line = dict(metric="Mvoxels/s fit (GQI ODF + peaks, 140^3 x 270-dir); Mpoints/s streamline in extra",
                    value=value, unit="Mvoxels/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                    dtype="f32" if not split else "f32 (exact 3xbf16 operand splits on the bf16 matrix cores, f32 accumulate)", data="synthetic",
                    config=dict(workload="gqi_rec + find_peaks + qa normalisation, 140x140x140 x 270 frames "
                                         "(18 x b=5 + 84 dirs x {1000,2000,3000}), sphere_642, mask = all ones, "
                                         "one volume per GPU", voxels_per_gpu=nvox, frames=nvol, odf_vertices=nvert,
                                parallelism="volumes sharded over ranks, 1-float all-reduce(MAX)" if world > 1 else "single GPU"),
                    roofline=roofline, cpu_baseline=cpu, extra=extra)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
