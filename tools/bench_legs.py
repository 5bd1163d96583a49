"""The legs of bench.py beyond its headline step: every other BASELINE config (C2 DTI, C4 tracking, C5 DSI + 3-peak tracking), the
less flattering inputs of the headline, the host-tier (PCIe-inclusive) calls, the file-to-file pipeline, the adjacent tracking
modes and RUMBA-SD.  Each function returns a dict of NUMBERS (what a key means is documented here and in DESIGN.md §3, not in the
result); bench.py writes all of them to bench_extra.json and keeps a whitelisted few per leg in its one result line.

Nothing here imports the oracle: the CPU baselines live in bench.py's cpu_baseline leg."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec
ACHIEVABLE_HBM_GBS = 6300.0    # MI355X_MICROARCH.md: what a copy kernel reaches (6.29 TB/s measured)
PEAK_BF16_TFLOPS = 2500.0


def stored_traffic():
    """profiles/traffic.json: HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes,
    tools/collect_profiles.sh) -- STORED figures of the last collection, not measured in this run"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:                                                        # noqa: BLE001
        return {}


def _kernel_traffic(tj, *prefixes):
    tot = 0.0
    for name, rec in (tj.get("kernels") or {}).items():
        if any(name.startswith(p) for p in prefixes):
            tot += rec.get("hbm_bytes_per_launch", 0.0)
    return tot or None


def host_tier(ctx):
    """the boundary a Julia caller pays for (SURVEY 8d "report both"): the fib_* entry points on pageable host arrays, PCIe both ways.
    Run FIRST among the legs: a process that has just released tens of GB of device memory sees slower downloads for a few seconds
    (tools/host_tier_state_check.py).  Per leg: median and best of the calls, pcie floor = max(bytes in, out) / 63 GB/s."""
    import host_tier_probe as htp
    from fibers_jl_amd import phantom
    ht = {}
    legs = [("gqi_rec", lambda: htp.leg_odf("gqi", shape=ctx.shape, reps=5, dev=ctx.dev))]

    def ball():
        mb = np.ascontiguousarray(phantom.ball_mask_torch(ctx.shape, ctx.dev).reshape(-1).cpu().numpy().astype(np.uint8))
        r = htp.leg_odf("gqi", shape=ctx.shape, reps=5, mask=mb, dev=ctx.dev)
        r["voxels_in_mask"] = int(mb.sum())
        return r
    legs += [("gqi_rec_ball_mask", ball), ("dti_fit", lambda: htp.leg_dti(shape=ctx.shape, reps=5, dev=ctx.dev)),
             ("dsi_rec", lambda: htp.leg_odf("dsi", shape=ctx.shape, reps=3, dev=ctx.dev)),
             ("stream_c4", lambda: htp.leg_stream(shape=ctx.shape, reps=3, dev=ctx.dev))]
    for name, fn in legs:
        try:
            r = fn()
            r.pop("note", None)
            r["pcie_roof_frac_median"] = r["pcie_floor_ms"] / r["e2e_pcie_ms_median"]
            ht[name] = r
        except Exception as e:                                                      # noqa: BLE001
            ht[name] = dict(error=str(e))
    return ht


def in_kernel_clock(ctx):
    """in-kernel clock of the contraction kernels (MI355X_MICROARCH.md "DVFS give-back" item 6): a CHILD process loads the diagnostic
    build (libfibers_hip_stamp.so: one s_memtime / s_memrealtime pair around each workgroup's work loop) and runs the GQI and DSI
    steps for 2 s each; the product library never executes a stamp"""
    if not os.path.exists(os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so")) or ctx.shape != (140, 140, 140):
        return dict(skipped=1)                                                   # (no diagnostic build / reduced shape)
    o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_clock.py"), "--seconds", "2.0", "--kernels", "fused,dsi"],
                       capture_output=True, text=True, timeout=300)
    kc = json.loads([ln for ln in o.stdout.splitlines() if ln.startswith("{")][-1])
    kc.pop("note", None)
    return kc


def power(ctx, step, gemm_avg_ms, nloc):
    """roofline.power: the step's Joules LIVE from the board's energy counter around ~2 s of back-to-back steps; the Joules per byte /
    flop / instruction of its ingredients STORED (tools/energy_model.py -> profiles/energy_model.json).  `frac` is the calibrated
    figure (components scaled so that components + idle = the measured Joules), `frac_raw` the uncalibrated one."""
    import torch
    from fibers_jl_amd import energy as en
    rs = en.rsmi_index_of(ctx.dev.index)
    em = en.measure(step, torch.cuda.synchronize, seconds=2.0, dev=rs)
    if em is None:
        return dict(error="no board energy counter (librocm_smi64 / rsmi_dev_energy_count_get)")
    torch.cuda.synchronize()
    idle_w = en.idle_watts(1.2, dev=rs)
    stored = json.load(open(os.path.join(ROOT, "profiles", "energy_model.json")))
    gm = stored.get("gqi_model") or {}
    pw = en.gqi_power_roofline(gm["joules_per_unit"], nloc, gemm_avg_ms, em["ms_per_step"], em["joules_per_step"], idle_w)
    pw.update(board_watts_while_stepping=em["watts"], smu_sclk_mhz_while_stepping=em["sclk_mhz_mean"], steps_measured=em["steps"], rsmi_index=rs)
    return pw


def gqi_variants(ctx, plan, dwi, out, step_ms):
    """the headline step on other inputs: (a) ball mask (36 % inside) with ~1 % of the samples non-positive (SURVEY 8d variant B),
    (b) rank 0 of 8's z-slab alone on this GPU, the step of odf_rec_sharded without its collective (bounds the strong-scaling
    efficiency from the fixed per-step cost: ideal = the N=1 step x slab share), (c) the other operand format"""
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import dist as fd, phantom
    args, dev, shape = ctx.args, ctx.dev, ctx.shape
    nx, ny, nz = shape
    nxy, nvox = nx * ny, nx * ny * nz
    res = {}
    bm_h = phantom.ball_mask_torch(shape, dev)
    g = torch.Generator(device=dev); g.manual_seed(17)
    dwi_np = dwi.clone()
    hit = torch.rand(dwi_np.shape, generator=g, device=dev) < 0.01
    dwi_np[hit] = torch.where(torch.rand(int(hit.sum()), generator=g, device=dev) < 0.5, 0.0, -3.0)
    del hit
    out_b = fj.odf_rec_device(plan, dwi_np, bm_h, normalize=True)
    nb = max(2, args.steps // 2)
    t_b = ctx.timed(lambda: fj.odf_rec_device(plan, dwi_np, bm_h, out=out_b, normalize=True), nb, 1) / nb
    gb_ms, gb_n = ctx.prof_get("odf_gemm")
    nin = int(bm_h.sum())
    res["gqi_ball_mask_nonpositive"] = dict(voxels_in_mask=nin, ms_per_step=t_b * 1e3, mvoxels_in_mask_per_s=nin / t_b / 1e6,
                                            mvoxels_of_volume_per_s=nvox / t_b / 1e6, gemm_kernel_ms=gb_ms / max(gb_n, 1))
    del dwi_np, out_b, bm_h
    try:
        zs0, zs1 = fd.slab_bounds(nz, 8, 0, nxy)
        ns = (zs1 - zs0) * nxy
        counts8 = [(b - a) * nxy for a, b in (fd.slab_bounds(nz, 8, r, nxy) for r in range(8))]
        dwi_s = dwi[:, :ns].contiguous()
        mask_s = torch.ones(ns, dtype=torch.uint8, device=dev)
        out_s = fj.odf_rec_device(plan, dwi_s, mask_s, normalize=False)
        nss = max(4, args.steps)
        t_s = ctx.timed(lambda: fd.odf_rec_sharded(plan, dwi_s, mask_s, out=out_s, counts=counts8), nss, 2) / nss
        gs_ms, gs_n = ctx.prof_get("odf_gemm")
        res["gqi_slab_1of8"] = dict(voxels=ns, nz=zs1 - zs0, ms_per_step=t_s * 1e3, gemm_kernel_ms=gs_ms / max(gs_n, 1),
                                    ideal_ms=step_ms * ns / nvox, efficiency_before_collectives=(step_ms * 1e-3 * ns / nvox) / t_s)
        del dwi_s, mask_s, out_s
    except Exception as e:                                                      # noqa: BLE001
        res["gqi_slab_1of8"] = dict(error=str(e))
    try:
        exact = plan.format == "bf16x3"
        bval, bvec = phantom.scheme_gqi()
        plan_x = fj.OdfPlan("gqi", bval, bvec, ctx.sph, sigma=1.25, device=dev.index, format="fp16x2" if exact else "bf16x3")
        mask = torch.ones(dwi.shape[1], dtype=torch.uint8, device=dev)
        out_x = fj.odf_rec_device(plan_x, dwi, mask, normalize=True)
        t_x = ctx.timed(lambda: fj.odf_rec_device(plan_x, dwi, mask, out=out_x, normalize=True), args.steps, 1) / args.steps
        gx_ms, gx_n = ctx.prof_get("odf_gemm")
        den = out["odf"].abs().amax(dim=0).clamp_min(1e-30)
        res["gqi_other_format"] = dict(format=plan_x.format, ms_per_step=t_x * 1e3, mvoxels_per_s=nvox / t_x / 1e6, gemm_kernel_ms=gx_ms / max(gx_n, 1),
                                       odf_max_difference_of_voxel_max=float(((out_x["odf"] - out["odf"]).abs().amax(dim=0) / den).max()),
                                       first_peak_identical_fraction=float((out_x["peak"][0] == out["peak"][0]).all(dim=0).float().mean()))
        del out_x
        plan_x.close()
    except Exception as e:                                                      # noqa: BLE001
        res["gqi_other_format"] = dict(error=str(e))
    return res


def gqi_weak(ctx, plan, bval, bvec):
    """weak-scaling figure of the headline step: one whole volume per rank, odfmax all-reduced"""
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import dist as fd, phantom
    nvox = ctx.nvox
    dwi_w, _ = phantom.make_dwi_torch(ctx.shape, bval, bvec, seed=3 + ctx.rank, device=ctx.dev)
    mask_w = torch.ones(nvox, dtype=torch.uint8, device=ctx.dev)
    out_w = fj.odf_rec_device(plan, dwi_w, mask_w, normalize=False)
    nst = max(2, ctx.args.steps // 2)
    t_w = ctx.timed(lambda: fd.odf_rec_sharded(plan, dwi_w, mask_w, out=out_w, counts=[nvox] * ctx.world, always=ctx.force_pg), nst, 1)
    return dict(mvoxels_per_s=ctx.world * nvox * nst / t_w / 1e6, ms_per_step=t_w / nst * 1e3)


def dti_and_c4(ctx):
    """C2: DTI fit 140^3 x 64 in z-slabs (no exchange step).  C4: streamlines from its principal eigenvector, ball mask: the slab's
    field is all-gathered over RCCL inside the timed step (16 B/voxel), seeds round-robin, no collective after.  At N = 1 also the
    one-call / enqueue forms, the trilinear option, the divergent-termination phantom, the microscopy and LCM modes and RUMBA-SD."""
    import torch
    import torch.distributed as dist
    import fibers_jl_amd as fj
    from fibers_jl_amd import dist as fd, phantom
    args, dev, shape, L, world, multi, rank = ctx.args, ctx.dev, ctx.shape, ctx.L, ctx.world, ctx.multi, ctx.rank
    counts, v0, v1, nloc, nvox = ctx.counts, ctx.v0, ctx.v1, ctx.nloc, ctx.nvox
    timed, prof_get = ctx.timed, ctx.prof_get
    tj = stored_traffic()
    res = {}
    b2, g2 = phantom.scheme_dti(60, 4, 1000.0, seed=2)
    d2f, _ = phantom.make_dwi_torch(shape, b2, g2, seed=2, device=dev, nfib=1)
    d2 = d2f[:, v0:v1].contiguous() if world > 1 else d2f
    del d2f
    mask = torch.ones(nloc, dtype=torch.uint8, device=dev)
    p2 = fj.DtiPlan(b2, g2, device=dev.index)
    o2 = fj.dti_fit_device(p2, d2, mask)
    t_dti = timed(lambda: fj.dti_fit_device(p2, d2, mask, out=o2), args.steps, 1) / args.steps
    k_ms, k_n = prof_get("dti_fit")
    dbytes = (4.0 * len(b2) + 1 + 64) * nloc
    k_avg = k_ms / max(k_n, 1)
    res["dti_fit_140x64"] = dict(mvoxels_per_s=nvox / t_dti / 1e6, ms_per_step=t_dti * 1e3, kernel_ms=k_avg, algorithmic_bytes=dbytes,
                                 hbm_gbs=dbytes / (k_avg * 1e-3) / 1e9 if k_n else 0.0, hbm_frac=dbytes / (k_avg * 1e-3) / 1e9 / PEAK_HBM_GBS if k_n else 0.0)
    bm_full = phantom.ball_mask_torch(shape, dev)
    field_loc, mout_loc = fj.stream_field_device([o2["eigvec1"]], fa=o2["fa"], fa_thresh=0.1, mask=bm_full[v0:v1].contiguous())
    mout = fd.allgather_slabs(mout_loc, counts, always=ctx.force_pg)
    seeds_all = torch.nonzero(mout).flatten()
    sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
    xyz_buf = {}

    def xyz_out(npnt):                                                   # steady-state output buffer (no per-call allocation)
        if xyz_buf.get("t") is None or xyz_buf["t"].numel() < 3 * npnt:
            xyz_buf["t"] = torch.empty(int(3 * npnt * 1.05) + 16, dtype=torch.float32, device=dev)
        return xyz_buf["t"]
    r_ = {}
    sbuf4 = fj.StreamBuffers(dev) if not multi else None

    def stream_step():
        field = fd.allgather_slabs(field_loc, counts, always=ctx.force_pg)   # shared peak field over xGMI
        if multi:
            r_["r"] = fd.stream_sharded(field, shape, seeds_all, sub, xyz_out=xyz_out)
        else:                                                            # one GPU: the one-call form into kept buffers
            r_["r"] = fj.stream_device_run(field, shape, seeds_all, sub, buffers=sbuf4)
    nst = max(2, args.steps // 2)
    t_st = timed(stream_step, nst, 2)
    r = r_["r"]
    cnt = torch.tensor([float(r["xyz"].shape[0]), float(r["npts"].numel())], device=dev, dtype=torch.float64)
    if multi:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    npoints, nlines, t_st = int(cnt[0].item()), int(cnt[1].item()), t_st / nst
    tr_ms, tr_n = prof_get("stream_trace")
    pk_ms, pk_n = prof_get("stream_pack")
    sc_ms, _ = prof_get("stream_scan")
    ksum = (tr_ms + pk_ms + sc_ms) / max(tr_n, 1)
    alg = 25.0 * (npoints / world)                                       # 25 B per emitted point (nvec = 1), rank 0's share
    pmc = _kernel_traffic(tj, "stream_trace_kernel<1", "stream_pack_tile_kernel", "scan_block_kernel") if shape == (140, 140, 140) else None
    res["stream_dti_ball"] = dict(seeds=int(seeds_all.numel()), lines=nlines, points=npoints, mpoints_per_s=npoints / t_st / 1e6, ms_per_step=t_st * 1e3,
                                  trace_kernel_ms=tr_ms / max(tr_n, 1), pack_kernel_ms=pk_ms / max(pk_n, 1), kernel_sum_ms=ksum, algorithmic_bytes=25.0 * npoints,
                                  frac=alg / (ksum * 1e-3) / 1e9 / PEAK_HBM_GBS if tr_n else 0.0,
                                  pmc_bytes=pmc, traffic_frac=(pmc / (ksum * 1e-3) / 1e9 / ACHIEVABLE_HBM_GBS) if (pmc and tr_n and world == 1) else None)
    if rank == 0 and world == 1:
        try:
            f_once = fd.allgather_slabs(field_loc, counts)
            cnt2 = torch.zeros(2, dtype=torch.int64, device=dev)
            t_enq = timed(lambda: fj.stream_device_run_enqueue(f_once, shape, seeds_all, sub, sbuf4, counts=cnt2), nst, 1) / nst
            torch.cuda.synchronize()
            ke_ms, ke_n = prof_get("stream_trace")
            kp_ms, _ = prof_get("stream_pack")
            ks_ms, _ = prof_get("stream_scan")
            res["stream_dti_ball"]["enqueue_form"] = dict(ms_per_step=t_enq * 1e3, mpoints_per_s=int(cnt2[1]) / t_enq / 1e6, lines=int(cnt2[0]), points=int(cnt2[1]),
                                                          kernel_sum_ms=(ke_ms + kp_ms + ks_ms) / max(ke_n, 1))
            del f_once
        except Exception as e:                                                  # noqa: BLE001
            res["stream_dti_ball"]["enqueue_form"] = dict(error=str(e))
        # the trilinear option (fib_stream_params.interp = 1; not in the reference) on the same field and seeds
        field_all = fd.allgather_slabs(field_loc, counts)
        rt = {}

        def tri_step():
            rt["r"] = fj.stream_device(field_all, shape, seeds_all, sub, xyz_out=xyz_out, interp="trilinear")
        t_tri = timed(tri_step, nst, 1) / nst
        tt_ms, tt_n = prof_get("stream_trace")
        npt = int(rt["r"]["xyz"].shape[0])
        res["stream_dti_ball_trilinear"] = dict(lines=int(rt["r"]["npts"].numel()), points=npt, mpoints_per_s=npt / t_tri / 1e6,
                                                ms_per_step=t_tri * 1e3, trace_kernel_ms=tt_ms / max(tt_n, 1))
        del rt, field_all
        # divergent termination: the bundle phantom (lines end where bundles meet at > 45 degrees); static_lane_idle_frac = the share of
        # lane-steps that idle because a wave runs as long as its longest line
        ovb, mb = phantom.bundle_field_torch(shape, dev)
        fb, mob = fj.stream_field_device([ovb], mask=mb)
        sb = torch.nonzero(mob).flatten()
        div = {}
        for nsub_b in (1, 10):
            subb = torch.from_numpy(fj.make_sublist(nsub_b, np.random.default_rng(5))).to(dev) if nsub_b > 1 else sub
            rb = {}

            def bstep():
                rb["r"] = fj.stream_device(fb, shape, sb, subb, want_all_npts=True, xyz_out=xyz_out)
            t_b = timed(bstep, 3, 1) / 3
            tb_ms, tb_n = prof_get("stream_trace")
            nall = rb["r"]["all_npts"].cpu().numpy().astype(np.int64)
            it = nall + 2                                            # loop trips of a lane: its points + the two failed steps
            w = np.concatenate([it, np.zeros((-len(it)) % 64, np.int64)]).reshape(-1, 64)
            div["nsub%d" % nsub_b] = dict(lines=int(len(nall)), points=int(rb["r"]["xyz"].shape[0]), npts_median=float(np.median(nall)), npts_max=int(nall.max()),
                                          static_lane_idle_frac=float(1.0 - it.sum() / (w.max(1).sum() * 64.0)),
                                          trace_kernel_ms=tb_ms / max(tb_n, 1), ms_per_step=t_b * 1e3, mpoints_per_s=int(rb["r"]["xyz"].shape[0]) / t_b / 1e6)
            del rb
        res["stream_bundle_divergent"] = div
        del ovb, mb, fb, mob, sb
        # microscopy regime (stream.jl:547-619) on the same field: every 8th seed, reference defaults of the regime
        sm = seeds_all[::8].contiguous()
        z1_ = torch.zeros((1, 3), dtype=torch.float32, device=dev)
        kw = dict(ang_thresh=20, step_size=1.0, smooth_coeff=0.0, search_dist=15, search_ang=10, xyz_out=xyz_out)
        fj.stream_device(field_loc, shape, sm, z1_, **kw)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        t0 = time.perf_counter()
        rm = fj.stream_device(field_loc, shape, sm, z1_, **kw)
        torch.cuda.synchronize()
        t_m = time.perf_counter() - t0
        L.fib_profile_enable(0)
        mk_ms, mk_n = prof_get("stream_trace_micro")
        npm = int(rm["xyz"].shape[0])
        res["stream_micro_ball"] = dict(seeds=int(sm.numel()), lines=int(rm["npts"].numel()), points=npm, mpoints_per_s=npm / t_m / 1e6, ms_per_step=t_m * 1e3,
                                        trace_kernel_ms=mk_ms / max(mk_n, 1))
        del rm
        # LCM-guided tracking (stream.jl:380-495) on a synthetic 2-D section: 2048^2 pixels, 3 orientation ANGLES each
        n2 = 2048
        g = torch.Generator(device=dev); g.manual_seed(11)
        ang = [((torch.rand(n2 * n2, device=dev, generator=g) - 0.5 + k * 3.14159265 / 3 + 1.5707963) % 3.14159265) - 1.5707963 for k in range(3)]
        ov2 = [fj.angles_to_vectors_device(a_.clamp(-1.5707963, 1.5707963), volres=(0.5, 0.5, 2.0))[0] for a_ in ang]
        lc = torch.rand((10, n2 * n2), device=dev, generator=g)
        fld, mo = fj.stream_field_device(ov2, mask=torch.ones(n2 * n2, dtype=torch.uint8, device=dev))
        sd2 = torch.nonzero(mo).flatten()
        s2 = torch.tensor([[0.1, -0.2, 0.0]], dtype=torch.float32, device=dev)
        kw = dict(lcms=lc, lcm_thresh=0.099, strdims=(0, 1), rng_seed=7, len_max=140, xyz_out=xyz_out)
        fj.stream_device(fld, (n2, n2, 1), sd2, s2, **kw)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        t0 = time.perf_counter()
        rl = fj.stream_device(fld, (n2, n2, 1), sd2, s2, **kw)
        torch.cuda.synchronize()
        t_l = time.perf_counter() - t0
        L.fib_profile_enable(0)
        lk_ms, lk_n = prof_get("stream_trace_lcm")
        npl = int(rl["xyz"].shape[0])
        res["stream_lcm_2d"] = dict(seeds=int(sd2.numel()), lines=int(rl["npts"].numel()), points=npl, mpoints_per_s=npl / t_l / 1e6, ms_per_step=t_l * 1e3,
                                    trace_kernel_ms=lk_ms / max(lk_n, 1), flagged_fraction=float(rl["flags"].float().mean()))
        del rl, fld, lc, ov2, ang
        # RUMBA-SD (rusd.jl, row N4): 140^3 x 270 frames, ball mask, sphere_724 (364 compartments), 10 iterations
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
        gm_ms, gm_n = prof_get("matrix_gemm")
        tv_ms, _ = prof_get("rumba_tv")
        el_ms, _ = prof_get("rumba_elementwise")
        nmask = int(bm_full.sum())
        kk, _nd = rp.kernel().shape[1], rp.kernel().shape[0]
        res["rumba_140_ball"] = dict(voxels=nmask, compartments=kk, dirs=_nd, iterations=nit, ms_total=t_r * 1e3, ms_per_iteration=(gm_ms + tv_ms + el_ms) / nit,
                                     gemm_ms_per_iteration=gm_ms / nit, tv_ms_per_iteration=tv_ms / nit, elementwise_ms_per_iteration=el_ms / nit,
                                     gemm_tflops=3 * 2.0 * kk * _nd * nmask * nit / (gm_ms * 1e-3) / 1e12 if gm_n else 0.0, snr_mean=rr["snr_mean"])
        del rr, d4, rp
    return res


def c5(ctx):
    """C5 (BASELINE config 5) at every N: DSI 515-direction reconstruction in z-slabs (dsi.jl:197) with the global odfmax all-reduced
    (dsi.jl:263); then the 3-peak field + mask all-gathered over RCCL inside the timed step and ~10 M seeds x offsets round-robin over
    the ranks (stream.jl:757-761)"""
    import torch
    import torch.distributed as dist
    import fibers_jl_amd as fj
    from fibers_jl_amd import dist as fd, phantom
    args, dev, shape, world, multi = ctx.args, ctx.dev, ctx.shape, ctx.world, ctx.multi
    counts, v0, v1, nloc, nvox = ctx.counts, ctx.v0, ctx.v1, ctx.nloc, ctx.nvox
    timed, prof_get = ctx.timed, ctx.prof_get
    tj = stored_traffic()
    res = {}
    nvert = ctx.sph.nvert
    b5, g5 = phantom.scheme_dsi()
    d5f, _ = phantom.make_dwi_torch(shape, b5, g5, seed=5, device=dev)
    d5 = d5f[:, v0:v1].contiguous() if world > 1 else d5f
    del d5f
    torch.cuda.empty_cache()
    mask = torch.ones(nloc, dtype=torch.uint8, device=dev)
    p5 = fj.OdfPlan("dsi", b5, g5, ctx.sph, hann_width=32, device=dev.index)
    o5 = fj.odf_rec_device(p5, d5, mask, normalize=False)
    nd = max(2, args.steps // 2)

    def dsi_step():
        if not multi:
            fj.odf_rec_device(p5, d5, mask, out=o5, normalize=True)
        else:
            fd.odf_rec_sharded(p5, d5, mask, out=o5, counts=counts, always=ctx.force_pg)
    t_dsi = timed(dsi_step, nd, 1) / nd
    g_ms, g_n = prof_get("odf_gemm")
    q_ms, q_n = prof_get("odf_post")
    n5 = len(b5)
    dsi_bytes = (4.0 * n5 + 1 + 4.0 * n5 + 4.0 * nvert + 48) * nloc          # 5 456 B / voxel (DWI + mask in; pdf, odf, peaks, qa out)
    dsi_k_ms = g_ms / max(g_n, 1)
    nprod5 = 6 if p5.format == "bf16x3" else 3
    dsi_exec = nprod5 * 2.0 * (320 + 288) * 272 * nloc                       # executed MFMA flops: piece products x (10 + 9 blocks) x 32 rows x 17 stages x 16
    res["dsi_rec_140x515"] = dict(mvoxels_per_s=nvox / t_dsi / 1e6, ms_per_step=t_dsi * 1e3, gemm_kernel_ms=dsi_k_ms, post_kernel_ms=q_ms / max(q_n, 1),
                                  algorithmic_bytes=dsi_bytes, frac=dsi_bytes / (dsi_k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if g_n else 0.0,
                                  mfma_frac=dsi_exec / (dsi_k_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if g_n else 0.0)
    del d5
    bm_full5 = phantom.ball_mask_torch(shape, dev)
    f3_loc, m3_loc = fj.stream_field_device(o5["peak"], f=o5["qa"], f_thresh=0.03, mask=bm_full5[v0:v1].contiguous())
    mout3 = fd.allgather_slabs(m3_loc, counts, always=ctx.force_pg)
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
        field3 = fd.allgather_slabs(f3_loc, counts, always=ctx.force_pg)       # the shared 3-peak field over xGMI (48 B / voxel)
        if multi:
            r3["r"] = fd.stream_sharded(field3, shape, seeds3, sub10, xyz_out=xyz_out5)
        else:                                                                  # one GPU: fibd_stream_run (from 2^21 lines: the fused trace + look-back + pack kernel)
            r3["r"] = fj.stream_device_run(field3, shape, seeds3, sub10, buffers=sbuf5)
    t3 = timed(c5_step, 3, 2) / 3
    tr_ms, tr_n = prof_get("stream_trace")
    pk_ms, pk_n = prof_get("stream_pack")
    sc_ms, _ = prof_get("stream_scan")
    cnt3 = torch.tensor([float(r3["r"]["xyz"].shape[0]), float(r3["r"]["npts"].numel())], device=dev, dtype=torch.float64)
    if multi:
        dist.all_reduce(cnt3, op=dist.ReduceOp.SUM)
    np3, nl3 = int(cnt3[0].item()), int(cnt3[1].item())
    ksum3 = (tr_ms + pk_ms + sc_ms) / max(tr_n, 1)
    pmc3 = (tj.get("stream_c5_bytes_per_step") if shape == (140, 140, 140) else None)
    res["stream_dsi_3peaks_10M"] = dict(seeds=int(seeds3.numel()), nsub=10, lines=nl3, points=np3, mpoints_per_s=np3 / t3 / 1e6, ms_per_step=t3 * 1e3,
                                        trace_kernel_ms=tr_ms / max(tr_n, 1), pack_kernel_ms=pk_ms / max(pk_n, 1), kernel_sum_ms=ksum3, algorithmic_bytes=49.0 * np3,
                                        frac=49.0 * (np3 / world) / (ksum3 * 1e-3) / 1e9 / PEAK_HBM_GBS if tr_n else 0.0,
                                        pmc_bytes=pmc3, traffic_frac=(pmc3 / (ksum3 * 1e-3) / 1e9 / ACHIEVABLE_HBM_GBS) if (pmc3 and tr_n and world == 1) else None)
    if not multi:                                                          # the same without the host round trip at the end of every call
        try:
            field3e = fd.allgather_slabs(f3_loc, counts)
            cnt5 = torch.zeros(2, dtype=torch.int64, device=dev)
            t5e = timed(lambda: fj.stream_device_run_enqueue(field3e, shape, seeds3, sub10, sbuf5, counts=cnt5), 3, 1) / 3
            torch.cuda.synchronize()
            res["stream_dsi_3peaks_10M"]["enqueue_form"] = dict(ms_per_step=t5e * 1e3, mpoints_per_s=int(cnt5[1]) / t5e / 1e6, lines=int(cnt5[0]), points=int(cnt5[1]))
            del field3e
        except Exception as e:                                                  # noqa: BLE001
            res["stream_dsi_3peaks_10M"]["enqueue_form"] = dict(error=str(e))
    return res


def pipeline(ctx):
    """the drop-in path file to file on the tutorial's shape (tools/pipeline.py): .nii (mmap) -> dti_fit + gqi_rec -> tracking on eigvec1 / FA
    and on GQI peaks / QA -> .trk through the GPU serialiser, wall time per stage; host_path_*: the same with a volume read into memory,
    fib_stream into host arrays and the NumPy trk_write.  The two ways' .trk files are compared byte for byte."""
    import pipeline as pl
    shape = pl.TUTORIAL_SHAPE if ctx.shape == (140, 140, 140) else (ctx.shape[0], ctx.shape[1], max(4, ctx.shape[2] * 92 // 140))
    r = pl.measure(shape, ctx.dev)
    if not r["trk_files_identical"]:
        raise RuntimeError("the device and host paths wrote different .trk files")
    return r
