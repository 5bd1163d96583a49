#!/usr/bin/env python3
"""Energy model of the contraction kernels on one MI355X: is "the 1 400 W cap binds" a measured bound?

1. runs tools/probes/energy_probe (Joules per byte / flop / instruction of each ingredient, from the board's energy counter);
2. measures the product's steps the same way (GQI default + exact split, DSI, DTI, tracking: Joules per step, board power);
3. runs tools/kernel_clock.py (diagnostic build: in-kernel clock from s_memtime / s_memrealtime, with the SMU's reported shader
   clock and the board power sampled in the same seconds);
4. composes per step: joules_by_component = count_i x (J per unit)_i, the floor a kernel made of nothing but the unavoidable
   ingredients would reach under the cap, and the fraction of it the kernel reaches.
Writes one JSON document (default profiles/r05/energy_model.json when run through gpurun: gpurun_out/r05/energy_model.json).

usage: python tools/energy_model.py [--seconds 3] [--out gpurun_out/r05/energy_model.json] [--skip-probe]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPE = (140, 140, 140)
NVOX = 140 ** 3


def run_probe(seconds):
    exe = os.path.join(ROOT, "tools", "probes", "energy_probe")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-I/opt/rocm/include", exe + ".hip", "-o", exe,
                               "-L/opt/rocm/lib", "-lrocm_smi64"])
    o = subprocess.run([exe, str(seconds)], capture_output=True, text=True, timeout=1800)
    rows = [json.loads(ln) for ln in o.stdout.splitlines() if ln.startswith("{")]
    if o.returncode != 0:
        print(o.stderr[-2000:], file=sys.stderr)
    return {r["mode"]: r for r in rows}


def product_steps(seconds):
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import energy as en, phantom
    dev = torch.device("cuda", 0)
    sync = torch.cuda.synchronize
    L = fj.lib()
    res = dict(idle_watts=en.idle_watts(2.0))
    mask = torch.ones(NVOX, dtype=torch.uint8, device=dev)

    def timed(label, step, kname):
        L.fib_profile_enable(1)
        L.fib_profile_reset()
        r = en.measure(step, sync, seconds=seconds)
        import ctypes as C
        ms, n = C.c_double(0), C.c_int64(0)
        L.fib_profile_get(kname.encode(), C.byref(ms), C.byref(n))
        L.fib_profile_enable(0)
        if r is not None:
            r["kernel_ms"] = ms.value / max(n.value, 1)
        res[label] = r
        print(label, r, flush=True)

    bval, bvec = phantom.scheme_gqi()
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=3, device=dev)
    for fmt in ("fp16x2", "bf16x3"):
        plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0, format=fmt)
        out = fj.odf_rec_device(plan, dwi, mask, normalize=True)
        timed("gqi_" + fmt, lambda: fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True), "odf_gemm")
        del out
        plan.close()
    del dwi
    torch.cuda.empty_cache()
    b5, g5 = phantom.scheme_dsi()
    d5, _ = phantom.make_dwi_torch(SHAPE, b5, g5, seed=5, device=dev)
    p5 = fj.OdfPlan("dsi", b5, g5, fj.sphere_642, hann_width=32, device=0)
    o5 = fj.odf_rec_device(p5, d5, mask, normalize=True)
    timed("dsi", lambda: fj.odf_rec_device(p5, d5, mask, out=o5, normalize=True), "odf_gemm")
    del d5, o5
    torch.cuda.empty_cache()
    b2, g2 = phantom.scheme_dti(60, 4, 1000.0, 2)
    d2, _ = phantom.make_dwi_torch(SHAPE, b2, g2, 2, dev, nfib=1)
    p2 = fj.DtiPlan(b2, g2)
    o2 = fj.dti_fit_device(p2, d2, mask)
    timed("dti", lambda: fj.dti_fit_device(p2, d2, mask, out=o2), "dti_fit")
    bm = phantom.ball_mask_torch(SHAPE, dev)
    field, mout = fj.stream_field_device([o2["eigvec1"]], fa=o2["fa"], fa_thresh=0.1, mask=bm)
    seeds = torch.nonzero(mout).flatten()
    sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
    bufs = fj.StreamBuffers(dev)
    fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
    timed("stream_c4", lambda: fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs), "stream_trace")
    return res


def compose(probe, steps, clock):
    """roofline.power for the GQI step (fp16 pieces): fibers_jl_amd.energy.gqi_power_roofline on the probes' Joules per unit"""
    from fibers_jl_amd import energy as en
    if not probe or not steps or not steps.get("gqi_fp16x2"):
        return None
    g = steps["gqi_fp16x2"]
    idle = steps.get("idle_watts") or probe.get("idle", {}).get("watts") or 260.0

    def pj(mode):
        r = probe.get(mode)
        return r["pj_per_unit_above_idle"] * 1e-12 if r else None
    unit = dict(hbm_bytes=pj("hbm_rw_gqi"), mfma_flops=pj("mfma_reg_sleep0"), lds_fragment_bytes=pj("lds_read_sleep0"),
                l2_to_lds_bytes=pj("ldsdma_l2"), lds_other_bytes=pj("lds_write_read"), valu_wave_instructions=pj("valu_sleep0"))
    out = en.gqi_power_roofline(unit, NVOX, g.get("kernel_ms"), g["ms_per_step"], g["joules_per_step"], idle)
    out["board_watts"] = g["watts"]
    ess = probe.get("essential_gqi_step")
    if ess and ess.get("rate_per_s"):
        # NOT a floor: a straightforward kernel that does one step's HBM bytes + MFMAs side by side and nothing else turned out slower than
        # the product kernel and below the cap (it is latency / issue bound, not power bound) -- recorded as what it is
        out["essential_probe"] = dict(ms_per_step_equivalent=1e3 / ess["rate_per_s"], watts=ess["watts"], in_kernel_clock_ghz=ess.get("in_kernel_clock_ghz"),
                                      note="one launch = the step's algorithmic HBM bytes + executed MFMAs, nothing else (tools/probes/energy_probe.hip k_essential): "
                                           "slower than the product kernel and below the power cap, so it bounds nothing; kept for the record")
    out["probe_operating_points"] = {m: dict(watts=probe[m]["watts"], in_kernel_clock_ghz=probe[m].get("in_kernel_clock_ghz"), smu_sclk_mhz=probe[m].get("sclk_smi_mhz"),
                                               rate_per_s=probe[m].get("rate_per_s"), unit=probe[m].get("unit"))
                                      for m in ("hbm_rw_gqi", "mfma_reg_sleep0", "lds_read_sleep0", "ldsdma_l2", "lds_write_read", "valu_sleep0", "essential_gqi_step",
                                                "essential_gqi_step_lds_fragments", "mfma_lds2_sleep0", "mfma_lds1_sleep0", "spin") if m in probe}
    if clock and clock.get("gqi_fused"):
        out["in_kernel_clock_ghz_diagnostic_build"] = clock["gqi_fused"].get("clock_ghz_median")
        out["smu_sclk_mhz_same_seconds"] = clock["gqi_fused"].get("smu_sclk_mhz_mean")
    out["smu_sclk_mhz_product_kernel"] = g.get("sclk_mhz_mean")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05", "energy_model.json"))
    ap.add_argument("--skip-probe", action="store_true")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    doc = {}
    if not args.skip_probe:
        doc["probe"] = run_probe(args.seconds)                 # (a child process; this one has not touched the GPU yet)
        json.dump(doc, open(args.out, "w"), indent=1)
    try:
        o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_clock.py"), "--seconds", str(max(2.0, args.seconds)), "--kernels", "fused,dsi"],
                           capture_output=True, text=True, timeout=900)
        doc["kernel_clock"] = json.loads([ln for ln in o.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:                                     # noqa: BLE001
        doc["kernel_clock"] = dict(error=repr(e))
    json.dump(doc, open(args.out, "w"), indent=1)
    doc["steps"] = product_steps(args.seconds)
    doc["gqi_model"] = compose(doc.get("probe"), doc["steps"], doc.get("kernel_clock"))
    doc["source"] = "tools/energy_model.py --seconds %g (tools/probes/energy_probe.hip + the product's steps under the board's energy counter)" % args.seconds
    json.dump(doc, open(args.out, "w"), indent=1)
    print(json.dumps(doc["gqi_model"], indent=1))


if __name__ == "__main__":
    main()
