#!/usr/bin/env python3
"""A/B of the two MFMA shapes of the split-bf16 contraction on the benchmark phantom, interleaved rounds in ONE process
(cdna_hip_programming.md rule 24): plan A = v_mfma_f32_16x16x32_bf16 (odf_gemm16_kernel, FIBERS_ODF_SHAPE16=1 at plan creation), plan B = the 32x32x16
kernels (default).  Prints per round the kernel time (hipEvents on the launch stream) and the
step wall time of both, then the agreement of their outputs.  usage: shape_ab.py [gqi|gqi_unfused|dsi] [rounds] [shape]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "gqi"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    shape = tuple(int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "140,140,140").split(","))
    dev = torch.device("cuda", 0)
    L = fj.lib()
    nvox = shape[0] * shape[1] * shape[2]
    sph = fj.sphere_642
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    if what.startswith("gqi"):
        bval, bvec = phantom.scheme_gqi()
        dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3, device=dev)
        mk = lambda: fj.OdfPlan("gqi", bval, bvec, sph, sigma=1.25, device=0)
    else:
        bval, bvec = phantom.scheme_dsi()
        dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=5, device=dev)
        mk = lambda: fj.OdfPlan("dsi", bval, bvec, sph, hann_width=32, device=0)
    if what == "gqi_unfused":
        os.environ["FIBERS_ODF_UNFUSED"] = "1"
    os.environ["FIBERS_ODF_SHAPE16"] = "1"
    pa = mk()
    os.environ.pop("FIBERS_ODF_SHAPE16", None)
    pb = mk()
    oa = fj.odf_rec_device(pa, dwi, mask, normalize=True)
    ob = fj.odf_rec_device(pb, dwi, mask, normalize=True)
    torch.cuda.synchronize()

    def run(plan, out, n=40):
        for _ in range(3):
            fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        t0 = time.perf_counter()
        for _ in range(n):
            fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n
        L.fib_profile_enable(0)
        ms, cnt = C.c_double(0), C.c_int64(0)
        L.fib_profile_get(b"odf_gemm", C.byref(ms), C.byref(cnt))
        pk, pc = C.c_double(0), C.c_int64(0)
        L.fib_profile_get(b"odf_peaks", C.byref(pk), C.byref(pc))
        return ms.value / max(cnt.value, 1), pk.value / max(pc.value, 1), wall * 1e3
    res = dict(what=what, shape=shape, rounds=[])
    for r in range(rounds):
        ka, qa_, wa = run(pa, oa)
        kb, qb_, wb = run(pb, ob)
        res["rounds"].append(dict(k16=ka, k32=kb, peaks16=qa_, peaks32=qb_, step16=wa, step32=wb))
        print("round %d: 16x16x32 kernel %.3f ms (peaks %.3f) step %.3f | 32x32x16 kernel %.3f ms (peaks %.3f) step %.3f | ratio kernel %.3f step %.3f"
              % (r, ka, qa_, wa, kb, qb_, wb, kb / ka, wb / wa), flush=True)
    k16 = np.median([x["k16"] for x in res["rounds"]]); k32 = np.median([x["k32"] for x in res["rounds"]])
    s16 = np.median([x["step16"] for x in res["rounds"]]); s32 = np.median([x["step32"] for x in res["rounds"]])
    res["median"] = dict(k16=k16, k32=k32, step16=s16, step32=s32, kernel_speedup=k32 / k16, step_speedup=s32 / s16)
    # agreement of the two shapes' outputs (the MFMA shapes sum their k terms in different orders: rounding-level differences)
    oda, odb = oa["odf"], ob["odf"]
    vmax = odb.abs().amax(dim=0).clamp_min(1e-30)
    res["odf_max_rel_diff_of_voxel_max"] = float(((oda - odb).abs().amax(dim=0) / vmax).max())
    pka, pkb = oa["peak"][0], ob["peak"][0]
    res["peak1_identical_frac"] = float((pka == pkb).all(dim=0).float().mean())
    res["qa1_max_abs_diff"] = float((oa["qa"][0] - ob["qa"][0]).abs().max())
    if "pdf" in oa and oa["pdf"] is not None:
        pm = ob["pdf"].abs().amax(dim=0).clamp_min(1e-30)
        res["pdf_max_rel_diff_of_voxel_max"] = float(((oa["pdf"] - ob["pdf"]).abs().amax(dim=0) / pm).max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
