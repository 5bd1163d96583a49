#!/usr/bin/env python3
"""Accuracy of the ODF contraction kernels against a float64 contraction of the same float32 operands.

  f32     odf_gemm_kernel   v_mfma_f32_32x32x2_f32 (k-ordered f32 fma chain)
  bf16x3  odf_gemm3_kernel  three exact bf16 pieces per operand, six v_mfma_f32_32x32x16_bf16 per 16 frames (FIBERS_ODF_FORMAT=bf16x3)
  fp16x2  odf_gemm3_kernel  two fp16 pieces per operand (23 significant bits, per-voxel power-of-two sample scale), three
                            v_mfma_f32_32x32x16_f16 per 16 frames: the default

For every voxel the error is |odf - A64 @ max(s,0)64| relative to the voxel's largest |odf| entry; the script
prints max / mean / rms over a volume for GQI (270 frames) and DSI (515 frames, folded), plus how many peak
vertices differ between the two kernels.  Run on the GPU box: python tools/gemm_accuracy.py [n]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(kind, mode, bval, bvec, dwi, mask, sph):
    import fibers_jl_amd as fj
    os.environ.pop("FIBERS_ODF_FORMAT", None)
    if mode in ("f32", "bf16x3"): os.environ["FIBERS_ODF_FORMAT"] = mode
    plan = fj.OdfPlan(kind, bval, bvec, sph, sigma=1.25, hann_width=32, device=0)
    out = fj.odf_rec_device(plan, dwi, mask)
    A = plan.matrix()
    res = dict(odf=out["odf"].cpu().numpy().astype(np.float64), peak=out["peak"][0].cpu().numpy(), A=A,
               pdf=out["pdf"].cpu().numpy().astype(np.float64) if kind == "dsi" else None)
    plan.close()
    return res


def main():
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    shape = (n, n, n)
    nvox = n ** 3
    dev = torch.device("cuda", 0)
    sph = fj.sphere_642
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    for kind in ("gqi", "dsi"):
        bval, bvec = phantom.scheme_gqi() if kind == "gqi" else phantom.scheme_dsi()
        dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3, device=dev)
        s64 = np.maximum(dwi.cpu().numpy().astype(np.float64), 0.0)
        res = {m: run(kind, m, bval, bvec, dwi, mask, sph) for m in ("f32", "bf16x3", "fp16x2")}
        A64 = res["f32"]["A"].astype(np.float64)
        nrow0 = A64.shape[0] - sph.nvert
        ref = A64[nrow0:] @ s64
        if kind == "dsi":                                    # p ./ sum(p): sum(p) = nfft^3 H(0) s(q=0)
            i0 = int(np.argmin(bval))
            # the scale is whatever the kernel used; recover it from the f32 result to compare the contraction only
            scale = (res["f32"]["odf"] * ref).sum(0) / np.maximum((ref * ref).sum(0), 1e-300)
            ref = ref * scale
        vmax = np.abs(ref).max(0)
        print("%s  %d^3 x %d frames, %d rows" % (kind, n, len(bval), A64.shape[0]))
        for m in ("f32", "bf16x3", "fp16x2"):
            e = np.abs(res[m]["odf"] - ref) / vmax
            print("  %-7s rel. error of odf vs float64: max %.3e  mean %.3e  rms %.3e" % (m, e.max(), e.mean(), np.sqrt((e * e).mean())))
        for m in ("f32", "fp16x2"):
            d = np.abs(res[m]["odf"] - res["bf16x3"]["odf"]) / vmax
            pk = np.mean(np.any(res[m]["peak"] != res["bf16x3"]["peak"], axis=0))
            print("  %s vs bf16x3: max rel. difference %.3e; first-peak vertex differs in %.4f %% of voxels" % (m, d.max(), 100 * pk))


if __name__ == "__main__":
    main()
