#!/usr/bin/env python3
"""A/B check of the software-pipelined GQI kernel (odf_pipe_kernel, FIBERS_ODF_PIPE=1) against the fused kernel
(odf_gemm3_kernel<FUSE>) on the same device buffers: ODF rows, peaks, raw qa and odfmax must be bit-identical.
With --time also reports the step time of both (HIP events around 20 calls)."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402


def run(plan, dwi, mask, pipe):
    os.environ["FIBERS_ODF_PIPE"] = "1" if pipe else "0"
    out = fj.odf_rec_device(plan, dwi, mask, normalize=False)
    torch.cuda.synchronize()
    return out


def timed(plan, dwi, mask, pipe, n=20):
    os.environ["FIBERS_ODF_PIPE"] = "1" if pipe else "0"
    for _ in range(3):
        fj.odf_rec_device(plan, dwi, mask, normalize=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fj.odf_rec_device(plan, dwi, mask, normalize=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda", 0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    shape = tuple(int(x) for x in (args[:3] if len(args) >= 3 else (48, 48, 48)))
    bval, bvec = phantom.scheme_gqi()
    dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3, device=dev, noise_frac=0.1)
    nvox = dwi.shape[1]
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
    ok = True
    for label in ("ones", "ball", "poison", "noisy"):
        d = dwi.clone()
        mask = torch.ones(nvox, dtype=torch.uint8, device=dev) if label != "ball" else phantom.ball_mask_torch(shape, dev)
        if label == "poison":
            d[5, 100] = float("nan"); d[7, 2000] = float("inf"); d[:, 3000] = 0.0; d[:, 3001] = -1.0
            d[:, 5000:5064] = 1000.0      # identical isotropic voxels: ties everywhere
        if label == "noisy":              # many local maxima per voxel: list overflow and the >3-per-block path
            g = torch.Generator(device=dev); g.manual_seed(11)
            d = torch.rand(d.shape, generator=g, device=dev) * 100.0
        a = run(plan, d, mask, True)
        b = run(plan, d, mask, False)
        oa, ob = a["odf"].nan_to_num(nan=-7.0, posinf=-8.0, neginf=-9.0), b["odf"].nan_to_num(nan=-7.0, posinf=-8.0, neginf=-9.0)
        same_odf = torch.equal(oa, ob)
        if not same_odf:
            df = (oa != ob)
            print("   odf rows differing:", df.any(1).nonzero().flatten().tolist()[:24], "voxels", df.any(0).nonzero().flatten().tolist()[:12],
                  "n", int(df.sum()))
        same_pk = all(torch.equal(a["peak"][k].nan_to_num(nan=-7.0), b["peak"][k].nan_to_num(nan=-7.0)) for k in range(3))
        same_qa = all(torch.equal(a["qa"][k].nan_to_num(nan=-7.0), b["qa"][k].nan_to_num(nan=-7.0)) for k in range(3))
        om_a, om_b = a["odfmax"].cpu().numpy(), b["odfmax"].cpu().numpy()
        same_om = np.array_equal(om_a, om_b, equal_nan=True)
        print("%-7s odf %s peaks %s qa %s odfmax %s (%r)" % (label, same_odf, same_pk, same_qa, same_om, om_a))
        for k in range(3):
            bad = (a["peak"][k].nan_to_num(nan=-7.0) != b["peak"][k].nan_to_num(nan=-7.0)).any(0).nonzero().flatten()
            badq = (a["qa"][k].nan_to_num(nan=-7.0) != b["qa"][k].nan_to_num(nan=-7.0)).nonzero().flatten()
            if bad.numel() or badq.numel():
                print("   peak", k, "differs at", bad[:10].tolist(), "n", bad.numel(), "| qa at", badq[:10].tolist(), "n", badq.numel())
        ok &= same_odf and same_pk and same_qa and same_om
    print("OK" if ok else "MISMATCH")
    if "--time" in sys.argv:
        mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
        for rep in range(2):
            print("step ms: pipe %.3f  fused %.3f" % (timed(plan, dwi, mask, True), timed(plan, dwi, mask, False)))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
