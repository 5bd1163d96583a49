#!/usr/bin/env python3
"""mask_compact_kernel's clearing workgroups decide, each from its own bounded poll, whether the outputs outside the mask are cleared as
slices of whole arrays or span by span; a workgroup whose poll times out covers its share of BOTH partitions (a mixture of
decisions must not leave stale values behind).  The DIAGNOSTIC build can force the time-out in every n-th workgroup
(FIBERS_COMPACT_UNKNOWN=n): with NaN-filled output buffers, masks on both sides of the 25 % threshold and n = 2, 3, 7 the outputs must
be bit-identical to the unforced run and exactly zero outside the mask.  Exit code 0 = all good.  usage: compact_mixture_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the switch this tool flips exists in the DIAGNOSTIC build only (csrc/common.h ab_env; make -C fibers.jl_amd/csrc stamp)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so"))
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    bad = 0
    for kind, shape in (("gqi", (96, 96, 90)), ("dsi", (64, 64, 60))):
        bval, bvec = phantom.scheme_gqi() if kind == "gqi" else phantom.scheme_dsi()
        dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3, device=dev)
        nvox = dwi.shape[1]
        g = torch.Generator(device=dev); g.manual_seed(1)
        masks = dict(ball=phantom.ball_mask_torch(shape, dev).reshape(-1),                                # 64 % outside: whole-array clearing
                     mostly_inside=(torch.rand(nvox, device=dev, generator=g) < 0.9).to(torch.uint8),     # 10 % outside: span by span
                     slab=torch.zeros(nvox, dtype=torch.uint8, device=dev))
        masks["slab"][nvox // 3: nvox // 3 + nvox // 5] = 1                                              # 80 % outside, long runs
        plan = fj.OdfPlan(kind, bval, bvec, fj.sphere_642, device=0)
        for mname, mask in masks.items():
            res = {}
            for n in (0, 2, 3, 7):
                if n:
                    os.environ["FIBERS_COMPACT_UNKNOWN"] = str(n)
                else:
                    os.environ.pop("FIBERS_COMPACT_UNKNOWN", None)
                fj.odf_rec_device(plan, dwi, mask)                                                        # (the list unit follows the previous call's mask: settle it)
                out = fj.odf_rec_device(plan, dwi, mask)
                for t in [out["odf"]] + out["peak"] + out["qa"] + ([out["pdf"]] if "pdf" in out else []):
                    t.fill_(float("nan"))
                out = fj.odf_rec_device(plan, dwi, mask, out=out, normalize=False)
                torch.cuda.synchronize()
                res[n] = [out["odf"].clone()] + [t.clone() for t in out["peak"]] + [t.clone() for t in out["qa"]] + ([out["pdf"].clone()] if "pdf" in out else [])
                dead = mask == 0
                for t in res[n]:
                    v = t.reshape(-1, nvox)[:, dead]
                    if not bool((v == 0).all()):
                        print("FAIL %s %s n=%d: %d values outside the mask are not zero" % (kind, mname, n, int((v != 0).sum() + torch.isnan(v).sum())))
                        bad += 1
            for n in (2, 3, 7):
                same = all(torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)) for a, b in zip(res[0], res[n]))
                print("%s %-13s n=%d  identical to the unforced run: %s" % (kind, mname, n, same), flush=True)
                bad += 0 if same else 1
        os.environ.pop("FIBERS_COMPACT_UNKNOWN", None)
        plan.close()
    print("compact mixture check:", "ok" if bad == 0 else "%d FAILURES" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
