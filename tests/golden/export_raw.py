#!/usr/bin/env python3
"""Writes the INPUTS of the golden fixtures (tests/golden/*.npz) in the exchange format of refio.py, for
julia/make_reference_fixtures.jl to read on a machine that has Julia and the reference:

    python tests/golden/export_raw.py                      # -> tests/golden/raw/<case>/   (git-ignored: derived data)
    julia --threads 1 julia/make_reference_fixtures.jl <path-to-Fibers.jl> tests/golden/raw tests/golden/reference
    python -m pytest tests/test_reference_fixtures.py      # oracle (and, with -m gpu, the HIP path) against the reference's own outputs

`--oracle-as-reference DIR` writes what the Julia script would write, computed by the ORACLE: a stand-in that lets the consuming
tests run where there is no Julia (tests/test_reference_fixtures.py uses it to test its own plumbing; it pins nothing)."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import refio  # noqa: E402

CASES = {   # fixture -> (kind, input arrays, scalars)
    "dti_8x8x8x7": ("dti", ["dwi", "mask", "bval", "bvec"], []),
    "dti_6x5x4x33_nonpositive": ("dti", ["dwi", "mask", "bval", "bvec"], []),
    "gqi_6x6x6x63_sphere642": ("gqi", ["dwi", "mask", "bval", "bvec"], ["sphere", "sigma"]),
    "gqi_5x4x3x63_sphere362": ("gqi", ["dwi", "mask", "bval", "bvec"], ["sphere", "sigma"]),
    "dsi_3x2x2x515": ("dsi", ["dwi", "mask", "bval", "bvec"], ["hann_width"]),
    "find_peaks_ties_sphere642": ("peaks", ["odf"], []),
    "stream_12": ("stream", ["ovec", "f", "fa", "mask", "seed", "sublist"],
                  ["kw_f_thresh", "kw_fa_thresh", "kw_len_min", "kw_ang_thresh", "kw_step_size", "kw_smooth_coeff"]),
    "stream_micro_14": ("micro", ["ovec", "f", "mask", "seed", "sublist"],
                        ["kw_f_thresh", "kw_ang_thresh", "kw_step_size", "kw_smooth_coeff", "kw_search_dist", "kw_search_ang", "kw_len_max"]),
}


def load(name):
    return dict(np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False))


def scalar(v):
    v = np.asarray(v)
    return v.item() if v.ndim == 0 else v


def export_inputs(outdir):
    for name, (kind, arrs, scs) in CASES.items():
        g = load(name)
        refio.write_case(os.path.join(outdir, name), {k: g[k] for k in arrs}, dict(kind=kind, **{k: scalar(g[k]) for k in scs}))
        print("inputs:", name)


def oracle_as_reference(outdir):
    """the files julia/make_reference_fixtures.jl writes, with the oracle in the reference's place (same keys, same layouts)"""
    from oracle import oracle as orc
    import fibers_jl_amd as fj
    for name, (kind, _, _) in CASES.items():
        g = load(name)
        out = {}
        if kind == "dti":
            r = orc.dti_fit(g["dwi"], g["mask"], g["bval"], g["bvec"], nthreads=2)
            adc, s0 = orc.adc_fit(g["dwi"], g["mask"], g["bval"], nthreads=2)
            out = {k: r[k] for k in ("s0", "eigval1", "eigval2", "eigval3", "eigvec1", "eigvec2", "eigvec3", "rd", "md", "fa")}
            out.update(adc=adc, adc_s0=s0)
        elif kind in ("gqi", "dsi"):
            sph = getattr(fj, str(g["sphere"])) if kind == "gqi" else fj.sphere_642
            r = (orc.gqi_rec(g["dwi"], g["mask"], g["bval"], g["bvec"], sph.vertices, sph.faces, float(g["sigma"]), nthreads=2) if kind == "gqi"
                 else orc.dsi_rec(g["dwi"], g["mask"], g["bval"], g["bvec"], sph.vertices, sph.faces, int(g["hann_width"]), nthreads=2))
            out = dict(odf=r["odf"])
            if kind == "dsi":
                out["pdf"] = r["pdf"]
            for k in range(3):
                out["peak%d" % (k + 1)] = r["peak"][k]
                out["qa%d" % (k + 1)] = r["qa"][k]
        elif kind == "peaks":
            out = dict(isort_top=g["isort_top"], nvalid=g["nvalid"])
        elif kind == "stream":
            out = dict(multi_npts=g["multi_npts"], multi_xyz=g["multi_xyz"], single_npts=g["single_npts"], single_xyz=g["single_xyz"])
        elif kind == "micro":
            out = dict(npts=g["npts"], xyz=g["xyz"])
        refio.write_case(os.path.join(outdir, name), out, dict(kind=kind, source="oracle stand-in (NOT the reference)"))
        print("stand-in outputs:", name)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "raw"))
    ap.add_argument("--oracle-as-reference", default=None)
    a = ap.parse_args()
    export_inputs(a.out)
    if a.oracle_as_reference:
        oracle_as_reference(a.oracle_as_reference)
