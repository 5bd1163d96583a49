"""The consumer of the pinning hook (julia/make_reference_fixtures.jl): when a directory of outputs computed BY THE REFERENCE on the
golden fixtures' inputs exists (tests/golden/reference/, or $FIBERS_REFERENCE_FIXTURES), the oracle and -- with -m gpu -- the HIP path
are compared with it at the tolerances of SURVEY.md 8d.  Without it these tests are skipped and parity stays "unpinned".
The plumbing itself (exchange format, loaders, comparisons) is exercised on every CPU run with the oracle standing in for the reference."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import export_raw  # noqa: E402
import refio  # noqa: E402
from util import assert_dti_close, peak_mismatches_are_ties  # noqa: E402

REF_DIR = os.environ.get("FIBERS_REFERENCE_FIXTURES", os.path.join(HERE, "golden", "reference"))
HAVE_REF = os.path.isfile(os.path.join(REF_DIR, "dti_8x8x8x7", "meta.txt"))
CASES = sorted(export_raw.CASES)


# ---- what the two candidates compute, keyed like the reference's files ---------------------------------------------------------------
def oracle_outputs(name, orc, fj, tmp):
    export_raw.oracle_as_reference(tmp) if not os.path.isdir(os.path.join(tmp, name)) else None
    return refio.read_case(os.path.join(tmp, name))


def gpu_outputs(name, fj):
    kind = export_raw.CASES[name][0]
    g = export_raw.load(name)
    out = {}
    if kind == "dti":
        d = fj.dti_fit(fj.MRI(g["dwi"], g["bval"], g["bvec"]), fj.MRI(g["mask"]))
        out = {k: getattr(d, k).vol for k in fj.dti.DTI_FIELDS}
        for k in ("s0", "eigval1", "eigval2", "eigval3", "rd", "md", "fa"):
            out[k] = out[k][..., 0]
        adc, s0 = fj.adc_fit(fj.MRI(g["dwi"], g["bval"], g["bvec"]), fj.MRI(g["mask"]))
        out.update(adc=adc.vol[..., 0], adc_s0=s0.vol[..., 0])
    elif kind in ("gqi", "dsi"):
        sph = getattr(fj, str(g["sphere"])) if kind == "gqi" else fj.sphere_642
        r = (fj.gqi_rec(fj.MRI(g["dwi"], g["bval"], g["bvec"]), fj.MRI(g["mask"]), sph, float(g["sigma"])) if kind == "gqi"
             else fj.dsi_rec(fj.MRI(g["dwi"], g["bval"], g["bvec"]), fj.MRI(g["mask"]), sph, int(g["hann_width"])))
        out = dict(odf=r.odf.vol)
        if kind == "dsi":
            out["pdf"] = r.pdf.vol
        for k in range(3):
            out["peak%d" % (k + 1)] = r.peak[k].vol
            out["qa%d" % (k + 1)] = r.qa[k].vol[..., 0]
    elif kind == "peaks":
        import torch
        from fibers_jl_amd import phantom
        bval, bvec = phantom.scheme_gqi(2, 6, (1000.0,), 1)
        plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642)
        top, nvalid = fj.find_peaks_device(plan, torch.from_numpy(g["odf"]).cuda())
        torch.cuda.synchronize()
        out = dict(isort_top=top.cpu().numpy(), nvalid=nvalid.cpu().numpy())
    elif kind == "stream":
        kw = dict(f_thresh=float(g["kw_f_thresh"]), fa_thresh=float(g["kw_fa_thresh"]), len_min=int(g["kw_len_min"]),
                  ang_thresh=float(g["kw_ang_thresh"]), step_size=float(g["kw_step_size"]), smooth_coeff=float(g["kw_smooth_coeff"]))
        tr = fj.stream([fj.MRI(o) for o in g["ovec"]], f=[fj.MRI(x) for x in g["f"]], fa=fj.MRI(g["fa"]), mask=fj.MRI(g["mask"]),
                       seed=fj.MRI(g["seed"]), sublist=g["sublist"], **kw)
        t1 = fj.stream(fj.MRI(g["ovec"][0]), mask=fj.MRI(g["mask"]), sublist=g["sublist"])
        out = dict(multi_npts=tr.npts, multi_xyz=tr.xyz, single_npts=t1.npts, single_xyz=t1.xyz)
    elif kind == "micro":
        vol = fj.MRI(g["ovec"])
        vol.volres = (0.01, 0.01, 0.01)
        tr = fj.stream(vol, f=fj.MRI(g["f"]), mask=fj.MRI(g["mask"]), seed=fj.MRI(g["seed"]), sublist=g["sublist"],
                       f_thresh=float(g["kw_f_thresh"]), ang_thresh=float(g["kw_ang_thresh"]), step_size=float(g["kw_step_size"]),
                       smooth_coeff=float(g["kw_smooth_coeff"]), search_dist=int(g["kw_search_dist"]), search_ang=float(g["kw_search_ang"]),
                       len_max=int(g["kw_len_max"]))
        out = dict(npts=tr.npts, xyz=tr.xyz)
    return out


# ---- the comparison, at the tolerances of SURVEY.md 8d -------------------------------------------------------------------------------
def compare(name, got, ref, fj):
    kind = export_raw.CASES[name][0]
    g = export_raw.load(name)
    if kind == "dti":
        loose = "nonpositive" in name       # (the per-voxel pinv branch of dti.jl:297-303: rel 5e-3, tests/test_golden.py)
        assert_dti_close(got, ref, g["mask"], label=name,
                         **(dict(s0_rtol=2e-3, ev_rtol=5e-3, ev_atol=2e-6, fa_atol=5e-3, vec_tol=1e-3, gap=0.2) if loose else {}))
        np.testing.assert_allclose(got["adc"], ref["adc"], rtol=2e-3, atol=1e-7)
        np.testing.assert_allclose(got["adc_s0"], ref["adc_s0"], rtol=2e-3)
    elif kind in ("gqi", "dsi"):
        sph = getattr(fj, str(g["sphere"])) if kind == "gqi" else fj.sphere_642
        for key, tol in (("odf", 1e-4), ("pdf", 1e-4)):
            if key in ref:
                scale = np.abs(ref[key]).max(axis=3, keepdims=True) + 1e-30
                assert np.array_equal(np.isnan(got[key]), np.isnan(ref[key])), "%s %s: NaN pattern" % (name, key)
                err = np.nanmax(np.abs(got[key] - ref[key]) / scale) if np.isfinite(ref[key]).any() else 0.0
                assert err <= tol, "%s %s: %g of the voxel maximum" % (name, key, err)
        nv = sph.nvert
        peak_mismatches_are_ties(ref["odf"], [ref["peak%d" % k] for k in (1, 2, 3)], [got["peak%d" % k] for k in (1, 2, 3)],
                                 np.asarray(sph.vertices, np.float32)[:nv], faces=np.asarray(sph.faces))
        for k in (1, 2, 3):
            same = np.all(got["peak%d" % k] == ref["peak%d" % k], axis=3)
            np.testing.assert_allclose(got["qa%d" % k][same], ref["qa%d" % k][same], atol=1e-4, rtol=1e-5, equal_nan=True)
    elif kind == "peaks":
        assert np.array_equal(got["isort_top"], ref["isort_top"]) and np.array_equal(got["nvalid"], ref["nvalid"])
    elif kind == "stream":
        for pre in ("multi", "single"):
            assert np.array_equal(got[pre + "_npts"], ref[pre + "_npts"]), "%s %s: line lengths differ" % (name, pre)
            assert np.abs(got[pre + "_xyz"] - ref[pre + "_xyz"]).max() <= 1e-3, "%s %s: points differ by more than 1e-3 voxel" % (name, pre)
    elif kind == "micro":
        assert np.array_equal(got["npts"], ref["npts"]) and np.abs(got["xyz"] - ref["xyz"]).max() <= 1e-3


# ---- tests ----------------------------------------------------------------------------------------------------------------------------------
def test_exchange_format_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    arrs = dict(a=rng.random((3, 4, 5)).astype(np.float32), b=np.arange(7, dtype=np.int64), m=(rng.random((2, 3)) < 0.5).astype(np.uint8),
                x=np.asfortranarray(rng.random((4, 3))))
    refio.write_case(str(tmp_path / "c"), arrs, dict(kind="t", sigma=1.25, n=3))
    back = refio.read_case(str(tmp_path / "c"))
    for k, v in arrs.items():
        assert back[k].dtype == v.dtype and np.array_equal(back[k], v), k
    assert back["kind"] == "t" and back["sigma"] == 1.25 and back["n"] == 3
    raw = np.fromfile(str(tmp_path / "c" / "a.bin"), "<f4")
    assert np.array_equal(raw, arrs["a"].ravel(order="F"))          # column-major on disk: what Julia's read! expects


def test_plumbing_with_the_oracle_standing_in(tmp_path, orc, fj):
    """inputs exported, outputs written in the reference's file layout (by the oracle), read back and run through `compare`"""
    export_raw.export_inputs(str(tmp_path / "raw"))
    export_raw.oracle_as_reference(str(tmp_path / "ref"))
    for name in CASES:
        inp = refio.read_case(str(tmp_path / "raw" / name))
        g = export_raw.load(name)
        for k in export_raw.CASES[name][1]:
            assert np.array_equal(inp[k], g[k]) and inp[k].dtype == g[k].dtype, (name, k)
        ref = refio.read_case(str(tmp_path / "ref" / name))
        compare(name, ref, ref, fj)


@pytest.mark.skipif(not HAVE_REF, reason="no reference outputs (run julia/make_reference_fixtures.jl on a machine with Julia): parity unpinned")
@pytest.mark.parametrize("name", CASES)
def test_oracle_against_the_reference(name, orc, fj, tmp_path):
    export_raw.oracle_as_reference(str(tmp_path))
    compare(name, refio.read_case(str(tmp_path / name)), refio.read_case(os.path.join(REF_DIR, name)), fj)


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_REF, reason="no reference outputs (run julia/make_reference_fixtures.jl on a machine with Julia): parity unpinned")
@pytest.mark.parametrize("name", CASES)
def test_gpu_against_the_reference(name, fj):
    compare(name, gpu_outputs(name, fj), refio.read_case(os.path.join(REF_DIR, name)), fj)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_outputs_have_the_reference_files_layout(name, fj, orc, tmp_path):
    """without reference files: the GPU side of the comparison is exercised against the oracle stand-in, so that the day the
    reference's files arrive the only new thing is the files"""
    export_raw.oracle_as_reference(str(tmp_path))
    compare(name, gpu_outputs(name, fj), refio.read_case(str(tmp_path / name)), fj)
