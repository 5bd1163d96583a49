"""The drop-in path file to file (tools/pipeline.py): .nii + b-tables -> mri_read -> dti_fit + gqi_rec -> stream -> .trk
(mri.jl:611-733, dti.jl:221, gqi.jl:109, stream.jl:730-790, trk.jl:433-495; shape after docs/tutorial.ipynb cell 7: nz < nx, a frame
count that is not a multiple of 16).  The .trk files the GPU serialiser writes must be byte-identical (a) to the host path's files
(fib_stream into host arrays + trk.py's trk_write) and (b) to trk.py's serialisation of the ORACLE's lines traced on the fields the
fits produced."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
def test_pipeline_file_to_file_matches_the_host_path_and_the_oracle(fj, orc, tmp_path):
    import torch
    import pipeline as pl
    from fibers_jl_amd import phantom
    shape = (30, 28, 19)                                                    # nz < nx, odd nz, 15 960 voxels (not a multiple of 32 per slice)
    bval, bvec = phantom.scheme_gqi(3, 15, (1500.0, 3000.0), 7)             # 33 frames: not a multiple of 16
    work = str(tmp_path)
    p_dwi, p_mask = pl.write_inputs(work, shape, torch.device("cuda", 0), bval, bvec)
    assert os.path.getsize(p_dwi) == 352 + 4 * 33 * 30 * 28 * 19
    dev = pl.run(p_dwi, p_mask, work, mode="device")
    host = pl.run(p_dwi, p_mask, work, mode="host")
    assert isinstance(fj.mri_read(p_dwi, mmap=True).vol, np.memmap)         # (the device path's volume IS the file)
    for k in ("read_ms", "fit_ms", "track_ms", "write_ms", "total_ms"):
        assert dev[k] > 0 and host[k] > 0
    assert dev["lines"] == host["lines"] > 500 and dev["points"] == host["points"]
    for a, b in zip(dev["files"], host["files"]):
        assert open(a, "rb").read() == open(b, "rb").read(), (a, b)
    # the fits of the mapped file and of the in-memory copy are the same numbers
    for k in ("fa", "eigvec1", "s0"):
        assert np.array_equal(getattr(dev["fits"][0], k).vol, getattr(host["fits"][0], k).vol, equal_nan=True)
    assert np.array_equal(dev["fits"][1].odf.vol, host["fits"][1].odf.vol, equal_nan=True)

    # (b) the oracle's tracker on the SAME fields (tracking is chaotic at round() boundaries: it has to start from identical vectors)
    dti, gqi = dev["fits"]
    mask = dev["mask"]
    m3 = mask.vol[..., 0]
    cases = [(dict(ovec=dti.eigvec1.vol, fa=dti.fa.vol[..., 0]), dev["files"][0]),
             (dict(ovec=[p.vol for p in gqi.peak], f=[q.vol[..., 0] for q in gqi.qa]), dev["files"][1])]
    for kw, path in cases:
        ov = kw.pop("ovec")
        r = orc.stream(ov, pl.SUB, mask=m3, nthreads=4, **kw)
        tr = fj.Tract(xyz=r["xyz"], npts=r["npts"], volsize=shape, volres=mask.volres, vox2ras=mask.vox2ras)
        ref_file = path + ".oracle"
        assert not fj.trk_write(tr, ref_file, ref=mask)
        assert open(path, "rb").read() == open(ref_file, "rb").read(), path
    # and the fits themselves against the oracle (the tolerances of SURVEY 8d, as tests/test_gpu_dti.py / test_gpu_odf.py apply them)
    src = fj.mri_read(p_dwi)
    ro = orc.dti_fit(src.vol, m3, src.bval, src.bvec, nthreads=4)
    assert np.abs(dti.fa.vol[..., 0] - ro["fa"]).max() <= 1e-4
    rg = orc.gqi_rec(src.vol, m3, src.bval, src.bvec, fj.sphere_642.vertices, fj.sphere_642.faces, 1.25, nthreads=4)
    top = np.abs(rg["odf"]).max(axis=3, keepdims=True) + 1e-30
    assert (np.abs(gqi.odf.vol - rg["odf"]) / top).max() <= 2e-5


@pytest.mark.gpu
def test_pipeline_measure_reports_every_stage(fj):
    import pipeline as pl
    r = pl.measure((24, 20, 13), reps=1)
    assert r["trk_files_identical"] and r["lines"] > 0
    for k in ("read_ms", "fit_ms", "track_ms", "write_ms", "total_ms", "host_path_total_ms"):
        assert r[k] > 0
