#!/usr/bin/env python3
"""The drop-in path file to file, on the tutorial's shape (docs/tutorial.ipynb cells 7, 29, 37, 57: a 140 x 140 x 92 x 198 DWI series):

    <dwi>.nii + .bvals / .bvecs  --mri_read-->  dti_fit + gqi_rec  --stream-->  eigvec1 / FA lines and GQI peak / QA lines  --trk_write-->  .trk

(reference: mri.jl:611-733, dti.jl:221, gqi.jl:109-225, stream.jl:730-790, trk.jl:433-495).  Two ways through the same library:

  device  what rows N1 / N2 were built for: the .nii is memory-mapped (the fits' host tier gathers its chunks from the page cache into the
          pinned ring: file -> pinned -> HBM), the orientation field is repacked and tracked on the GPU and the pack kernel emits the .trk
          body (fibd_stream_pack_trk), which goes to the file in one write;
  host    the same fits on a volume read into memory (load_nifti without mmap), fib_stream into host arrays (a Tract), trk.py's NumPy
          trk_write.

Both write byte-identical .trk files (tests/test_gpu_pipeline.py).  Used by bench.py's `pipeline` leg (tools/bench_legs.py) and runnable:
    python tools/pipeline.py [--shape 140,140,92] [--keep DIR]
Nothing here touches the oracle."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TUTORIAL_SHAPE = (140, 140, 92)
SUB = np.array([[0.1, -0.2, 0.3]], np.float32)             # one explicit sub-voxel offset (the reference draws its own from the global RNG)


def scheme(nb0=18, ndir=90, shells=(1500.0, 3000.0), seed=7):
    """18 x b ~ 5 + 90 directions x {1500, 3000} = 198 frames (not a multiple of 16)"""
    from fibers_jl_amd import phantom
    return phantom.scheme_gqi(nb0, ndir, shells, seed)


def write_inputs(workdir, shape=TUTORIAL_SHAPE, dev=None, bval=None, bvec=None, seed=7):
    """synthetic DWI series + ball mask as uncompressed float32 / uint8 NIfTI-1 files with FSL b-tables; returns (dwi path, mask path)"""
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom
    dev = dev or torch.device("cuda", 0)
    if bval is None:
        bval, bvec = scheme()
    d, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=seed, device=dev)
    vol = d.cpu().numpy().T.reshape(tuple(shape) + (len(bval),), order="F")          # planar [nvol, nvox] == Fortran [nx, ny, nz, nvol]
    del d
    M = np.diag([1.25, 1.25, 1.25, 1.0]).astype(np.float32)
    M[:3, 3] = [-87.5, -87.5, -57.5]
    dwi = fj.MRI(vol, bval, bvec, volres=(1.25, 1.25, 1.25), vox2ras=M)
    p_dwi, p_mask = os.path.join(workdir, "dwi.nii"), os.path.join(workdir, "mask.nii")
    fj.mri_write(dwi, p_dwi)
    mask = fj.MRI(phantom.ball_mask(*shape), volres=(1.25, 1.25, 1.25), vox2ras=M)
    fj.mri_write(mask, p_mask)
    return p_dwi, p_mask


def _planar(vol4):
    """[nx, ny, nz, k] Fortran -> contiguous [k, nvox]"""
    k = vol4.shape[3]
    return np.ascontiguousarray(vol4.reshape(-1, k, order="F").T)


def run(p_dwi, p_mask, outdir, mode="device", device=0, track_kw=None):
    """one pass of the pipeline; returns dict(read_ms, fit_ms, track_ms, write_ms, total_ms, lines, points, files, fits)"""
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import trk
    kw = dict(len_min=3, ang_thresh=45, step_size=0.5, smooth_coeff=0.2)
    kw.update(track_kw or {})
    dev = torch.device("cuda", device)
    t = {}
    t0 = time.perf_counter()
    dwi = fj.mri_read(p_dwi, mmap=(mode == "device"))
    mask = fj.mri_read(p_mask)
    t["read_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    dti = fj.dti_fit(dwi, mask, device=device)
    gqi = fj.gqi_rec(dwi, mask, device=device)
    t["fit_ms"] = (time.perf_counter() - t0) * 1e3
    shape = dwi.volsize
    files = [os.path.join(outdir, "dti_%s.trk" % mode), os.path.join(outdir, "gqi_%s.trk" % mode)]
    track_s = write_s = 0.0
    lines = points = 0
    if mode == "device":
        m8 = torch.from_numpy(np.ascontiguousarray((mask.vol[..., 0] > 0).reshape(-1, order="F").astype(np.uint8))).to(dev)
        sub = torch.from_numpy(SUB).to(dev)
        jobs = [([dti.eigvec1], None, dti.fa), (gqi.peak, gqi.qa, None)]
        for (ov, f, fa), out in zip(jobs, files):
            t0 = time.perf_counter()
            ovd = [torch.from_numpy(_planar(o.vol)).to(dev) for o in ov]
            fd_ = None if f is None else [torch.from_numpy(_planar(x.vol)[0]).to(dev) for x in f]
            fad = None if fa is None else torch.from_numpy(_planar(fa.vol)[0]).to(dev)
            field, mout = fj.stream_field_device(ovd, f=fd_, f_thresh=0.03, fa=fad, fa_thresh=0.1, mask=m8)
            seeds = torch.nonzero(mout).flatten()
            tm = {}
            r = trk.stream_to_trk(out, field, shape, seeds, sub, ref=mask, timings=tm, **kw)
            track_s += tm["device_done"] - t0
            write_s += tm["file_done"] - tm["device_done"]
            lines += r["nlines"]; points += r["npoints"]
            del field, mout, seeds, ovd, fd_, fad
    else:
        for (ov, f, fa), out in zip([(dti.eigvec1, None, dti.fa), (gqi.peak, gqi.qa, None)], files):
            t0 = time.perf_counter()
            tr = fj.stream(ov, f=f, f_thresh=0.03, fa=fa, fa_thresh=0.1, mask=mask, sublist=SUB, device=device, **kw)
            t1 = time.perf_counter()
            fj.trk_write(tr, out, ref=mask)
            t2 = time.perf_counter()
            track_s += t1 - t0; write_s += t2 - t1
            lines += tr.nstr; points += int(tr.npts.sum())
            del tr
    t["track_ms"], t["write_ms"] = track_s * 1e3, write_s * 1e3
    t["total_ms"] = t["read_ms"] + t["fit_ms"] + t["track_ms"] + t["write_ms"]
    t.update(lines=lines, points=points, files=files, fits=(dti, gqi), mask=mask)
    return t


def measure(shape=TUTORIAL_SHAPE, dev=None, keep=None, reps=2):
    """writes the inputs, runs both modes `reps` times (the first pass of each also warms plans, the pinned ring and the page cache) and
    reports each mode's LAST pass; the .trk files of the two modes are compared byte for byte"""
    import torch
    dev = dev or torch.device("cuda", 0)
    work = keep or tempfile.mkdtemp(prefix="fibers_pipeline_", dir=os.environ.get("TMPDIR") or None)
    os.makedirs(work, exist_ok=True)
    try:
        p_dwi, p_mask = write_inputs(work, shape, dev)
        res = {}
        for mode in ("device", "host"):
            for _ in range(reps):
                r = run(p_dwi, p_mask, work, mode=mode, device=dev.index or 0)
                r.pop("fits"); r.pop("mask")
            res[mode] = r
        same = all(open(a, "rb").read() == open(b, "rb").read() for a, b in zip(res["device"]["files"], res["host"]["files"]))
        d, h = res["device"], res["host"]
        out = dict(shape=list(shape), frames=len(scheme()[0]), input_bytes=os.path.getsize(p_dwi), trk_bytes=sum(os.path.getsize(f) for f in d["files"]),
                   read_ms=d["read_ms"], fit_ms=d["fit_ms"], track_ms=d["track_ms"], write_ms=d["write_ms"], total_ms=d["total_ms"],
                   lines=d["lines"], points=d["points"],
                   host_path_read_ms=h["read_ms"], host_path_fit_ms=h["fit_ms"], host_path_track_ms=h["track_ms"], host_path_write_ms=h["write_ms"],
                   host_path_total_ms=h["total_ms"], trk_files_identical=bool(same))
        return out
    finally:
        if not keep:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="140,140,92")
    ap.add_argument("--keep", default=None)
    a = ap.parse_args()
    print(json.dumps(measure(tuple(int(v) for v in a.shape.split(",")), keep=a.keep)))
