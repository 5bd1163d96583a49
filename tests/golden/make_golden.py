#!/usr/bin/env python3
"""Generates the committed golden fixtures (inputs + expected outputs) under tests/golden/.

The reference (Julia) cannot be run in this image and ships no fixtures of its own, so these vectors
come from the CPU oracle (oracle/, the hand restatement of the reference algorithms): they pin the
oracle against regressions and give the GPU box a comparison that does not need the oracle at all.
Run:  python tests/golden/make_golden.py      (deterministic; seeded)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    print(name, {k: getattr(v, "shape", v) for k, v in arrs.items()})


def dti_case(name, shape, ndir, nb0, seed, nonpos):
    bval, bvec = phantom.scheme_dti(ndir, nb0, 1000.0, seed)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed, noise_frac=0.02, nonpositive_frac=nonpos)
    mask = (np.random.default_rng(seed + 1).random(shape) < 0.85).astype(np.uint8)
    if nonpos:
        dwi[0, 0, 0, :] = 0
        dwi[1, 0, 0, :nb0] = -1
        mask[:2, 0, 0] = 1
    out = orc.dti_fit(dwi, mask, bval, bvec, nthreads=2)
    adc, s0 = orc.adc_fit(dwi, mask, bval, nthreads=2)
    npart = out.pop("_npartial")
    save(name, dwi=dwi, mask=mask, bval=bval, bvec=bvec, npartial=npart, adc=adc, adc_s0=s0, **out)


def gqi_case(name, shape, sphere, seed):
    sph = getattr(fj, sphere)
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), seed)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed, noise_frac=0.02, nonpositive_frac=0.02, crossing=True)
    mask = (np.random.default_rng(seed + 1).random(shape) < 0.85).astype(np.uint8)
    dwi[0, 0, 0, :] = -1
    mask[0, 0, 0] = 1
    r = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=2)
    save(name, dwi=dwi, mask=mask, bval=bval, bvec=bvec, sphere=sphere, sigma=1.25, odf=r["odf"],
         peak=np.stack(r["peak"]), qa=np.stack(r["qa"]), odfmax=r["odfmax"])


def dsi_case(name, shape, seed):
    sph = fj.sphere_642
    bval, bvec = phantom.scheme_dsi()
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed, noise_frac=0.01, crossing=True)
    mask = np.ones(shape, np.uint8)
    dwi[0, 0, 0, :] = 0
    r = orc.dsi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 32, nthreads=2)
    save(name, dwi=dwi, mask=mask, bval=bval, bvec=bvec, hann_width=32, pdf=r["pdf"], odf=r["odf"],
         peak=np.stack(r["peak"]), qa=np.stack(r["qa"]), odfmax=r["odfmax"])


def peaks_case(name):
    sph = fj.sphere_642
    rng = np.random.default_rng(5)
    odf = rng.integers(0, 6, size=(sph.nvert, 97)).astype(np.float32)       # many exact ties
    odf[:, 0] = 0
    odf[:, 1] = -odf[:, 1]
    odf[:, 2] = 1.0
    f0 = orc.fold_faces(sph.faces, sph.nvert)
    top = np.zeros((3, odf.shape[1]), np.int32)
    nvalid = np.zeros(odf.shape[1], np.int32)
    for v in range(odf.shape[1]):
        isort, nv, _ = orc.find_peaks(odf[:, v], f0)
        top[:, v], nvalid[v] = isort[:3], nv
    save(name, odf=odf, isort_top=top, nvalid=nvalid)


def stream_case(name, n, seed):
    rng = np.random.default_rng(seed)
    x, y, z = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    c = (n - 1) / 2.0
    circ = np.stack([-(y - c), (x - c), 0.15 * np.ones_like(x, float)], -1)
    circ /= np.maximum(np.linalg.norm(circ, axis=-1, keepdims=True), 1e-9)
    wavy = np.stack([np.cos(0.2 * x + 0.1 * z), np.sin(0.2 * x + 0.1 * z), 0.3 * np.sin(y / 3.0)], -1)
    wavy /= np.linalg.norm(wavy, axis=-1, keepdims=True)
    noisy = wavy + 0.25 * rng.normal(size=wavy.shape)
    noisy /= np.linalg.norm(noisy, axis=-1, keepdims=True)
    ovs = [np.asfortranarray(v.astype(np.float32)) for v in (wavy, circ, noisy)]
    ovs[1][2:4, 2:4, 2:4] = 0
    fs = [np.asfortranarray(rng.uniform(0.0, 0.2, (n, n, n)).astype(np.float32)) for _ in range(3)]
    fa = np.asfortranarray(rng.uniform(0.0, 1.0, (n, n, n)).astype(np.float32))
    mask = (rng.random((n, n, n)) < 0.9).astype(np.uint8)
    seed_vol = (rng.random((n, n, n)) < 0.3).astype(np.uint8)
    sub = np.array([[0.1, -0.2, 0.3], [-0.45, 0.49, 0.0]], np.float32)
    kw = dict(f_thresh=0.05, fa_thresh=0.15, len_min=2, ang_thresh=60, step_size=0.75, smooth_coeff=0.35)
    multi = orc.stream(ovs, sub, f=fs, fa=fa, mask=mask, seed=seed_vol, nthreads=2, **kw)
    single = orc.stream(ovs[0], sub, mask=mask, nthreads=2)
    save(name, ovec=np.stack(ovs), f=np.stack(fs), fa=fa, mask=mask, seed=seed_vol, sublist=sub,
         multi_npts=multi["npts"], multi_seed_index=multi["seed_index"], multi_xyz=multi["xyz"],
         single_npts=single["npts"], single_seed_index=single["seed_index"], single_xyz=single["xyz"],
         **{"kw_" + k: v for k, v in kw.items()})


def micro_case(name, n, seed):
    """microscopy regime (stream.jl:547-619): smooth random unit field, mask holes, thresholded vectors"""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    g = rng.normal(size=(n, n, n, 3))
    for c in range(3):
        g[..., c] = gaussian_filter(g[..., c], 2.0)
    g[..., 0] += 0.15
    g /= np.linalg.norm(g, axis=3, keepdims=True)
    ov = np.asfortranarray(g.astype(np.float32))
    mask = (rng.random((n, n, n)) < 0.93).astype(np.uint8)
    f = np.asfortranarray(rng.random((n, n, n)).astype(np.float32))
    seed_vol = np.zeros((n, n, n), np.uint8)
    seed_vol[2::5, 3::4, 1::6] = 1
    sub = np.zeros((1, 3), np.float32)
    kw = dict(f_thresh=0.05, ang_thresh=20, step_size=1.0, smooth_coeff=0.0, search_dist=4, search_ang=15.0, len_max=50)
    r = orc.stream(ov, sub, f=f, mask=mask, seed=seed_vol, nthreads=2, **kw)
    save(name, ovec=ov, f=f, mask=mask, seed=seed_vol, sublist=sub, npts=r["npts"], seed_index=r["seed_index"], xyz=r["xyz"],
         **{"kw_" + k: v for k, v in kw.items()})


def lcm_case(name, n, seed):
    """LCM-guided tracking (stream.jl:380-495) on 2-D in-plane data, uniforms from the ABI's counter-based stream"""
    rng = np.random.default_rng(seed)
    ovs = []
    for k in range(2):
        a = rng.uniform(-0.5, 0.5, (n, n, 1)) + k * np.pi / 2
        ov = np.zeros((n, n, 1, 3), np.float32, order="F")
        ov[..., 0], ov[..., 1] = np.cos(a), np.sin(a)
        ovs.append(ov)
    mask = (rng.random((n, n, 1)) < 0.95).astype(np.uint8)
    lcms = np.asfortranarray(rng.random((n, n, 1, 10)).astype(np.float32))
    lcms[rng.random((n, n, 1)) < 0.05] = 0.0
    sub = np.array([[0.1, -0.2, 0.0], [0.3, 0.25, 0.0]], np.float32)
    r = orc.stream(ovs, sub, mask=mask, lcms=lcms, lcm_thresh=0.15, rng_seed=20251003, len_max=40, nthreads=2)
    save(name, ovec=np.stack(ovs), mask=mask, lcms=lcms, sublist=sub, lcm_thresh=0.15, rng_seed=20251003, len_max=40,
         npts=r["npts"], seed_index=r["seed_index"], xyz=r["xyz"], flags=r["flags"],
         uniforms=np.array([[orc.lib().orc_uniform(__import__("ctypes").c_uint64(20251003), __import__("ctypes").c_uint64(line),
                                                   __import__("ctypes").c_uint32(k)) for k in range(4)] for line in range(4)], np.float32))


def rumba_case(name, shape, seed, niter):
    bval, bvec = phantom.scheme_gqi(3, 30, (1000.0, 2500.0), seed)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed, noise_frac=0.03, crossing=True)
    mask = (np.random.default_rng(seed + 1).random(shape) < 0.85).astype(np.uint8)
    r = orc.rumba_rec(dwi, mask, bval, bvec, fj.sphere_724.vertices, niter=niter)
    save(name, dwi=dwi, mask=mask, bval=bval, bvec=bvec, niter=niter, fodf=r["fodf"], fgm=r["fgm"], fcsf=r["fcsf"], gfa=r["gfa"],
         var=r["var"], peak=np.stack(r["peak"]), snr_mean=r["snr_mean"], snr_std=r["snr_std"])


if __name__ == "__main__":
    micro_case("stream_micro_14", 14, seed=21)
    lcm_case("stream_lcm_20", 20, seed=22)
    rumba_case("rumba_5x4x4x63_sphere724", (5, 4, 4), seed=23, niter=20)
    dti_case("dti_8x8x8x7", (8, 8, 8), 6, 1, seed=1, nonpos=0.0)
    dti_case("dti_6x5x4x33_nonpositive", (6, 5, 4), 30, 3, seed=7, nonpos=0.03)
    gqi_case("gqi_6x6x6x63_sphere642", (6, 6, 6), "sphere_642", seed=3)
    gqi_case("gqi_5x4x3x63_sphere362", (5, 4, 3), "sphere_362", seed=4)
    dsi_case("dsi_3x2x2x515", (3, 2, 2), seed=5)
    peaks_case("find_peaks_ties_sphere642")
    stream_case("stream_12", 12, seed=6)
