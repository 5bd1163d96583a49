#!/usr/bin/env python3
"""The host-buffer (drop-in) entry points at benchmark size, PCIe included -- what a Julia caller of fib_* pays (SURVEY 8d "report
both") -- with the host tier's stages timed apart: gather (caller's rows -> pinned ring), H2D, kernels, D2H, scatter (ring -> caller's
rows), per-direction GB/s, and the fraction of the PCIe roof (max(bytes in, bytes out) / 63 GB/s: Gen5 x16, one direction).

usage: host_tier_probe.py [--legs gqi,dti,dsi,stream] [--bind gpu|other|none] [--reps 5] [--json out.json]
  --bind  where THIS process (= the caller: its arrays are first touched here) runs: on the CPUs of the GPU's NUMA node, on the other
          node's, or wherever the scheduler puts it.  The library binds its own copy threads and pinned ring to the GPU's node itself.
bench.py imports the leg functions for its extra.host_tier entries."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
SHAPE = (140, 140, 140)
PCIE_GBS = 63.0                      # PCIe Gen5 x16, one direction
STAGES = ("host_gather", "host_gather_wait", "host_h2d", "host_d2h", "host_scatter", "host_scatter_wait", "odf_gemm", "dti_fit",
          "stream_trace", "stream_pack", "stream_host_masks_first_pass", "stream_host_seed_list", "stream_host_upload_field", "stream_host_seeds_trace",
          "stream_host_results")


def gpu_numa_cpus(device=0):
    """CPUs of the NUMA node the GPU hangs off (sysfs), or None"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        txt = open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip()
        cpus = []
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        return cpus
    except Exception:                                                    # noqa: BLE001
        return None


def bind(where):
    """binds this process; returns a description"""
    if where == "none":
        return "not bound"
    local = gpu_numa_cpus()
    if not local:
        return "not bound (no sysfs NUMA information)"
    allowed = sorted(os.sched_getaffinity(0))
    want = [c for c in allowed if (c in local) == (where == "gpu")]
    if not want:
        return "not bound (no allowed CPU on that node)"
    os.sched_setaffinity(0, want)
    return "bound to %d CPUs %s the GPU's NUMA node" % (len(want), "on" if where == "gpu" else "OFF")


def _stages(L):
    out = {}
    for name in STAGES:
        ms, n = C.c_double(0), C.c_int64(0)
        L.fib_profile_get(name.encode(), C.byref(ms), C.byref(n))
        if n.value:
            out[name] = dict(ms=ms.value, count=n.value)
    return out


def _timed(L, call, reps, bytes_in, bytes_out, exclude=None):
    """exclude(): seconds of the last call that are not the entry point's (e.g. releasing its result), subtracted"""
    from fibers_jl_amd import _lib
    _lib.check(call())                                                   # warm-up: plan, ring, streams
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _lib.check(call())
        ts.append(time.perf_counter() - t0 - (exclude() if exclude else 0.0))
    L.fib_profile_enable(1)
    L.fib_profile_reset()
    t0 = time.perf_counter()
    _lib.check(call())                                                   # one more, with the stages timed (events around the copies)
    t_prof = time.perf_counter() - t0 - (exclude() if exclude else 0.0)
    st = _stages(L)
    L.fib_profile_enable(0)
    best, med = min(ts), float(np.median(ts))
    floor = max(bytes_in, bytes_out) / (PCIE_GBS * 1e9)
    r = dict(e2e_pcie_ms=best * 1e3, e2e_pcie_ms_median=med * 1e3, all_ms=[t * 1e3 for t in ts], bytes_in=bytes_in, bytes_out=bytes_out,
             link_gbs=(bytes_in + bytes_out) / best / 1e9, pcie_floor_ms=floor * 1e3, pcie_roof_frac=floor / best,
             stages_of_one_profiled_call=dict(e2e_ms=t_prof * 1e3, **{k: v for k, v in st.items()}))
    if "host_h2d" in st and st["host_h2d"]["ms"] > 0:
        r["h2d_gbs_while_copying"] = bytes_in / (st["host_h2d"]["ms"] * 1e-3) / 1e9
    if "host_d2h" in st and st["host_d2h"]["ms"] > 0:
        r["d2h_gbs_while_copying"] = bytes_out / (st["host_d2h"]["ms"] * 1e-3) / 1e9
    return r


def _host_dwi(shape, bval, bvec, seed, dev, **kw):
    import torch
    from fibers_jl_amd import phantom
    d, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=seed, device=dev, **kw)
    h = np.ascontiguousarray(d.cpu().numpy())                           # [nvol, nvox] planar == MRI.vol memory; first touched by THIS thread
    del d                                                                # (back to torch's cache, NOT to the driver: releasing GBs of device memory
    return h                                                             #  slows the next seconds' downloads -- tools/host_tier_state_check.py)


def leg_odf(kind, shape=SHAPE, reps=4, mask=None, dev=None):
    """fib_gqi_rec / fib_dsi_rec on pageable host arrays, outputs touched before the call"""
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import _lib, phantom
    dev = dev or torch.device("cuda", 0)
    L = fj.lib()
    nx, ny, nz = shape
    nvox = nx * ny * nz
    sph = fj.sphere_642
    bval, bvec = phantom.scheme_gqi() if kind == "gqi" else phantom.scheme_dsi()
    host = _host_dwi(shape, bval, bvec, 3 if kind == "gqi" else 5, dev)
    nvol, nvert = len(bval), sph.nvert
    m8 = np.ones(nvox, np.uint8) if mask is None else np.ascontiguousarray(mask, np.uint8)
    v = np.asfortranarray(sph.vertices, np.float32); f = np.asfortranarray(sph.faces, np.int32)
    bv = np.ascontiguousarray(bval, np.float32); bg = np.asfortranarray(np.asarray(bvec, np.float32))
    odf_h = np.ones((nvert, nvox), np.float32)
    pdf_h = np.ones((nvol, nvox), np.float32) if kind == "dsi" else None
    pk_h = [np.ones((3, nvox), np.float32) for _ in range(3)]
    qa_h = [np.ones(nvox, np.float32) for _ in range(3)]
    if kind == "gqi":
        def call():
            return L.fib_gqi_rec(0, host.ctypes.data, nx, ny, nz, nvol, m8.ctypes.data, 0, bv.ctypes.data, bg.ctypes.data, v.ctypes.data, v.shape[0],
                                 f.ctypes.data, f.shape[0], 1.25, odf_h.ctypes.data, _lib.P3(*[a.ctypes.data for a in pk_h]), _lib.P3(*[a.ctypes.data for a in qa_h]))
    else:
        def call():
            return L.fib_dsi_rec(0, host.ctypes.data, nx, ny, nz, nvol, m8.ctypes.data, 0, bv.ctypes.data, bg.ctypes.data, v.ctypes.data, v.shape[0],
                                 f.ctypes.data, f.shape[0], 32, pdf_h.ctypes.data, odf_h.ctypes.data, _lib.P3(*[a.ctypes.data for a in pk_h]),
                                 _lib.P3(*[a.ctypes.data for a in qa_h]))
    nin = float(m8.sum()) / nvox if mask is not None else 1.0
    bytes_in = host.nbytes * nin + nvox
    bytes_out = (odf_h.nbytes + (pdf_h.nbytes if pdf_h is not None else 0) + sum(a.nbytes for a in pk_h + qa_h)) * nin
    r = _timed(L, call, reps, bytes_in, bytes_out)
    r.update(voxels=nvox, mvoxels_per_s=nvox / (r["e2e_pcie_ms"] * 1e-3) / 1e6,
             note="fib_%s_rec on pageable host arrays (the call a Julia wrapper makes), outputs touched before the call: gather -> pinned ring -> H2D || "
                  "kernels || D2H -> scatter, the two host stages side by side" % kind)
    # the same call into FRESHLY zero-allocated outputs, as a Julia caller makes them (`zeros`, mri.jl:251-255)
    tf = []
    for _ in range(2):
        odf_h = np.zeros((nvert, nvox), np.float32)
        if pdf_h is not None:
            pdf_h = np.zeros((nvol, nvox), np.float32)
        pk_h = [np.zeros((3, nvox), np.float32) for _ in range(3)]
        qa_h = [np.zeros(nvox, np.float32) for _ in range(3)]
        t0 = time.perf_counter()
        _lib.check(call())
        tf.append(time.perf_counter() - t0)
    r["e2e_pcie_first_touch_ms"] = min(tf) * 1e3
    if mask is not None:
        # .. and with FIB_MASK_OUTPUTS_ZEROED, which such a caller can pass (the wrappers do): the voxels outside the mask are not written
        zflag = _lib.FIB_MASK_OUTPUTS_ZEROED
        tz = []
        for _ in range(3):
            odf_h = np.zeros((nvert, nvox), np.float32)
            if pdf_h is not None:
                pdf_h = np.zeros((nvol, nvox), np.float32)
            pk_h = [np.zeros((3, nvox), np.float32) for _ in range(3)]
            qa_h = [np.zeros(nvox, np.float32) for _ in range(3)]
            t0 = time.perf_counter()
            if kind == "gqi":
                rc = L.fib_gqi_rec(0, host.ctypes.data, nx, ny, nz, nvol, m8.ctypes.data, 0 | zflag, bv.ctypes.data, bg.ctypes.data, v.ctypes.data, v.shape[0],
                                   f.ctypes.data, f.shape[0], 1.25, odf_h.ctypes.data, _lib.P3(*[a.ctypes.data for a in pk_h]), _lib.P3(*[a.ctypes.data for a in qa_h]))
            else:
                rc = L.fib_dsi_rec(0, host.ctypes.data, nx, ny, nz, nvol, m8.ctypes.data, 0 | zflag, bv.ctypes.data, bg.ctypes.data, v.ctypes.data, v.shape[0],
                                   f.ctypes.data, f.shape[0], 32, pdf_h.ctypes.data, odf_h.ctypes.data, _lib.P3(*[a.ctypes.data for a in pk_h]),
                                   _lib.P3(*[a.ctypes.data for a in qa_h]))
            _lib.check(rc)
            tz.append(time.perf_counter() - t0)
        r["e2e_pcie_zeroed_outputs_flag_ms"] = min(tz) * 1e3
    return r


def leg_dti(shape=SHAPE, reps=4, dev=None):
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import _lib, phantom
    from fibers_jl_amd.dti import DTI_FIELDS
    dev = dev or torch.device("cuda", 0)
    L = fj.lib()
    nx, ny, nz = shape
    nvox = nx * ny * nz
    bval, bvec = phantom.scheme_dti(60, 4, 1000.0, seed=2)
    host = _host_dwi(shape, bval, bvec, 2, dev, nfib=1)
    m8 = np.ones(nvox, np.uint8)
    bv = np.ascontiguousarray(bval, np.float32); bg = np.asfortranarray(np.asarray(bvec, np.float32))
    outs = {k: np.ones((3 if "vec" in k else 1, nvox), np.float32) for k in DTI_FIELDS}
    o = _lib.DtiOut(*[outs[k].ctypes.data for k in DTI_FIELDS])

    def call():
        return L.fib_dti_fit(0, host.ctypes.data, nx, ny, nz, len(bval), m8.ctypes.data, 0, bv.ctypes.data, bg.ctypes.data, C.byref(o))
    r = _timed(L, call, reps, host.nbytes + nvox, sum(a.nbytes for a in outs.values()))
    r.update(voxels=nvox, mvoxels_per_s=nvox / (r["e2e_pcie_ms"] * 1e-3) / 1e6, note="fib_dti_fit on pageable host arrays, 64 frames in, 16 floats per voxel out")
    return r


def leg_stream(shape=SHAPE, reps=3, dev=None):
    """fib_stream (C4: DTI principal eigenvector, ball mask, one offset): field + mask in, ~1.5 GB of points out (D2H through the pinned
    ring into freshly malloc'ed arrays the caller frees with fib_tract_free)"""
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import _lib, phantom
    dev = dev or torch.device("cuda", 0)
    L = fj.lib()
    nx, ny, nz = shape
    nvox = nx * ny * nz
    ov = np.ascontiguousarray(np.asfortranarray(phantom.fibre_field(nx, ny, nz).astype(np.float32)).reshape(nvox, 3, order="F").T)   # [3, nvox] planar
    mask = np.ascontiguousarray(phantom.ball_mask(nx, ny, nz).reshape(-1, order="F").astype(np.uint8))
    sub = np.array([[0.1, -0.2, 0.3]], np.float32)
    smod = sys.modules[fj.stream_device_run.__module__]                    # (fj.stream is the function; its module holds _params)
    prm = smod._params((nx, ny, nz), 1, 3, max(shape), 45, 0.5, 0.2, 0, 10, None)
    pv = (C.c_void_p * 1)(ov.ctypes.data)
    res = {}

    def call():
        out = _lib.TractOut()
        rc = L.fib_stream(0, C.byref(prm), pv, None, C.c_float(0.03), None, C.c_float(0.1), mask.ctypes.data, 0, None, 0, sub.ctypes.data, 1, C.byref(out))
        res["lines"], res["points"] = int(out.nlines), int(out.npoints)
        res["pending"] = out                                              # (released OUTSIDE the timed call: free_pending)
        return rc

    def free_pending():
        t0 = time.perf_counter()
        if res.get("pending") is not None:
            L.fib_tract_free(C.byref(res["pending"]))
            res["pending"] = None
        return time.perf_counter() - t0
    _lib.check(call())
    free_pending()
    bytes_out = res["points"] * 12.0 + res["lines"] * 12.0
    frees = []

    def call_and_free():
        rc = call()
        res["t_free"] = free_pending()                                    # fib_tract_free: munmap of 1.5 GB, timed apart
        frees.append(res["t_free"])
        return rc
    r = _timed(L, call_and_free, reps, ov.nbytes + nvox, bytes_out, exclude=lambda: res.get("t_free", 0.0))
    r["tract_free_ms"] = float(np.median(frees)) * 1e3
    r.update(lines=res["lines"], points=res["points"], mpoints_per_s=res["points"] / (r["e2e_pcie_ms"] * 1e-3) / 1e6,
             note="fib_stream on host arrays: orientation field + mask up, field kernel, seeds, trace + scan + pack, then the packed points, counts and "
                  "seed indices down into arrays the library allocates (fib_tract_free)")
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", default="gqi,dti,dsi,stream")
    ap.add_argument("--bind", default="none", choices=["gpu", "other", "none"])
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import torch                                                         # (before binding: the sysfs lookup needs the device's PCI address)
    torch.cuda.init()
    how = bind(args.bind)
    doc = dict(caller=how, legs={})
    for leg in args.legs.split(","):
        if leg in ("gqi", "dsi"):
            r = leg_odf(leg, reps=args.reps)
        elif leg == "dti":
            r = leg_dti(reps=args.reps)
        elif leg == "stream":
            r = leg_stream(reps=max(2, args.reps - 1))
        else:
            continue
        doc["legs"][leg] = r
        st = r["stages_of_one_profiled_call"]
        print("%-7s [%s] e2e %.1f ms (median %.1f) = %.0f%% of the PCIe roof (%.1f ms); link %.1f GB/s; stages: %s" % (
            leg, how, r["e2e_pcie_ms"], r["e2e_pcie_ms_median"], 100 * r["pcie_roof_frac"], r["pcie_floor_ms"], r["link_gbs"],
            ", ".join("%s %.1f" % (k.replace("host_", ""), v["ms"]) for k, v in st.items() if isinstance(v, dict))), flush=True)
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        json.dump(doc, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
