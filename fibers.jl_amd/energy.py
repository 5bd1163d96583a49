"""Board energy / power / clock readings for measurement code (bench.py's roofline.power, tools/energy_model.py): the accumulated
energy counter of the GPU through librocm_smi64 (rsmi_dev_energy_count_get: counts x resolution in micro-Joules), read with
ctypes in this process -- no child process, no exec.  Not part of the hot path: nothing under csrc/ depends on it and every
reader degrades to None when the library or the counter is missing."""
import ctypes as C
import time

_lib = None
_tried = False


def _smi():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    for name in ("librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so.1"):
        try:
            lib = C.CDLL(name)
        except OSError:
            continue
        try:
            if lib.rsmi_init(C.c_uint64(0)) != 0:
                continue
        except Exception:                                            # noqa: BLE001
            continue
        lib.rsmi_dev_energy_count_get.argtypes = [C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        lib.rsmi_dev_power_get.argtypes = [C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        _lib = lib
        break
    return _lib


def rsmi_index_of(hip_device=0):
    """The rocm_smi index of a HIP / torch device.  rsmi enumerates every board and ignores HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES,
    so `cuda:0` need not be rsmi's board 0: match the PCI bus address (torch's device properties against rsmi_dev_pci_id_get, whose id
    is (domain << 32) | (bus << 8) | (device << 3) | function).  Raises if no board matches: a reading from another board must never be
    reported as this run's."""
    lib = _smi()
    if lib is None:
        return 0                                                     # (no library: every reader returns None anyway)
    import torch
    p = torch.cuda.get_device_properties(hip_device)
    want = (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
    n = C.c_uint32(0)
    if lib.rsmi_num_monitor_devices(C.byref(n)) != 0:
        raise RuntimeError("rsmi_num_monitor_devices failed")
    seen = []
    for i in range(n.value):
        bdf = C.c_uint64(0)
        if lib.rsmi_dev_pci_id_get(C.c_uint32(i), C.byref(bdf)) != 0:
            continue
        v = bdf.value
        got = ((v >> 32) & 0xFFFFFFFF, (v >> 8) & 0xFF, (v >> 3) & 0x1F)
        seen.append(got)
        if got == want:
            return i
    raise RuntimeError("no rocm_smi board at PCI %04x:%02x:%02x (boards seen: %s)" % (want + (seen,)))


def _resolve(dev):
    """dev None: the rocm_smi board of torch's current device (rsmi_index_of); an int: that rsmi index"""
    if dev is not None:
        return dev
    try:
        import torch
        return rsmi_index_of(torch.cuda.current_device()) if torch.cuda.is_available() else 0
    except ImportError:
        return 0


def energy_joules(dev=None):
    """Accumulated board energy in Joules (monotonic), or None"""
    lib = _smi()
    if lib is None:
        return None
    dev = _resolve(dev)
    c, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
    if lib.rsmi_dev_energy_count_get(dev, C.byref(c), C.byref(res), C.byref(ts)) != 0:
        return None
    return c.value * float(res.value) * 1e-6


def power_watts(dev=None):
    lib = _smi()
    if lib is None:
        return None
    dev = _resolve(dev)
    p, ty = C.c_uint64(0), C.c_int(0)
    if lib.rsmi_dev_power_get(dev, C.byref(p), C.byref(ty)) != 0:
        return None
    return p.value * 1e-6


class _Freq(C.Structure):
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


def sclk_mhz(dev=None):
    """The shader clock the SMU reports right now (sysfs pp_dpm_sclk's starred entry), or None.  NOT the in-kernel clock: see
    MI355X_MICROARCH.md 'DVFS give-back' (6) -- the contraction kernels' own clock comes from the diagnostic build's stamps."""
    lib = _smi()
    if lib is None:
        return None
    dev = _resolve(dev)
    f = _Freq()
    try:
        if lib.rsmi_dev_gpu_clk_freq_get(dev, 0, C.byref(f)) != 0 or f.current >= 33:
            return None
    except Exception:                                                # noqa: BLE001
        return None
    return f.frequency[f.current] * 1e-6


def measure(step, sync, seconds=2.0, batch=8, dev=None, sample_every=0.25):
    """Runs `step()` back to back for about `seconds` (after 0.5 s untimed), `sync()` after every `batch` calls.  Returns a dict:
    steps, seconds, joules, watts, joules_per_step, ms_per_step, sclk_mhz_mean (SMU samples while running) -- or None without a counter.
    dev: a rocm_smi index; None = the board of torch's current device (rsmi ignores HIP_VISIBLE_DEVICES: rsmi_index_of matches PCI addresses)."""
    dev = _resolve(dev)
    if energy_joules(dev) is None:
        return None
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(batch):
            step()
        sync()
    n, sc, ns = 0, 0.0, 0
    e0, t0 = energy_joules(dev), time.perf_counter()
    nxt = t0 + sample_every
    while time.perf_counter() - t0 < seconds:
        for _ in range(batch):
            step()
        sync()
        n += batch
        if time.perf_counter() >= nxt:
            s = sclk_mhz(dev)
            if s:
                sc += s
                ns += 1
            nxt += sample_every
    t1, e1 = time.perf_counter(), energy_joules(dev)
    return dict(steps=n, seconds=t1 - t0, joules=e1 - e0, watts=(e1 - e0) / (t1 - t0), joules_per_step=(e1 - e0) / n,
                ms_per_step=(t1 - t0) / n * 1e3, sclk_mhz_mean=sc / ns if ns else None)


def idle_watts(seconds=1.5, dev=None):
    dev = _resolve(dev)
    if energy_joules(dev) is None:
        return None
    time.sleep(0.3)
    e0, t0 = energy_joules(dev), time.perf_counter()
    time.sleep(seconds)
    e1, t1 = energy_joules(dev), time.perf_counter()
    return (e1 - e0) / (t1 - t0)


# ---- the GQI step's energy by component (bench.py roofline.power, tools/energy_model.py) -------------------------------------------------
def gqi_counts(nvox, nvol=270, nvert=321, valu_wave_instructions=None):
    """what one fused GQI step (fp16 pieces) does, per launch.  VALU: SQ_INSTS_VALU of the launch minus its MFMAs at 140^3
    (profiles/r04/summary.txt: 2.81e8 - 0.437e8), scaled by the voxel count"""
    items = -(-nvox // 256)
    nst = -(-nvol // 16)
    if valu_wave_instructions is None:
        valu_wave_instructions = 2.3727e8 * nvox / 2744000.0
    return dict(
        hbm_bytes=(4.0 * nvol + 1 + 4.0 * nvert + 48) * nvox,                 # algorithmic: DWI + mask in, ODF + peaks + qa out
        mfma_flops=3 * 2.0 * 320 * (16 * nst) * nvox,                          # executed: three fp16 piece products, K padded to whole stages
        lds_fragment_bytes=20.0 * 1024 * 8 * nst * items,                      # every wave re-reads the stage's 20 KiB of matrix pieces
        l2_to_lds_bytes=(20.0 * 1024 * nst) * items + 4.0 * 16 * nst * nvox,   # the pieces per workgroup and item + the samples, by LDS-DMA
        lds_other_bytes=(2 * 4.0 * 16 * nst + 2 * 4.0 * 320) * nvox,           # sample tiles read back + the epilogue's transposition (write + read)
        valu_wave_instructions=valu_wave_instructions)


UNAVOIDABLE = ("hbm_bytes", "mfma_flops")       # the algorithm's bytes and its (three-product) matrix-core work; the rest is this kernel's way of doing it


def gqi_power_roofline(joules_per_unit, nvox, kernel_ms, step_ms, step_joules, idle_w, cap_w=1400.0, **count_kw):
    """roofline.power: joules_by_component = counts x the probes' Joules per unit (above the idle board), and the floor a kernel made of
    nothing but the unavoidable components (the algorithm's HBM bytes and its executed MFMA flops) would reach under the cap:
    floor_ms / frac          (the figures to quote) every component scaled by ONE factor so that components + idle = the step's MEASURED
                             Joules -- the probes for HBM, LDS and the vector ALU ran at 2.4 GHz and its voltage, the kernel runs at 1.8-1.9 GHz
                             where every operation costs less -- then unavoidable Joules / (cap - idle).  Conservative: a shorter kernel would
                             run at a higher clock and pay more per operation than this assumes;
    floor_raw_ms / frac_raw  the same without the scaling: each ingredient at what it costs ALONE at its own clock.  An over-count (the model
                             then exceeds the measured Joules by ~25 %), so this 'floor' can exceed the kernel's time: an upper estimate."""
    counts = gqi_counts(nvox, **count_kw)
    joules = {k: (counts[k] * joules_per_unit[k] if joules_per_unit.get(k) is not None else None) for k in counts}
    dyn = sum(v for v in joules.values() if v)
    unavoidable = sum(joules[k] or 0.0 for k in UNAVOIDABLE)
    budget = cap_w - idle_w
    idle_j = idle_w * step_ms * 1e-3
    scale = max(0.0, step_joules - idle_j) / dyn if (step_joules and dyn > 0) else None
    out = dict(cap_w=cap_w, idle_w=idle_w, budget_w=budget, counts_per_step=counts, joules_per_unit=dict(joules_per_unit),
               joules_by_component_raw=joules, idle_joules_per_step=idle_j, modelled_joules_per_step_raw=dyn + idle_j,
               measured_joules_per_step=step_joules, model_over_measured=(dyn + idle_j) / step_joules if step_joules else None,
               unavoidable_components=list(UNAVOIDABLE), kernel_ms=kernel_ms, step_ms=step_ms,
               floor_raw_ms=unavoidable / budget * 1e3, frac_raw=(unavoidable / budget * 1e3) / kernel_ms if kernel_ms else None)
    if scale is not None:
        out["calibration_scale"] = scale
        out["joules_by_component"] = {k: (v * scale if v else v) for k, v in joules.items()}
        out["floor_ms"] = unavoidable * scale / budget * 1e3
        out["frac"] = out["floor_ms"] / kernel_ms if kernel_ms else None
    return out
