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


def energy_joules(dev=0):
    """Accumulated board energy in Joules (monotonic), or None"""
    lib = _smi()
    if lib is None:
        return None
    c, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
    if lib.rsmi_dev_energy_count_get(dev, C.byref(c), C.byref(res), C.byref(ts)) != 0:
        return None
    return c.value * float(res.value) * 1e-6


def power_watts(dev=0):
    lib = _smi()
    if lib is None:
        return None
    p, ty = C.c_uint64(0), C.c_int(0)
    if lib.rsmi_dev_power_get(dev, C.byref(p), C.byref(ty)) != 0:
        return None
    return p.value * 1e-6


class _Freq(C.Structure):
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


def sclk_mhz(dev=0):
    """The shader clock the SMU reports right now (sysfs pp_dpm_sclk's starred entry), or None.  NOT the in-kernel clock: see
    MI355X_MICROARCH.md 'DVFS give-back' (6) -- the contraction kernels' own clock comes from the diagnostic build's stamps."""
    lib = _smi()
    if lib is None:
        return None
    f = _Freq()
    try:
        if lib.rsmi_dev_gpu_clk_freq_get(dev, 0, C.byref(f)) != 0 or f.current >= 33:
            return None
    except Exception:                                                # noqa: BLE001
        return None
    return f.frequency[f.current] * 1e-6


def measure(step, sync, seconds=2.0, batch=8, dev=0, sample_every=0.25):
    """Runs `step()` back to back for about `seconds` (after 0.5 s untimed), `sync()` after every `batch` calls.  Returns a dict:
    steps, seconds, joules, watts, joules_per_step, ms_per_step, sclk_mhz_mean (SMU samples while running) -- or None without a counter."""
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


def idle_watts(seconds=1.5, dev=0):
    if energy_joules(dev) is None:
        return None
    time.sleep(0.3)
    e0, t0 = energy_joules(dev), time.perf_counter()
    time.sleep(seconds)
    e1, t1 = energy_joules(dev), time.perf_counter()
    return (e1 - e0) / (t1 - t0)
