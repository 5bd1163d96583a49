"""ctypes binding of libfibers_hip.so (include/fibers_hip.h).  There is no CPU fallback: a
missing library or a missing GPU is an error, never a silent detour."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FIBERS_HIP_LIB") or os.path.join(_HERE, "libfibers_hip.so")   # override: A/B builds of the same ABI

FIB_OK = 0
FIB_ERR_CAPACITY = -9
FIB_MASK_OUTPUTS_ZEROED = 0x100     # OR-ed into mask_dtype: the output arrays are freshly zero-allocated (include/fibers_hip.h)
DTYPES = {"uint8": 0, "int8": 1, "int16": 2, "uint16": 3, "int32": 4, "uint32": 5,
          "float32": 6, "float64": 7, "int64": 8, "bool": 9}


class FibersError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfibers_hip error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


class DtiOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("s0", "eigval1", "eigval2", "eigval3", "eigvec1", "eigvec2", "eigvec3", "rd", "md", "fa")]


class StreamParams(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("nvec", C.c_int32),
                ("len_min", C.c_int32), ("len_max", C.c_int32),
                ("cosang_thresh", C.c_float), ("step_size", C.c_float), ("smooth_coeff", C.c_float),
                ("search_dist", C.c_int32), ("search_cosang", C.c_float), ("ws", C.c_void_p), ("interp", C.c_int32),
                ("search_flat_axis", C.c_int32)]


class RumbaOut(C.Structure):
    _fields_ = [("fodf", C.c_void_p), ("fgm", C.c_void_p), ("fcsf", C.c_void_p), ("gfa", C.c_void_p), ("var", C.c_void_p),
                ("peak", C.c_void_p * 5)]


class TractOut(C.Structure):
    _fields_ = [("nlines", C.c_int64), ("npoints", C.c_int64),
                ("npts", C.POINTER(C.c_int32)), ("seed_index", C.POINTER(C.c_int64)),
                ("xyz", C.POINTER(C.c_float)), ("flags", C.POINTER(C.c_uint8))]


_lib = None
vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
P3 = C.c_void_p * 3

_PROTOS = {
    "fib_last_error": (C.c_char_p, []),
    "fib_version": (C.c_char_p, []),
    "fib_device_count": (i32, []),
    "fib_profile_enable": (i32, [i32]),
    "fib_profile_filter": (i32, [C.c_char_p]),
    "fib_profile_reset": (i32, []),
    "fib_profile_get": (i32, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]),
    "fib_dti_plan_create": (i32, [i32, vp, vp, i32, C.POINTER(vp)]),
    "fib_dti_plan_destroy": (None, [vp]),
    "fib_dti_plan_tables": (i32, [vp, vp, vp, C.POINTER(i32)]),
    "fib_gqi_plan_create": (i32, [i32, vp, vp, i32, vp, i32, vp, i32, f32, C.POINTER(vp)]),
    "fib_dsi_plan_create": (i32, [i32, vp, vp, i32, vp, i32, vp, i32, i32, C.POINTER(vp)]),
    "fib_gqi_plan_create_fmt": (i32, [i32, vp, vp, i32, vp, i32, vp, i32, f32, i32, C.POINTER(vp)]),
    "fib_dsi_plan_create_fmt": (i32, [i32, vp, vp, i32, vp, i32, vp, i32, i32, i32, C.POINTER(vp)]),
    "fib_odf_plan_format": (i32, [vp]),
    "fib_odf_plan_list_unit": (i32, [vp, vp]),
    "fib_odf_default_format": (i32, []),
    "fib_odf_plan_destroy": (None, [vp]),
    "fib_odf_plan_matrix": (i32, [vp, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "fibd_dti_fit": (i32, [vp, vp, vp, i64, C.POINTER(DtiOut), vp]),
    "fibd_adc_fit": (i32, [vp, vp, vp, i64, vp, vp, vp]),
    "fibd_st_eigen": (i32, [vp, i64, vp, vp, vp]),
    "fib_st_eigen": (i32, [i32, vp, i64, vp, vp]),
    "fibd_dti_last_partial_count": (i32, [vp, vp, C.POINTER(i64)]),
    "fibd_odf_rec": (i32, [vp, vp, vp, i64, vp, vp, P3, P3, vp, i32, vp]),
    "fibd_qa_normalize": (i32, [P3, i64, f32, vp]),
    "fibd_qa_normalize_dev": (i32, [P3, i64, vp, vp]),
    "fibd_qa_normalize_pair": (i32, [P3, i64, vp, vp]),
    "fibd_find_peaks": (i32, [vp, vp, i64, vp, vp, vp]),
    "fibd_stream_field": (i32, [i32, i64, vp, vp, f32, vp, f32, vp, vp, vp, vp]),
    "fibd_stream_trace": (i32, [C.POINTER(StreamParams), vp, vp, i64, vp, i32, vp,
                                C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]),
    "fibd_stream_run": (i32, [C.POINTER(StreamParams), vp, vp, i64, vp, i32, vp, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64), vp]),
    "fibd_stream_run_enqueue": (i32, [C.POINTER(StreamParams), vp, vp, i64, vp, i32, vp, vp, i64, vp, i64, vp, vp]),
    "fibd_stream_ws_create": (i32, [i32, C.POINTER(vp)]),
    "fibd_stream_ws_destroy": (None, [vp]),
    "fibd_stream_pack": (i32, [vp, vp, vp, vp, vp]),
    "fibd_stream_pack_trk": (i32, [vp, C.POINTER(C.c_float * 3), vp, vp]),
    "fibd_stream_trace_lcm": (i32, [C.POINTER(StreamParams), vp, vp, f32, i32, i32, C.c_uint64, vp, i64, vp, i32, vp,
                                    C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]),
    "fibd_stream_pack_flags": (i32, [vp, vp, vp, vp, vp, vp]),
    "fibd_stream_all_npts": (i32, [vp, vp, vp]),
    "fib_stream_job_destroy": (None, [vp]),
    "fib_init": (i32, [i32, vp]),
    "fib_trim": (i32, []),
    "fib_shutdown": (None, []),
    "fib_dti_fit": (i32, [i32, vp, i32, i32, i32, i32, vp, i32, vp, vp, C.POINTER(DtiOut)]),
    "fib_adc_fit": (i32, [i32, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp]),
    "fib_gqi_rec": (i32, [i32, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp, i32, f32, vp, P3, P3]),
    "fib_dsi_rec": (i32, [i32, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, vp, vp, P3, P3]),
    "fib_rumba_plan_create": (i32, [i32, vp, vp, i32, vp, i32, f32, f32, f32, f32, C.POINTER(vp)]),
    "fib_rumba_plan_destroy": (None, [vp]),
    "fib_rumba_plan_kernel": (i32, [vp, vp, C.POINTER(i32), C.POINTER(i32)]),
    "fibd_rumba_rec": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(RumbaOut), C.POINTER(f32),
                             C.POINTER(f32), vp]),
    "fib_rumba_rec": (i32, [i32, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, i32, f32, f32, f32, f32, i32, i32, i32,
                            i32, C.POINTER(RumbaOut), C.POINTER(f32), C.POINTER(f32)]),
    "fib_find_peaks": (i32, [i32, vp, i64, vp, i32, vp, i32, vp, vp]),
    "fib_find_peaks_work": (i32, [i32, vp, i64, vp, i32, vp, i32, vp, vp, vp]),
    "fibd_find_peaks_work": (i32, [vp, vp, i64, vp, vp, vp, vp]),
    "fib_stream": (i32, [i32, C.POINTER(StreamParams), vp, vp, f32, vp, f32, vp, i32, vp, i32, vp, i32,
                         C.POINTER(TractOut)]),
    "fib_stream_lcm": (i32, [i32, C.POINTER(StreamParams), vp, vp, f32, vp, f32, vp, i32, vp, i32, vp, i32,
                             vp, f32, C.c_uint64, C.POINTER(TractOut)]),
    "fib_tract_free": (None, [C.POINTER(TractOut)]),
}


def exported_symbols():
    """names declared in include/fibers_hip.h (kept in sync by tests/test_abi.py)"""
    return sorted(_PROTOS)


def lib():
    """Load libfibers_hip.so; raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FibersError(-2, "%s not found: build it with `make -C %s/csrc` (no CPU fallback exists)"
                              % (LIB_PATH, _HERE))
        # torch bundles its own HIP runtime (libamdhip64.so.7); two runtimes in one process cannot both
        # own the GPU.  Importing torch first makes the loader resolve our DT_NEEDED to that same copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:         # calling it later raises; tests/test_abi.py requires all
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


DEVICE_ALL = -1     # FIB_DEVICE_ALL: the device set declared with init()


def init(devices=None):
    """fib_init: the device set that `device=DEVICE_ALL` shards over (None / empty: every visible device; an index may
    repeat: two pipelines on one GPU)."""
    devs = [] if devices is None else [int(d) for d in devices]
    arr = (C.c_int * max(1, len(devs)))(*devs)
    check(lib().fib_init(len(devs), arr))


def shutdown():
    lib().fib_shutdown()


def trim():
    """fib_trim: the host tier's buffers kept between calls (pinned ring, its device mirror, fib_stream's device buffers, the tracer's
    workspace) go back to the driver; plans stay"""
    check(lib().fib_trim())


def check(rc):
    if rc != FIB_OK:
        raise FibersError(rc, lib().fib_last_error().decode())
    return rc
