"""fibers.jl_amd — MI355X (gfx950) back end for the Fibers.jl reconstruction + tractography hot path.

Host-side mirror of the reference's exported surface for that path (same names, argument meaning and
error behaviour): `MRI`, `ODF`, `sphere_362/642/724`, `DTI`, `dti_fit`, `adc_fit`, ... on top of the
C ABI in include/fibers_hip.h.  All compute happens in libfibers_hip.so (hand-written HIP); there is
no CPU path in this package."""
from ._lib import DEVICE_ALL, FibersError, LIB_PATH, init, lib, shutdown, trim  # noqa: F401
from .mri import MRI  # noqa: F401
from .odf import ODF, sphere_362, sphere_642, sphere_724  # noqa: F401
from .dti import DTI, DtiPlan, adc_fit, adc_fit_device, dti_fit, dti_fit_device  # noqa: F401
from .gqi import (DSI, GQI, OdfPlan, dsi_rec, find_peaks, find_peaks_device, find_peaks_work, gqi_rec, odf_rec_device,  # noqa: F401
                  qa_normalize_device)
from .rumba import RUMBASD, RumbaPlan, rumba_rec, rumba_rec_device  # noqa: F401
from .structens import st_eigen, st_eigen_device  # noqa: F401
from .tract import Tract  # noqa: F401
from .stream import (StreamBuffers, StreamWorkspace, angles_to_vectors, angles_to_vectors_device, make_sublist, stream,  # noqa: F401
                     stream_device, stream_device_run, stream_device_run_enqueue, stream_field_device)
from .nifti import (dsi_write, dti_write, gqi_write, load_nifti, mri_read, mri_read_bfiles, mri_write, rumba_write,  # noqa: F401
                    read_struct)
from .trk import str_add, stream_to_trk, tract_header, trk_read, trk_write  # noqa: F401
