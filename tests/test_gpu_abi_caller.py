"""The drop-in boundary from plain C (tests/abi_caller.c): a C program that includes include/fibers_hip.h, links libfibers_hip.so and
nothing else, and checks fib_dti_fit / fib_adc_fit / fib_stream against analytic known answers -- what the Julia wrapper does through
`ccall` (dti.jl:221, dti.jl:164, stream.jl:730), without Python or torch in the process."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_program_calls_the_boundary_without_python(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    libdir = os.path.join(ROOT, "fibers.jl_amd")
    exe = str(tmp_path / "abi_caller")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "abi_caller.c"), "-o", exe,
           "-L" + libdir, "-l:libfibers_hip.so", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "abi_caller ok" in run.stdout, (run.stdout[-2000:], run.stderr[-3000:])
