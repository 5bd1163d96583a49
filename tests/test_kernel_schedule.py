"""Compiler-output guard for the contraction kernels (no GPU: hipcc cross-compiles odf.hip to gfx950 assembly, ~90 s)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_stage_loops_do_not_wait_on_the_loads_they_have_just_issued():
    """every odf_gemm3_kernel / odf_dsi2_kernel instantiation: no `s_waitcnt vmcnt` between a stage's prefetch (>= 8 loads) and the
    MFMA block that is meant to cover it (tools/check_loop_waits.py; in round 3 a conditionally consumed load made hipcc put a
    vmcnt(0) there -- same instructions otherwise, 11 % slower)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_loop_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(": ok") >= 20, r.stdout[-2000:]
    # the kernel whose samples travel through LDS: its hand-counted vmcnt(2) waits follow exactly the two sample requests, and the
    # DRAIN wait sits directly in front of the first row store (tools/check_loop_waits.py check_slds)
    assert r.stdout.count("slds ok") >= 1, r.stdout[-2000:]
