#!/usr/bin/env python3
"""Guards the contraction kernels' software pipeline against the compiler: in every instantiation of odf_gemm3_kernel /
odf_dsi2_kernel the stage loop issues the NEXT stage's loads (direct-to-LDS pieces + the lanes' samples) and then runs the
MFMA block; an `s_waitcnt vmcnt(N)` that hipcc puts between those loads and the first MFMA of the block waits for the loads
just issued and serialises memory latency with the matrix cores (round 3: a register-allocation change did exactly that and
cost 11 % with an otherwise identical instruction stream).

Compiles odf.hip to gfx950 assembly (no GPU needed) and, per kernel, lists every vmcnt wait that sits between a VMEM load and
the next MFMA with no barrier in between, after at least a stage's worth of loads (8).  Exit code 1 if any is found.

  python tools/check_loop_waits.py [path/to/odf.hip] [-D...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(src, defs):
    if src.endswith(".s"):
        return open(src).read()
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out,
           "-I", os.path.dirname(src), src] + defs
    subprocess.run(cmd, check=True)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    """name -> instruction lines of every contraction kernel"""
    res, name, body = {}, None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w*(odf_gemm3_kernel|odf_dsi2_kernel)\w*):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            res[name] = body
            name = None
            continue
        s = line.split(";")[0].strip()
        if s and not s.startswith("."):
            body.append(s)
        elif s.startswith(".LBB"):
            body.append(s)
    return res


MIN_LOADS = 8      # a stage issues >= 8 sample loads + its pieces; fewer = the prologue's own waits


def check(body):
    """vmcnt waits with an un-waited VMEM load before them and an MFMA after them, in layout order with no barrier in between"""
    bad = []
    loads_since_sync = 0
    pending_wait = None
    for i, ins in enumerate(body):
        op = ins.split()[0]
        if op in ("s_barrier", "s_endpgm"):
            loads_since_sync, pending_wait = 0, None
            continue
        if ins.startswith(".LBB") or op.startswith("s_cbranch") or op == "s_branch":
            continue                                     # (the short skips around s_setprio / a dead stage's loads stay inside the run)
        if re.match(r"(buffer_load|global_load)", op):
            loads_since_sync += 1
            continue
        if op == "s_waitcnt" and "vmcnt" in ins:
            n = int(re.search(r"vmcnt\((\d+)\)", ins).group(1))
            if loads_since_sync > n and loads_since_sync >= MIN_LOADS:
                pending_wait = (i, ins, loads_since_sync)
            continue
        if op.startswith("v_mfma") and pending_wait is not None:
            bad.append(pending_wait)
            pending_wait = None
            loads_since_sync = 0
    return bad


VMEM = re.compile(r"(buffer_|global_|flat_|scratch_)(load|store|atomic)")
LDS_DMA = re.compile(r"(buffer_load_\w+ .* lds|global_load_lds_)")
SAMPLE_REQ = re.compile(r"buffer_load_dwordx4 v\d+, s\[\d+:\d+\], 0 offen( nt)? lds")   # (a piece request carries a scalar offset instead of the 0)


def check_slds(body):
    """Kernels whose samples travel through LDS (they issue `buffer_load_dwordx4 v, s[rsrc], 0 offen [nt] lds` requests): the stage
    loop's synchronisation is hand-counted, and a reordering by the compiler or an extra load in the split would silently leave a
    stale sample tile.  A stage's request is N instructions: 2 in the fused GQI kernel (non-temporal), 4 in the DSI pair kernel (two
    fold sides, default cache policy: the partner tile reads the same samples).  Asserted on the assembly:
      * every `s_waitcnt vmcnt(N)` behind the first MFMA (= in the stage loop) has N sample requests as the LAST N vector-memory
        instructions before it -- vmcnt(N) then means "everything but the request just issued";
      * between the first row store (`global_store_dwordx4`) behind each MFMA run and the last request to LDS before it there is
        an `s_waitcnt vmcnt(0)` (the DRAIN wait: after the stores no wait can tell the requests in flight from the stores)."""
    reqs = [x for x in body if SAMPLE_REQ.match(x)]
    if not reqs:
        return []
    n = 2 if all(" nt " in x for x in reqs) else 4
    bad = []
    first_mfma = next((i for i, x in enumerate(body) if x.startswith("v_mfma")), None)
    if first_mfma is None:
        return ["no MFMA in a kernel with sample requests"]
    nwait = 0
    for i, ins in enumerate(body):
        if i < first_mfma or not (ins.startswith("s_waitcnt") and "vmcnt(%d)" % n in ins):
            continue
        nxt = next((x for x in body[i + 1:i + 4] if not x.startswith(".LBB")), "")
        if not nxt.startswith("s_barrier"):      # (a wait of the compiler's own for its own loads, e.g. a table prologue: not a stage's closing wait)
            continue
        nwait += 1
        prev = [x for x in body[:i] if VMEM.match(x)][-n:]
        if len(prev) < n or not all(SAMPLE_REQ.match(x) for x in prev):
            bad.append("line %d: vmcnt(%d) does not follow the %d sample requests (last vector-memory instructions: %s)" % (i, n, n, prev))
    if nwait == 0:
        bad.append("no vmcnt(%d) wait in front of a barrier in the stage loop" % n)
    # the DRAIN wait: walk back from the first row store after an MFMA run -- an s_waitcnt vmcnt(0) must come before any LDS-DMA request
    seen_mfma = False
    ndrain = 0
    for i, ins in enumerate(body):
        if ins.startswith("v_mfma"):
            seen_mfma = True
            continue
        if seen_mfma and re.match(r"global_store_dwordx4 ", ins):
            j = i - 1
            ok = False
            while j >= 0 and not LDS_DMA.match(body[j]):
                if body[j].startswith("s_waitcnt") and "vmcnt(0)" in body[j]:
                    ok = True
                    break
                j -= 1
            ndrain += 1
            if not ok:
                bad.append("line %d: a request to LDS is still in flight at the first row store after the MFMA block (no s_waitcnt vmcnt(0) in between)" % i)
            seen_mfma = False
    if ndrain == 0:
        bad.append("no row store found behind the MFMA block")
    return bad


def demangled(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name


def main():
    args = sys.argv[1:]
    defs = [a for a in args if a.startswith("-D")]
    srcs = [a for a in args if not a.startswith("-D")]
    src = srcs[0] if srcs else os.path.join(ROOT, "fibers.jl_amd", "csrc", "odf.hip")
    ks = kernels(device_asm(src, defs))
    nbad = 0
    for name, body in sorted(ks.items()):
        bad = check(body)
        nm = sum(1 for x in body if x.startswith("v_mfma"))
        short = re.sub(r"\(anonymous namespace\)::|\(\(anonymous namespace\)::GemmArgs\)", "", demangled(name))
        print("%-60s %5d instructions, %3d MFMAs: %s" % (short, len(body), nm, "ok" if not bad else "%d WAITS ON FRESH LOADS" % len(bad)))
        for i, ins, nl in bad:
            print("      line %d: %s  (after %d loads issued since the last barrier / branch)" % (i, ins, nl))
        nbad += len(bad)
        sl = check_slds(body)
        if any(SAMPLE_REQ.match(x) for x in body):
            print("%-60s sample-tile synchronisation: %s" % ("", "slds ok" if not sl else "%d PROBLEMS" % len(sl)))
        for msg in sl:
            print("      " + msg)
        nbad += len(sl)
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
