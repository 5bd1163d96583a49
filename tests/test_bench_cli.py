"""bench.py's command line: `--gpus N` means N ranks.  Without a launcher the N ranks are started as child processes of
torch.distributed.run (before the parent imports torch or touches the GPU); with a launcher WORLD_SIZE must equal --gpus; a
`--gpus 8` run can never print an `n_gpus: 1` line.  (Reference split the ranks reproduce: z-slabs gqi.jl:132 + odfmax gqi.jl:164,
seed chunks stream.jl:757-761.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


@pytest.mark.parametrize("gpus,world", [(2, 1), (8, 1), (1, 2)])
def test_world_size_must_equal_gpus(gpus, world):
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--steps", "1"], env=_env(WORLD_SIZE=str(world)),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]          # no result line
    assert "--gpus %d but WORLD_SIZE=%d" % (gpus, world) in out.stderr


def test_self_launch_fails_loudly_when_the_ranks_fail():
    """No GPU here (or, on the 1-GPU box, no second device for RCCL): the child ranks fail, and so must the parent -- without a line."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices: the launch would succeed")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra"],
                         env=_env(FIBERS_BENCH_SHAPE="8,8,8"), capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "2-rank launch failed" in out.stderr


@pytest.mark.gpu
def test_gpus_2_without_a_launcher_runs_two_ranks():
    """`python bench.py --gpus 2`, no launcher: two ranks (here both on cuda:0 over gloo -- the one-device hook, RCCL refuses two ranks
    on one device), one line, n_gpus == 2."""
    env = _env(FIBERS_BENCH_BACKEND="gloo", FIBERS_BENCH_ONE_DEVICE="1", FIBERS_BENCH_SHAPE="40,36,30")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["voxels_per_gpu"] < line["config"]["voxels"]


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _check_result_line(text):
    """what the driver needs of the ONE result line (BENCH_r05.parsed was null because the line was 25 KB)"""
    assert "\n" not in text and len(text) < 6144, len(text)
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line, k
    assert len(line["dtype"]) <= 120 and len(line["config"]["workload"]) <= 200 and len(line["roofline"]["kernel"]) <= 100
    assert line["roofline"]["frac"] > 0 and line["roofline"]["bound"] in ("hbm", "mfma") and "traffic" in line["roofline"]
    for leg, rec in line["extra"].items():                                           # number-only legs: no prose in the line
        for k, v in rec.items():
            vals = v.values() if isinstance(v, dict) else [v]
            assert all(x is None or isinstance(x, (int, float)) for x in vals), (leg, k, v)
    return line


def test_result_line_is_compact_for_the_largest_record_on_file():
    """compact_line() on round 5's full record (25 KB, the one the driver could not parse) and on one with every string blown up"""
    bench = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "bench_r05.json")))
    line = _check_result_line(json.dumps(bench.compact_line(full)))
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["kind"] == "port"
    assert line["extra"]["host_tier"]["gqi_rec"]["ms_median"] == pytest.approx(full["extra"]["host_tier"]["gqi_rec"]["e2e_pcie_ms_median"], rel=1e-4)
    full["dtype"] = "x" * 5000
    full["config"]["workload"] = "y" * 5000
    full["roofline"]["kernel"] = "z" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    for rec in full["extra"].values():
        if isinstance(rec, dict):
            rec["note"] = "n" * 3000
    _check_result_line(json.dumps(bench.compact_line(full)))


@pytest.mark.gpu
def test_small_run_prints_one_parsable_compact_line():
    """the whole bench on a small shape (every leg, CPU baseline included): the LAST stdout line is the result line, < 6 KB, carries
    roofline.frac, cpu_baseline.value and config.workload; the full record goes to an earlier `bench_extra ` line"""
    env = _env(FIBERS_BENCH_SHAPE="40,36,32")
    out = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len([ln for ln in lines if ln.startswith("{")]) == 1 and lines[-1].startswith("{")
    line = _check_result_line(lines[-1])
    assert line["cpu_baseline"]["value"] > 0 and line["config"]["workload"]
    assert [ln for ln in lines if ln.startswith("bench_extra ")]
    failed = [k for k, v in line["extra"].items() if "error" in v]
    assert not failed, failed


@pytest.mark.gpu
def test_gpus_8_without_a_launcher_runs_eight_ranks():
    """`--gpus 8` on the one-device gloo hook with a small shape: eight ranks (slabs 18x6 + 16x2 at nz = 140; here nz = 40), every
    multi-rank leg (slab fits, odfmax all-reduce, field all-gather with unequal slabs, round-robin seeds), one line with n_gpus == 8"""
    env = _env(FIBERS_BENCH_BACKEND="gloo", FIBERS_BENCH_ONE_DEVICE="1", FIBERS_BENCH_SHAPE="24,20,40")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = _check_result_line(lines[0])
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["config"]["voxels_per_gpu"] < line["config"]["voxels"]
    assert line["extra"]["stream_dsi_3peaks_10M"]["points"] > 0 and line["extra"]["stream_dti_ball"]["points"] > 0
