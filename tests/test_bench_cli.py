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
