"""Two ranks, DEVICE kernels (-m gpu): the multi-GPU flow of fibers.jl_amd/dist.py — z-slab fits, the odfmax all-reduce with its
NaN flag, the all-gather of the slab-fitted orientation field, round-robin seed shards — against the single-rank result of the
same kernels, bit for bit.  The box has one GPU, so both ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one
device); the sharding code is the one bench.py and a multi-GPU host run with backend nccl."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fibers_jl_amd as fj
        from fibers_jl_amd import dist as fd, phantom
        dev = torch.device("cuda", 0)
        shape = (24, 20, 13)                                   # nz = 13: ragged slabs
        nx, ny, nz = shape
        nxy, nvox = nx * ny, nx * ny * nz
        bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
        dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=4, device=dev, noise_frac=0.05)
        rng = np.random.default_rng(1)
        mask = torch.from_numpy((rng.random(nvox) < 0.9).astype(np.uint8)).to(dev)
        plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642)
        z0, z1 = fd.slab_bounds(nz, world, rank)
        v0, v1 = z0 * nxy, z1 * nxy
        counts = [(b - a) * nxy for a, b in (fd.slab_bounds(nz, world, r) for r in range(world))]
        for poison in (False, True):
            d = dwi.clone()
            if poison:
                d[3, nvox - 5] = float("nan")                  # a NaN voxel in the LAST slab: the flag must reach every rank
            full = fj.odf_rec_device(plan, d, mask, normalize=True)
            torch.cuda.synchronize()
            full = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in full.items()}
            loc = fd.odf_rec_sharded(plan, d[:, v0:v1].contiguous(), mask[v0:v1].contiguous(), counts=counts)
            torch.cuda.synchronize()
            assert torch.equal(loc["odfmax"].nan_to_num(nan=-7.0), full["odfmax"].nan_to_num(nan=-7.0)), (rank, poison, loc["odfmax"], full["odfmax"])
            assert torch.equal(loc["odf"].nan_to_num(nan=-7.0), full["odf"][:, v0:v1].nan_to_num(nan=-7.0))
            for k in range(3):
                assert torch.equal(loc["peak"][k], full["peak"][k][:, v0:v1])
                assert torch.equal(loc["qa"][k].nan_to_num(nan=-7.0), full["qa"][k][v0:v1].nan_to_num(nan=-7.0))
            if poison:
                assert float(loc["odfmax"][1]) == 1.0 and bool(torch.isnan(loc["odfmax"][0]))
                continue
            # ---- field from the slab's peaks, all-gathered; seeds round-robin; lines merged back into reference order ----
            f_loc, m_loc = fj.stream_field_device(loc["peak"], f=loc["qa"], f_thresh=0.03, mask=mask[v0:v1].contiguous())
            f_full, m_full = fj.stream_field_device(full["peak"], f=full["qa"], f_thresh=0.03, mask=mask)
            field = fd.allgather_slabs(f_loc, counts)
            mout = fd.allgather_slabs(m_loc, counts)
            assert torch.equal(field, f_full) and torch.equal(mout, m_full)
            seeds = torch.nonzero(mout).flatten()
            sub = torch.from_numpy(fj.make_sublist(2, np.random.default_rng(3))).to(dev)
            one = fj.stream_device(f_full, shape, seeds, sub, len_min=2)
            mine = fd.stream_sharded(field, shape, seeds, sub, len_min=2)
            parts = fd.gather_objects({k: v.cpu().numpy() for k, v in mine.items()})
            merged = fd.merge_tracts(parts)
            assert np.array_equal(merged["npts"], one["npts"].cpu().numpy())
            assert np.array_equal(merged["seed_index"], one["seed_index"].cpu().numpy())
            assert np.array_equal(merged["xyz"], one["xyz"].cpu().numpy())
        # ---- BASELINE config 5 in small: DSI slabs (dsi.jl:197), global odfmax (dsi.jl:263), the 3-peak field + qa all-gathered,
        # seeds x 3 sub-voxel offsets round-robin (stream.jl:757-761), lines merged back into the reference's order ----------------
        shape5 = (12, 10, 7)
        nx, ny, nz = shape5
        nxy, nvox = nx * ny, nx * ny * nz
        b5, g5 = phantom.scheme_dsi()
        d5, _ = phantom.make_dwi_torch(shape5, b5, g5, seed=5, device=dev, noise_frac=0.05)
        m5 = torch.from_numpy((np.random.default_rng(2).random(nvox) < 0.9).astype(np.uint8)).to(dev)
        p5 = fj.OdfPlan("dsi", b5, g5, fj.sphere_642, hann_width=32)
        z0, z1 = fd.slab_bounds(nz, world, rank)
        v0, v1 = z0 * nxy, z1 * nxy
        counts = [(b - a) * nxy for a, b in (fd.slab_bounds(nz, world, r) for r in range(world))]
        full = fj.odf_rec_device(p5, d5, m5, normalize=True)
        torch.cuda.synchronize()
        full = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in full.items()}
        d5_loc, m5_loc = d5[:, v0:v1].contiguous(), m5[v0:v1].contiguous()
        side = torch.cuda.Stream()                             # a non-default stream: kernels, collective and normalisation in its order
        side.wait_stream(torch.cuda.current_stream())          # (after the slab copies above were queued)
        loc = fd.odf_rec_sharded(p5, d5_loc, m5_loc, stream=side)
        side.synchronize()
        assert torch.equal(loc["odfmax"], full["odfmax"]), (rank, loc["odfmax"], full["odfmax"])
        assert torch.equal(loc["odf"], full["odf"][:, v0:v1]) and torch.equal(loc["pdf"], full["pdf"][:, v0:v1])
        for k in range(3):
            assert torch.equal(loc["peak"][k], full["peak"][k][:, v0:v1])
            assert torch.equal(loc["qa"][k], full["qa"][k][v0:v1])
        f_loc, m_loc = fj.stream_field_device(loc["peak"], f=loc["qa"], f_thresh=0.03, mask=m5[v0:v1].contiguous())
        f_full, m_full = fj.stream_field_device(full["peak"], f=full["qa"], f_thresh=0.03, mask=m5)
        assert f_full.shape[1] == 3                             # three candidate directions per voxel
        field = fd.allgather_slabs(f_loc, counts)
        mout = fd.allgather_slabs(m_loc, counts)
        assert torch.equal(field, f_full) and torch.equal(mout, m_full)
        seeds = torch.nonzero(mout).flatten()
        sub = torch.from_numpy(fj.make_sublist(3, np.random.default_rng(5))).to(dev)
        one = fj.stream_device(f_full, shape5, seeds, sub, len_min=2)
        mine = fd.stream_sharded(field, shape5, seeds, sub, len_min=2)
        merged = fd.merge_tracts(fd.gather_objects({k: v.cpu().numpy() for k, v in mine.items()}))
        assert int(one["npts"].numel()) > 100
        assert np.array_equal(merged["npts"], one["npts"].cpu().numpy())
        assert np.array_equal(merged["seed_index"], one["seed_index"].cpu().numpy())
        assert np.array_equal(merged["xyz"], one["xyz"].cpu().numpy())
        # ---- a cut that leaves ONE slab unaligned (13 x 11 x 9: slabs of 715 and 572 voxels; 715 % 4 = 3): the unaligned slab cannot
        # run the fused peak scan, so EVERY slab must take the separate peak finder (FIB_ODF_SEPARATE_PEAKS contract) -- with the
        # slab counts given (no collective) and without them (the ranks agree through one 1-int all-reduce) ---------------------------
        shape_u = (13, 11, 9)
        nx, ny, nz = shape_u
        nxy, nvox = nx * ny, nx * ny * nz
        du, _ = phantom.make_dwi_torch(shape_u, bval, bvec, seed=9, device=dev, noise_frac=0.05)
        mu = torch.from_numpy((np.random.default_rng(4).random(nvox) < 0.9).astype(np.uint8)).to(dev)
        z0, z1 = fd.slab_bounds(nz, world, rank)
        v0, v1 = z0 * nxy, z1 * nxy
        counts = [(b - a) * nxy for a, b in (fd.slab_bounds(nz, world, r) for r in range(world))]
        assert counts == [715, 572] and fd.any_unaligned(counts)
        full = fj.odf_rec_device(plan, du, mu, normalize=True)          # 1287 voxels: unaligned as a whole as well
        torch.cuda.synchronize()
        full = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in full.items()}
        for cnts in (counts, None):
            loc = fd.odf_rec_sharded(plan, du[:, v0:v1].contiguous(), mu[v0:v1].contiguous(), counts=cnts)
            torch.cuda.synchronize()
            assert torch.equal(loc["odfmax"], full["odfmax"]), (rank, loc["odfmax"], full["odfmax"])
            assert torch.equal(loc["odf"], full["odf"][:, v0:v1]), "unaligned cut: the ODF depends on the cut (rank %d)" % rank
            for k in range(3):
                assert torch.equal(loc["peak"][k], full["peak"][k][:, v0:v1]) and torch.equal(loc["qa"][k], full["qa"][k][v0:v1])
        q.put((rank, "ok"))
    except Exception as e:                                     # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def _rccl_worker(port, q):
    """one rank, backend nccl (= RCCL): the library is loaded, a communicator is created and the path's two collectives run"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from fibers_jl_amd import dist as fd
        assert dist.get_backend() == "nccl"
        # {odfmax, NaN flag}: all-reduce(MAX), finite and NaN (gqi.jl:164: maximum() propagates NaN)
        om = torch.tensor([0.25, 0.0], device=dev)
        fd.allreduce_odfmax(om, always=True)
        torch.cuda.synchronize()
        assert om.tolist() == [0.25, 0.0]
        om = torch.tensor([0.5, 1.0], device=dev)              # raw pair: {maximum of the non-NaN means, NaN flag}
        fd.allreduce_odfmax(om, always=True)
        torch.cuda.synchronize()
        assert om.tolist() == [0.5, 1.0] and bool(torch.isnan(fd.odfmax_value(om)))
        # slab all-gather of a float4 field (all_gather_into_tensor) and of a mask
        g = torch.Generator(device=dev); g.manual_seed(3)
        f = torch.rand((4096, 3, 4), device=dev, generator=g)
        full = fd.allgather_slabs(f, [4096], always=True)
        mk = (torch.rand(4096, device=dev, generator=g) < 0.5).to(torch.uint8)
        mfull = fd.allgather_slabs(mk, [4096], always=True)
        torch.cuda.synchronize()
        assert full.data_ptr() != f.data_ptr() and torch.equal(full, f) and torch.equal(mfull, mk)
        # the sharded driver end to end on a non-default stream (one slab = the whole volume)
        import fibers_jl_amd as fj
        from fibers_jl_amd import phantom
        shape = (16, 12, 8)
        bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
        dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=4, device=dev)
        mask = torch.ones(16 * 12 * 8, dtype=torch.uint8, device=dev)
        plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642)
        ref = fj.odf_rec_device(plan, dwi, mask, normalize=True)
        torch.cuda.synchronize()
        ref = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in ref.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        got = fd.odf_rec_sharded(plan, dwi, mask, stream=side)
        side.synchronize()
        for k in range(3):
            assert torch.equal(got["qa"][k], ref["qa"][k])
        got = fd.odf_rec_sharded(plan, dwi, mask, stream=side, always=True)    # .. and with the collectives taken (one-rank group)
        side.synchronize()
        assert torch.equal(got["odfmax"], ref["odfmax"])
        for k in range(3):
            assert torch.equal(got["qa"][k], ref["qa"][k])
        objs = fd.gather_objects({"rank": 0, "n": 3}, always=True)             # all_gather_object through RCCL
        assert objs == [{"rank": 0, "n": 3}]
        q.put("ok")
    except Exception as e:                                     # noqa: BLE001
        import traceback
        q.put("FAIL: %s\n%s" % (e, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_backend_runs_the_paths_collectives_with_one_rank():
    """RCCL itself (torch.distributed backend "nccl") on the GPU box: process group, all-reduce(MAX) of {odfmax, NaN flag},
    all_gather_into_tensor of the field -- with world_size 1, which is what a 1-GPU box allows; the 2-rank control flow is the
    gloo test below, and `bench.py --gpus N` runs these same calls with N ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=60)
    assert res == "ok", res


def test_two_ranks_with_device_kernels_match_one_rank_bit_for_bit():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=600) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), "\n".join("rank %s: %s" % (r[0], r[1]) for r in sorted(res))


def test_bench_rccl_branch_with_one_rank():
    """bench.py's backend-nccl branch (init_process_group("nccl", device_id=...), barrier, float64 all-reduces, the sharded drivers with
    their all-reduce, the field all-gather) executed on the GPU box: FIBERS_BENCH_FORCE_PG=1 takes every multi-rank branch with ONE
    rank -- what a 1-GPU box allows; the first 8-GPU launch is then not this code's first execution"""
    import json
    import subprocess
    env = dict(os.environ, FIBERS_BENCH_FORCE_PG="1", FIBERS_BENCH_SHAPE="40,36,30", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("FIBERS_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["extra"]["gqi_weak_scaling"]["mvoxels_per_s"] > 0                  # (only the multi-rank branch produces it)
    assert line["extra"]["stream_dti_ball"]["points"] > 0 and line["extra"]["stream_dsi_3peaks_10M"]["points"] > 0


def test_bench_multi_rank_control_flow_on_one_device():
    """bench.py's N = 2 path (slabs, all-reduce, field all-gather, seed shards, max-over-ranks timing) end to end on a small volume"""
    import json
    import subprocess
    env = dict(os.environ, FIBERS_BENCH_BACKEND="gloo", FIBERS_BENCH_ONE_DEVICE="1", FIBERS_BENCH_SHAPE="40,36,30")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["extra"]["stream_dti_ball"]["points"] > 0 and line["extra"]["gqi_weak_scaling"]["mvoxels_per_s"] > 0
    # BASELINE config 5 has a multi-rank path: DSI slabs + all-reduced odfmax, 3-peak field all-gathered, seeds x 10 offsets round-robin
    assert line["extra"]["dsi_rec_140x515"]["mvoxels_per_s"] > 0 and line["extra"]["stream_dsi_3peaks_10M"]["points"] > 0
    assert line["cpu_baseline"] is not None and line["cpu_baseline"]["value"] > 0
