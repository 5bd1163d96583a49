"""world_size-2 (gloo, CPU) tests of the multi-GPU host logic in fibers.jl_amd/dist.py: z-slab sharding of
the fits with the odfmax all-reduce, slab all-gather of the orientation field, round-robin seed sharding
and restoration of the reference's line order.  The compute step is the CPU oracle standing in for the
device kernels (this is a test: no GPU here), so that the sharded result can be compared with the
single-process result exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fibers_jl_amd as fj
        from fibers_jl_amd import dist as fd, phantom
        from oracle import oracle as orc
        sph = fj.sphere_362
        shape = (6, 5, 7)                                   # nz = 7 does not divide by 2: ragged slabs
        bval, bvec = phantom.scheme_gqi(2, 12, (1000.0, 2500.0), 3)
        dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed=4, crossing=True)
        mask = (np.random.default_rng(1).random(shape) < 0.9).astype(np.uint8)
        full = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25)

        # ---- fits: z-slab + 1-float all-reduce(MAX) -------------------------------------------
        z0, z1 = fd.slab_bounds(shape[2], world, rank)
        loc = orc.gqi_rec(dwi[:, :, z0:z1], mask[:, :, z0:z1], bval, bvec, sph.vertices, sph.faces, 1.25)
        # undo the local normalisation to get what the device tier returns with normalize=False
        qa_raw = [q * np.float32(loc["odfmax"]) for q in loc["qa"]]
        om = torch.tensor([loc["odfmax"], 0.0], dtype=torch.float32)
        fd.allreduce_odfmax(om)
        assert abs(float(om[0]) - full["odfmax"]) <= 1e-6 * abs(full["odfmax"])
        for k in range(3):
            got = qa_raw[k] / np.float32(om[0])
            np.testing.assert_allclose(got, full["qa"][k][:, :, z0:z1], rtol=2e-6, atol=1e-7)
            assert np.array_equal(loc["peak"][k], full["peak"][k][:, :, z0:z1])
        assert np.array_equal(loc["odf"], full["odf"][:, :, z0:z1])

        # NaN in one rank's maximum must win everywhere (Julia maximum propagates NaN): the raw pair {maximum of the means that
        # are not NaN, NaN flag} goes through ONE all-reduce(MAX); the flag turns the result into NaN
        om2 = torch.tensor([2.0 if rank == 1 else 3.0, 1.0 if rank == 1 else 0.0])
        fd.allreduce_odfmax(om2)
        assert om2.tolist() == [3.0, 1.0] and torch.isnan(fd.odfmax_value(om2))
        om3 = torch.tensor([float("-inf") if rank == 1 else 3.0, 0.0])        # a rank with no voxel at all
        fd.allreduce_odfmax(om3)
        assert om3.tolist() == [3.0, 0.0] and float(fd.odfmax_value(om3)) == 3.0

        # ---- field all-gather from ragged slabs ------------------------------------------------
        nxy = shape[0] * shape[1]
        counts = [(b - a) * nxy for a, b in (fd.slab_bounds(shape[2], world, r) for r in range(world))]
        pk = full["peak"][0].reshape(-1, 3, order="F")       # [nvox, 3] voxel-major
        mine = torch.from_numpy(np.ascontiguousarray(pk[z0 * nxy: z1 * nxy]))
        allf = fd.allgather_slabs(mine, counts)
        assert np.array_equal(allf.numpy(), pk)

        # ---- tracking: round-robin seeds, merge restores reference order ------------------------
        ov = [np.asfortranarray(p) for p in full["peak"]]
        fs = [np.asfortranarray(q) for q in full["qa"]]
        sub = np.array([[0.1, -0.2, 0.3], [-0.3, 0.2, 0.1]], np.float32)
        kw = dict(f=fs, f_thresh=0.03, mask=mask, len_min=2)
        ref = orc.stream(ov, sub, **kw)
        mk, _ = orc.stream_work(ov, fs, 0.03, None, 0.1, mask)
        seeds_all = orc.seeds_from_mask(mk)
        lin = (seeds_all[:, 0] - 1) + shape[0] * ((seeds_all[:, 1] - 1) + shape[1] * (seeds_all[:, 2] - 1))
        mine_lin, gi = fd.shard_seeds(lin.astype(np.int64), world, rank)
        seedvol = np.zeros(shape, np.uint8, order="F")
        seedvol.reshape(-1, order="F")[mine_lin] = 1
        part = orc.stream(ov, sub, seed=seedvol, **kw)
        ls = part["seed_index"] // 2
        part_g = dict(npts=part["npts"], xyz=part["xyz"], seed_index=gi[ls] * 2 + part["seed_index"] % 2)
        parts = fd.gather_objects(part_g)
        merged = fd.merge_tracts(parts)
        assert np.array_equal(merged["npts"], ref["npts"])
        assert np.array_equal(merged["seed_index"], ref["seed_index"])
        assert np.array_equal(merged["xyz"], ref["xyz"])
        q.put((rank, "ok"))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_sharding_world2_gloo():
    from oracle import oracle
    oracle.lib()                                             # build once before forking workers
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_slab_bounds_and_seed_shards():
    sys.path.insert(0, ROOT)
    from fibers_jl_amd import dist as fd
    for nz in (1, 7, 140):
        for w in (1, 2, 3, 8):
            b = [fd.slab_bounds(nz, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == nz
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(z1 - z0 for z0, z1 in b) - min(z1 - z0 for z0, z1 in b) <= 1
    # with the slice size given the cut keeps every slab's voxel count a multiple of 32 where the slice count allows it (the slabs' rows
    # then start on cache lines: tools/slab_alignment.py), still contiguous, complete and near-equal
    for nz, nxy in ((140, 140 * 140), (141, 140 * 140), (7, 140 * 140), (64, 33 * 31), (50, 128 * 128)):
        for w in (1, 2, 3, 4, 8):
            b = [fd.slab_bounds(nz, w, r, nxy) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == nz and all(b[i][1] == b[i + 1][0] for i in range(w - 1)), (nz, nxy, w, b)
            unit = 32 // np.gcd(nxy, 32)
            if nz // unit >= w:
                assert all(((z1 - z0) * nxy) % 32 == 0 for z0, z1 in b[:-1]), (nz, nxy, w, b)
                assert max(z1 - z0 for z0, z1 in b) - min(z1 - z0 for z0, z1 in b) <= 2 * unit
            else:
                assert b == [fd.slab_bounds(nz, w, r) for r in range(w)]
    assert [z1 - z0 for z0, z1 in (fd.slab_bounds(140, 8, r, 19600) for r in range(8))] == [18, 18, 18, 18, 18, 18, 16, 16]
    seeds = np.arange(11) * 3
    got = np.sort(np.concatenate([fd.shard_seeds(seeds, 4, r)[0] for r in range(4)]))
    assert np.array_equal(got, seeds)
