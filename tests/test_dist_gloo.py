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


def _worker_drivers(rank, world, port, q):
    """the sharded DRIVERS themselves (fd.odf_rec_sharded, fd.stream_sharded, fd.allgather_slabs with unequal slabs) at world = 4 / 8
    on a 20 x 20 x 24 volume, with the CPU oracle standing in for the two device calls they make (monkeypatched: this is a test)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fibers_jl_amd as fj
        from fibers_jl_amd import dist as fd, phantom
        fgqi, fstream = sys.modules[fj.odf_rec_device.__module__], sys.modules[fj.stream_device.__module__]   # (fj.stream is the function)
        from oracle import oracle as orc
        sph = fj.sphere_362
        shape = (20, 20, 24)
        nx, ny, nz = shape
        nxy, nvox = nx * ny, nx * ny * nz
        bval, bvec = phantom.scheme_gqi(2, 12, (1000.0, 2500.0), 3)
        dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed=4, crossing=True)
        mask = (np.random.default_rng(1).random(shape) < 0.9).astype(np.uint8)
        mask = np.asfortranarray(mask)

        # ---- slabs: the 32-voxel alignment rule (nxy = 400 -> units of 2 slices), unequal at world = 8 -------------------------
        bounds = [fd.slab_bounds(nz, world, r, nxy) for r in range(world)]
        counts = [(b - a) * nxy for a, b in bounds]
        assert bounds[0][0] == 0 and bounds[-1][1] == nz and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
        assert all(c % 32 == 0 for c in counts) and not fd.any_unaligned(counts)
        if world == 8:
            assert [b - a for a, b in bounds] == [4, 4, 4, 4, 2, 2, 2, 2]           # 12 units of 2 slices over 8 ranks
        z0, z1 = bounds[rank]

        seen = {}

        def fake_odf_rec_device(plan, dwi_l, mask_l, out=None, normalize=True, stream=None, out_prezeroed=False, separate_peaks=False, raw_odfmax=False):
            assert not normalize and raw_odfmax                                    # what the sharded driver must ask for
            seen["separate_peaks"] = separate_peaks
            r = orc.gqi_rec(dwi_l, mask_l, bval, bvec, sph.vertices, sph.faces, 1.25)
            means = r["odf"].reshape(-1, sph.nvert, order="F").mean(axis=1, dtype=np.float32)
            ok = ~np.isnan(means)
            pair = [means[ok].max() if ok.any() else -np.inf, 0.0 if ok.all() else 1.0]
            lm = np.float32(r["odfmax"])
            with np.errstate(all="ignore"):
                qa_raw = [np.where(np.isnan(lm), np.nan, q_ * lm).astype(np.float32) if np.isnan(lm) else (q_ * lm) for q_ in r["qa"]]
            return dict(odf=r["odf"], peak=r["peak"], qa=[torch.from_numpy(np.ascontiguousarray(q_.reshape(-1, order="F"))) for q_ in qa_raw],
                        odfmax=torch.tensor(pair, dtype=torch.float32))

        def fake_qa_normalize_device(qa, odfmax, stream=None, raw=False):
            assert raw
            d = float("nan") if float(odfmax[1]) > 0 else float(odfmax[0])
            for t in qa:
                t /= d
            odfmax[0] = d
        fgqi.odf_rec_device, fgqi.qa_normalize_device = fake_odf_rec_device, fake_qa_normalize_device

        full = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25)
        got = fd.odf_rec_sharded(None, dwi[:, :, z0:z1], mask[:, :, z0:z1], counts=counts)
        assert seen["separate_peaks"] is False
        assert abs(float(got["odfmax"][0]) - full["odfmax"]) <= 1e-6 * abs(full["odfmax"])
        for k in range(3):
            np.testing.assert_allclose(got["qa"][k].numpy(), full["qa"][k][:, :, z0:z1].reshape(-1, order="F"), rtol=3e-6, atol=1e-7)
            assert np.array_equal(got["peak"][k], full["peak"][k][:, :, z0:z1])
        assert np.array_equal(got["odf"], full["odf"][:, :, z0:z1])

        # ---- a NaN sample in the LAST slab: maximum(mean(odf)) is NaN (gqi.jl:164), so every rank's qa is NaN where it had a peak ------
        dwi_n = dwi.copy()
        vx = np.argwhere(mask[:, :, nz - 1] > 0)[0]
        dwi_n[vx[0], vx[1], nz - 1, 3] = np.nan
        full_n = orc.gqi_rec(dwi_n, mask, bval, bvec, sph.vertices, sph.faces, 1.25)
        assert np.isnan(full_n["odfmax"])
        got_n = fd.odf_rec_sharded(None, dwi_n[:, :, z0:z1], mask[:, :, z0:z1], counts=counts)
        assert got_n["odfmax"].tolist()[1] == 1.0 and np.isnan(float(got_n["odfmax"][0]))
        for k in range(3):
            ref_q = full_n["qa"][k][:, :, z0:z1].reshape(-1, order="F")
            assert np.array_equal(np.isnan(got_n["qa"][k].numpy()), np.isnan(ref_q))

        # ---- an unaligned cut: 3 x 2 slices (6 voxels); rank 0's slab has 12 voxels (aligned), the others 6 -> EVERY rank must take the
        # separate peak finder, with `counts` and -- through one 1-int all-reduce -- without ----------------------------------------------
        nz_u = world + 1
        counts_u = [(b - a) * 6 for a, b in (fd.slab_bounds(nz_u, world, r, 6) for r in range(world))]
        assert counts_u == [12] + [6] * (world - 1) and fd.any_unaligned(counts_u) and not fd.any_unaligned(counts_u[:1])
        bu, gu = phantom.scheme_gqi(2, 12, (1000.0, 2500.0), 3)
        du, _, _ = phantom.make_volume((3, 2, nz_u), bu, gu, seed=5, crossing=True)
        zu0, zu1 = fd.slab_bounds(nz_u, world, rank, 6)
        mu = np.ones((3, 2, zu1 - zu0), np.uint8)
        fd.odf_rec_sharded(None, du[:, :, zu0:zu1], mu, counts=counts_u)
        assert seen["separate_peaks"] is True
        seen.clear()
        fd.odf_rec_sharded(None, du[:, :, zu0:zu1], torch.from_numpy(mu.reshape(-1)), counts=None)
        assert seen["separate_peaks"] is True                                       # (rank 0 too, whose own 12 voxels are aligned)

        # ---- the 3-peak field [nvox, 3, 4] (+ per-vector masks in w) all-gathered from UNEQUAL slabs (padded all_gather on CPU tensors) ----
        pk3 = np.stack([np.concatenate([full["peak"][k].reshape(-1, 3, order="F"), full["qa"][k].reshape(-1, 1, order="F")], axis=1) for k in range(3)], axis=1)
        mine = torch.from_numpy(np.ascontiguousarray(pk3[z0 * nxy: z1 * nxy]))
        allf = fd.allgather_slabs(mine, counts)
        assert allf.shape == (nvox, 3, 4) and np.array_equal(allf.numpy(), pk3)
        m_all = fd.allgather_slabs(torch.from_numpy(np.ascontiguousarray(mask.reshape(-1, order="F")[z0 * nxy: z1 * nxy])), counts)
        assert np.array_equal(m_all.numpy(), mask.reshape(-1, order="F"))

        # ---- tracking: round-robin seeds x nsub = 10 through fd.stream_sharded, merged == the one-rank order ----------------------------
        ov = [np.asfortranarray(p) for p in full["peak"]]
        fs = [np.asfortranarray(q_) for q_ in full["qa"]]
        sub = fj.make_sublist(10, np.random.default_rng(5))
        kw = dict(f=fs, f_thresh=0.03, mask=mask, len_min=2)
        ref = orc.stream(ov, sub, **kw)
        mk, _ = orc.stream_work(ov, fs, 0.03, None, 0.1, mask)
        seeds_all = orc.seeds_from_mask(mk)
        lin = ((seeds_all[:, 0] - 1) + nx * ((seeds_all[:, 1] - 1) + ny * (seeds_all[:, 2] - 1))).astype(np.int64)

        def fake_stream_device(field, shp, seeds, sublist, **_kw):
            seedvol = np.zeros(shape, np.uint8, order="F")
            seedvol.reshape(-1, order="F")[seeds.numpy()] = 1
            part = orc.stream(ov, np.asarray(sublist), seed=seedvol, **kw)
            return dict(npts=torch.from_numpy(part["npts"]), xyz=torch.from_numpy(part["xyz"]), seed_index=torch.from_numpy(part["seed_index"].astype(np.int64)))
        fstream.stream_device = fake_stream_device
        part = fd.stream_sharded(allf, shape, lin, torch.from_numpy(sub))
        nloc_seeds = len(range(rank, len(lin), world))
        assert int(part["seed_index"].max()) < len(lin) * 10 and len(torch.unique(part["seed_index"] // 10)) <= nloc_seeds
        assert bool(((part["seed_index"] // 10) % world == rank).all())             # seed i -> rank i mod world
        parts = fd.gather_objects(dict(npts=part["npts"].numpy(), xyz=part["xyz"].numpy(), seed_index=part["seed_index"].numpy()))
        merged = fd.merge_tracts(parts)
        assert len(parts) == world and len(ref["npts"]) > 1000
        assert np.array_equal(merged["npts"], ref["npts"]) and np.array_equal(merged["seed_index"], ref["seed_index"])
        assert np.array_equal(merged["xyz"], ref["xyz"])
        q.put((rank, "ok"))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_sharded_drivers_world_4_and_8_gloo(world):
    """the shape of the first real 8-GPU run (slab bounds incl. the alignment rule, one unaligned slab ->
    FIB_ODF_SEPARATE_PEAKS everywhere, NaN in the last slab, padded all-gather of unequal slabs, 3-peak field, round-robin seeds x
    nsub = 10, merge order == one-rank order) had only ever executed at world = 2"""
    from oracle import oracle
    oracle.lib()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_drivers, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
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
