"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  The reference is single-process (Threads.@threads over z-slices,
dti.jl:258 / gqi.jl:132 / dsi.jl:197, contiguous seed chunks, stream.jl:757-761); here

  * fits shard by contiguous z-slabs (voxels are independent): no halo, no data-path collective; the only
    exchange step is odfmax = max_vox mean_v(odf) (gqi.jl:164, dsi.jl:263) -> 1-float all-reduce(MAX);
  * the orientation field every tracker needs in full is all-gathered from the slabs that fitted it
    (one collective of 16*nvec bytes per voxel: 44-132 MB at 140^3);
  * seeds shard round-robin (seed i -> rank i mod G) for balance; `seed_index` restores reference order.

Functions take the compute step as a callable so that the same sharding code is exercised on CPU (gloo +
the oracle as stand-in compute, tests/test_dist_gloo.py) and on GPUs (device-tier functions)."""
from typing import Callable, List, Sequence, Tuple

import numpy as np


def slab_bounds(nz: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous near-equal z-slabs, like Threads.@threads :static over 1:nz"""
    q, r = divmod(nz, world)
    z0 = rank * q + min(rank, r)
    return z0, z0 + q + (1 if rank < r else 0)


def slab_of_planar(vol, shape, z0, z1):
    """rows [z0,z1) of a planar [nframes, nvox] tensor/array (x fastest): a contiguous voxel range per frame"""
    nx, ny, _ = shape
    return vol[..., z0 * nx * ny: z1 * nx * ny]


def shard_seeds(seeds, world: int, rank: int):
    """round-robin: global seed i -> rank i % world; returns (local seeds, global seed numbers)"""
    idx = np.arange(rank, len(seeds), world) if isinstance(seeds, np.ndarray) else None
    if idx is not None:
        return seeds[idx], idx
    import torch
    gi = torch.arange(rank, seeds.numel(), world, device=seeds.device)
    return seeds[gi], gi


def allreduce_odfmax(odfmax, group=None):
    """odfmax: tensor [2] = {local max of per-voxel ODF means, nan flag}; in-place MAX over ranks.
    A NaN anywhere must win (Julia's maximum propagates NaN): the flag is reduced too."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        m = torch.nan_to_num(odfmax[:1], nan=float("-inf"))
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        fl = odfmax[1:2].clone()
        dist.all_reduce(fl, op=dist.ReduceOp.MAX, group=group)
        odfmax[0] = torch.where(fl[0] > 0, torch.full_like(m[0], float("nan")), m[0])
        odfmax[1] = fl[0]
    return odfmax


def allgather_slabs(local, counts: Sequence[int], group=None):
    """local: [counts[rank], ...] slab (voxel-major, e.g. the float4 field [nvox_local, nvec, 4]); returns the
    full volume [sum(counts), ...] on every rank.  Slabs may differ in size (nz % world != 0), so the
    collective is a padded all_gather (== G broadcasts over xGMI)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return local
    world = dist.get_world_size(group)
    counts = [int(c) for c in counts]
    assert local.shape[0] == counts[dist.get_rank(group)]
    pad = local.new_zeros((max(counts),) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def merge_tracts(parts: List[dict]) -> dict:
    """concatenate per-rank results (npts, seed_index = global seed*nsub+sub, xyz) and restore the
    reference's (seed, sub) order (stream.jl:761-787)."""
    npts = np.concatenate([p["npts"] for p in parts])
    sidx = np.concatenate([p["seed_index"] for p in parts])
    xyz = np.concatenate([p["xyz"] for p in parts])
    off = np.concatenate([[0], np.cumsum(npts, dtype=np.int64)])
    order = np.argsort(sidx, kind="stable")
    pieces = [xyz[off[i]:off[i + 1]] for i in order]
    return dict(npts=npts[order], seed_index=sidx[order],
                xyz=np.concatenate(pieces) if pieces else np.zeros((0, 3), np.float32))


def gather_objects(obj, group=None):
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return [obj]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


# ---------------------------------------------------------------------------------------------
# sharded drivers (device tier)
# ---------------------------------------------------------------------------------------------
def odf_rec_sharded(plan, dwi_local, mask_local, group=None, stream=None):
    """gqi_rec / dsi_rec on this rank's z-slab + the global QA normalisation across ranks."""
    from .gqi import odf_rec_device, qa_normalize_device
    import torch
    out = odf_rec_device(plan, dwi_local, mask_local, normalize=False, stream=stream)
    allreduce_odfmax(out["odfmax"], group)
    qa_normalize_device(out["qa"], out["odfmax"], stream=stream)     # the divisor is read on the device: no host round trip
    return out


def stream_sharded(field_full, shape, seeds_all, sublist, group=None, **kw):
    """round-robin seed shard of stream_device; returns this rank's lines with GLOBAL seed_index."""
    import torch
    import torch.distributed as dist
    from .stream import stream_device
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    local, gi = shard_seeds(seeds_all, world, rank)
    res = stream_device(field_full, shape, local.contiguous(), sublist, **kw)
    nsub = sublist.shape[0]
    ls = res["seed_index"] // nsub
    res["seed_index"] = gi[ls] * nsub + (res["seed_index"] - ls * nsub)
    return res
