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
    """round-robin: global seed i -> rank i % world; returns (local seeds, global seed numbers), NumPy in -> NumPy out,
    torch in -> torch out (same device)"""
    if isinstance(seeds, np.ndarray):
        idx = np.arange(rank, len(seeds), world)
        return seeds[idx], idx
    import torch
    gi = torch.arange(rank, seeds.numel(), world, device=seeds.device)
    return seeds[gi], gi


def _torch_stream(stream):
    """None, a torch.cuda.Stream or a raw hipStream_t handle -> the torch stream object (None: torch's current stream)"""
    import torch
    if stream is None or isinstance(stream, torch.cuda.Stream):
        return stream
    return torch.cuda.ExternalStream(int(stream))


def allreduce_odfmax(odfmax, group=None, always=False):
    """odfmax: tensor [2] = {local max of per-voxel ODF means, nan flag}; in-place MAX over ranks.
    A NaN anywhere must win (Julia's maximum propagates NaN): the flag is reduced too.
    always: run the collective even in a one-rank group (tests: exercises the RCCL path on a 1-GPU box)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or always):
        m = torch.nan_to_num(odfmax[:1], nan=float("-inf"))
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        fl = odfmax[1:2].clone()
        dist.all_reduce(fl, op=dist.ReduceOp.MAX, group=group)
        odfmax[0] = torch.where(fl[0] > 0, torch.full_like(m[0], float("nan")), m[0])
        odfmax[1] = fl[0]
    return odfmax


def allgather_slabs(local, counts: Sequence[int], group=None, always=False):
    """local: [counts[rank], ...] slab (voxel-major, e.g. the float4 field [nvox_local, nvec, 4]); returns the
    full volume [sum(counts), ...] on every rank.  Slabs may differ in size (nz % world != 0): with RCCL the
    collective is one padded all_gather (G simultaneous broadcasts over the xGMI links); backends without a device
    all_gather (gloo on CUDA tensors, used by the 1-GPU tests) run the G broadcasts one after the other."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and (dist.get_world_size(group) > 1 or always)):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [int(c) for c in counts]
    assert local.shape[0] == counts[rank]
    full = local.new_empty((sum(counts),) + tuple(local.shape[1:]))
    offs = np.concatenate([[0], np.cumsum(counts)])
    if local.is_cuda and dist.get_backend(group) != "nccl":
        for r in range(world):
            piece = full[offs[r]:offs[r + 1]]
            if r == rank:
                piece.copy_(local)
            dist.broadcast(piece, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
        return full
    if len(set(counts)) == 1:                                  # equal slabs: gather straight into the result
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
        return full
    pad = local.new_zeros((max(counts),) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    for r in range(world):
        full[offs[r]:offs[r + 1]] = bufs[r][:counts[r]]
    return full


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
def odf_rec_sharded(plan, dwi_local, mask_local, group=None, stream=None, out=None):
    """gqi_rec / dsi_rec on this rank's z-slab + the global QA normalisation across ranks (gqi.jl:164-168): the slab's
    {odfmax, NaN flag} pair is all-reduced with MAX, the divisor never leaves the device."""
    import contextlib
    import torch
    from .gqi import odf_rec_device, qa_normalize_device
    out = odf_rec_device(plan, dwi_local, mask_local, out=out, normalize=False, stream=stream)
    # the collective runs on the stream the kernels run on: it follows the kernel that writes out["odfmax"] and precedes the
    # normalisation in stream order, whatever torch's current stream is
    ts = _torch_stream(stream)
    with (torch.cuda.stream(ts) if ts is not None else contextlib.nullcontext()):
        allreduce_odfmax(out["odfmax"], group)
    qa_normalize_device(out["qa"], out["odfmax"], stream=stream)
    return out


def stream_sharded(field_full, shape, seeds_all, sublist, group=None, **kw):
    """round-robin seed shard of stream_device; returns this rank's lines with GLOBAL seed_index.  seeds_all: int64 CUDA
    tensor (or NumPy array) of the whole seed list in the reference's findall order."""
    import torch
    import torch.distributed as dist
    from .stream import stream_device
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if isinstance(seeds_all, np.ndarray):
        seeds_all = torch.from_numpy(np.ascontiguousarray(seeds_all, np.int64)).to(field_full.device)
    if world == 1:                                             # the whole list: nothing to renumber
        return stream_device(field_full, shape, seeds_all, sublist, **kw)
    local, gi = shard_seeds(seeds_all, world, rank)
    res = stream_device(field_full, shape, local.contiguous(), sublist, **kw)
    nsub = sublist.shape[0]
    ls = res["seed_index"] // nsub
    res["seed_index"] = gi[ls] * nsub + (res["seed_index"] - ls * nsub)
    return res
