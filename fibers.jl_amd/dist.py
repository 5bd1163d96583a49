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

import math

import numpy as np


def slab_bounds(nz: int, world: int, rank: int, nxy: int = 0) -> Tuple[int, int]:
    """contiguous near-equal z-slabs, like Threads.@threads :static over 1:nz.

    nxy (= nx * ny, optional): cut in units of slices whose voxel count is a multiple of 32, when there are enough of them.  A slab
    is a planar [frames][voxels] array of its own; with a voxel count that is not a multiple of 32 (128 bytes) its rows start off the
    cache lines and the contraction kernels run slower on it than on a LARGER aligned slab (140 x 140 slices, one MI355X: 17 slices
    0.318 ms per GQI step, 18 slices 0.299; tools/slab_alignment.py).  Results do not depend on the cut."""
    unit = 1
    if nxy > 0:
        unit = 32 // math.gcd(int(nxy), 32)
        if nz // unit < world:
            unit = 1
    nu = nz // unit                                        # whole units; the remainder goes to the last rank
    q, r = divmod(nu, world)
    u0 = rank * q + min(rank, r)
    u1 = u0 + q + (1 if rank < r else 0)
    return u0 * unit, (nz if rank == world - 1 else u1 * unit)


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
    handle = int(getattr(stream, "value", stream) or 0)        # ctypes.c_void_p (the package's own handle type) or a plain integer
    if handle == 0:                                             # the null stream: torch's default stream
        return torch.cuda.default_stream()
    return torch.cuda.ExternalStream(handle)


def allreduce_odfmax(odfmax, group=None, always=False, raw=True):
    """odfmax: tensor [2] = {local max of the per-voxel ODF means that are not NaN (-Inf if there is none), NaN flag} as
    fibd_odf_rec writes it with FIB_ODF_RAW_ODFMAX; in-place MAX over ranks: ONE 2-float all-reduce.  A NaN anywhere must win
    (Julia's maximum propagates NaN, gqi.jl:164): the flag is reduced with the maximum, and the consumers
    (fibd_qa_normalize_pair -- qa_normalize_device(raw=True) --, odfmax_value) turn {m, flag > 0} into NaN.
    raw=True (default, what odf_rec_sharded passes): the input IS the raw pair (odf_rec_device(raw_odfmax=True)) and the collective is
    the only device work.  raw=False: the plain form, whose element 0 may itself be NaN -- what a MAX all-reduce does with a NaN is
    undefined, so it is moved into the flag first (three small torch kernels).
    always: run the collective even in a one-rank group (tests: exercises the RCCL path on a 1-GPU box)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or always):
        if not raw:
            isn = torch.isnan(odfmax[0])
            odfmax[1] = torch.where(isn, torch.ones_like(odfmax[1]), odfmax[1])
            odfmax[0] = torch.where(isn, torch.full_like(odfmax[0], float("-inf")), odfmax[0])
        dist.all_reduce(odfmax, op=dist.ReduceOp.MAX, group=group)
    return odfmax


def odfmax_value(odfmax):
    """{max, NaN flag} -> maximum(mean(odf, dims=4)) as the reference computes it (NaN if any voxel's mean is NaN)"""
    import torch
    return torch.where(odfmax[1] > 0, torch.full_like(odfmax[0], float("nan")), odfmax[0])


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


def gather_objects(obj, group=None, always=False):
    import torch.distributed as dist
    if not (dist.is_initialized() and (dist.get_world_size(group) > 1 or always)):
        return [obj]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


# ---------------------------------------------------------------------------------------------
# sharded drivers (device tier)
# ---------------------------------------------------------------------------------------------
def any_unaligned(counts: Sequence[int]) -> bool:
    """FIB_ODF_SEPARATE_PEAKS contract (include/fibers_hip.h): when ONE volume is cut into pieces and any piece's voxel count is
    not a multiple of 4, EVERY piece runs find_peaks! as its own kernel (an unaligned piece cannot run the fused scan, and the
    two forms differ in two ODF rows at rounding level).  counts = every rank's slab voxel count (slab_bounds: the same on every
    rank, no collective)."""
    return any(int(c) % 4 != 0 for c in counts)


def odf_rec_sharded(plan, dwi_local, mask_local, group=None, stream=None, out=None, counts: Sequence[int] = None,
                    out_prezeroed: bool = False, always: bool = False):
    """gqi_rec / dsi_rec on this rank's z-slab + the global QA normalisation across ranks (gqi.jl:164-168): the slab's
    {odfmax, NaN flag} pair is all-reduced with MAX (one collective), the divisor never leaves the device.
    counts: the voxel counts of ALL ranks' slabs (e.g. nx*ny*(z1-z0) from slab_bounds); they decide, identically on every rank,
    whether every slab takes the separate peak finder (any_unaligned).  Without them the ranks agree through one extra 1-int
    all-reduce per call.  always: run the collectives in a one-rank group too (tests on a 1-GPU box)."""
    import contextlib
    import torch
    import torch.distributed as dist
    from .gqi import odf_rec_device, qa_normalize_device
    if counts is not None:
        sep = any_unaligned(counts)
    elif dist.is_initialized() and (dist.get_world_size(group) > 1 or always):
        fl = torch.tensor([int(mask_local.numel() % 4 != 0)], dtype=torch.int32, device=mask_local.device)
        dist.all_reduce(fl, op=dist.ReduceOp.MAX, group=group)
        sep = bool(fl.item())
    else:
        sep = False                                             # one piece: the library picks by the piece's own alignment
    out = odf_rec_device(plan, dwi_local, mask_local, out=out, normalize=False, stream=stream, separate_peaks=sep,
                         out_prezeroed=out_prezeroed, raw_odfmax=True)
    # the collective runs on the stream the kernels run on: it follows the kernel that writes out["odfmax"] and precedes the
    # normalisation in stream order, whatever torch's current stream is
    ts = _torch_stream(stream)
    with (torch.cuda.stream(ts) if ts is not None else contextlib.nullcontext()):
        allreduce_odfmax(out["odfmax"], group, always=always)
    qa_normalize_device(out["qa"], out["odfmax"], stream=stream, raw=True)   # {max, flag} -> NaN divisor if any rank saw a NaN mean
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
