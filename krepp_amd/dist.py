"""Multi-GPU plumbing for the `dist` path: reads shard across ranks, the index is replicated.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).
There is no collective on the data path: the only communication is the load-time broadcast of
the index's flat buffers from rank 0 and whatever the caller does with results.
"""
from __future__ import annotations


def shard_bounds(n_items, rank, world):
    """Contiguous shard [lo, hi) of `n_items` for `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_reads(bases, offsets, rank, world):
    """The rank's slice of a packed read batch (offsets rebased to 0)."""
    n = len(offsets) - 1
    lo, hi = shard_bounds(n, rank, world)
    o = offsets[lo:hi + 1]
    return bases[int(o[0]):int(o[-1])], o - o[0], lo


def broadcast_blob(dist, blob, src=0):
    """Broadcast a small bytes object (the index descriptor)."""
    obj = [blob if dist.get_rank() == src else None]
    dist.broadcast_object_list(obj, src=src)
    return obj[0]


def broadcast_buffers(dist, tensors, src=0):
    """Broadcast every flat index buffer (uint8 tensors, on the rank's device) from `src`.
    Over RCCL each call is one ring broadcast across the xGMI links; buffers are few and large."""
    for t in tensors:
        if t.numel():
            dist.broadcast(t, src=src)


class DevPtr:
    """Expose a raw device pointer (from kr_index_export / kr_index_import) to torch through
    __cuda_array_interface__ so that torch.distributed can fill it in place."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def replicate_index(dist, torch, capi, dx_root, device, src=0):
    """Rank `src` holds an uploaded DeviceIndex; every other rank gets a byte-identical replica
    in its own HBM.  Returns the rank's DeviceIndex."""
    rank = dist.get_rank()
    desc, bufs = (dx_root.export() if rank == src else (None, None))
    desc = broadcast_blob(dist, desc, src)
    if rank == src:
        dx = dx_root
    else:
        dx, bufs = capi.DeviceIndex.import_empty(desc, device.index)
    tensors = [torch.as_tensor(DevPtr(p, nb), device=device) if nb else torch.empty(0, dtype=torch.uint8, device=device)
               for p, nb in bufs]
    broadcast_buffers(dist, tensors, src)
    torch.cuda.synchronize(device)
    return dx
