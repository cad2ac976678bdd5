"""GOP sharding across ranks (one process per GPU, no data-path collective).

Closed GOPs are independent units: rank r encodes the GOPs of gop_range(n_gops, world, r)
with its own encoder context seeded with the right frame numbers; the only "exchange" is gathering the
finished byte strings on rank 0 (host memory, torch.distributed object gather -- works on gloo and
nccl alike) and joining them with dsv1_concat_gops, which rewrites the packet links exactly as a
serial encode would have (dsv_encoder.c:171-192)."""


def gop_range(n_gops, world, rank):
    """contiguous, balanced partition: the first n_gops % world ranks take one extra GOP"""
    base, extra = divmod(n_gops, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_streams(local_streams, dist=None, dst=0):
    """local_streams: list of (gop_index, bytes) produced by this rank.  Returns on rank `dst` the list of
    per-GOP byte strings ordered by GOP index (None on the other ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [b for _, b in sorted(local_streams)]
    world = dist.get_world_size()
    gathered = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(local_streams, gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    allp = [p for part in gathered for p in part]
    return [b for _, b in sorted(allp)]


def pin_rank_to_cores(local_rank, local_world):
    """Give this rank a contiguous share of the cores the process may run on (os.sched_setaffinity) -- call it BEFORE
    anything touches the GPU, so that the runtime's helper threads, the pinned staging buffers (first touch) and the
    session layer's worker threads (dsv1_par_for sizes itself from the affinity mask) all stay on that share.
    Returns the list of cores taken (all allowed cores when there is nothing to split)."""
    import os
    cores = sorted(os.sched_getaffinity(0))
    if local_world <= 1 or len(cores) < local_world:
        return cores
    per = len(cores) // local_world
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine
