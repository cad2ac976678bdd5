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
