"""Chunk-list sharding across the GPUs of a node: units are independent `Chunk::build` inputs, so the
data path has no collective; only the final gather of encoded buffers (host side) talks between ranks."""


def partition(cells_per_chunk, world):
    """Greedy longest-processing-time assignment balanced by cell count (the last time segment of a raster is
    short, dataset.rs:838).  Returns a list `owner[chunk] -> rank`; deterministic."""
    order = sorted(range(len(cells_per_chunk)), key=lambda i: (-cells_per_chunk[i], i))
    load = [0] * world
    owner = [0] * len(cells_per_chunk)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += cells_per_chunk[i]
    return owner


def my_chunks(owner, rank):
    return [i for i, r in enumerate(owner) if r == rank]


def gather_encoded(local, n_chunks, dist=None, dst=0):
    """local: {chunk_id: bytes} encoded on this rank.  Returns the full ordered list on rank `dst` (None elsewhere).
    Uses a host-side object gather (gloo or RCCL-backed process group); no device collective is involved."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local[i] for i in range(n_chunks)]
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, parts, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged = {}
    for p in parts:
        merged.update(p)
    return [merged[i] for i in range(n_chunks)]
