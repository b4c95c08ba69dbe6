"""Frame sharding across the GPUs of one node: independent frames, one process per
GPU, no data-path collective; the only exchange is the all-gather of per-image keypoint counts
(RCCL over xGMI on GPUs; the same code runs on gloo/CPU tensors in the tests)."""


def shard_pairs(rank, world, pairs_per_rank):
    """Global stereo-pair indices owned by `rank` (weak scaling: every rank owns pairs_per_rank
    pairs of its own camera stream; streams are disjoint and contiguous in the global index)."""
    if not (0 <= rank < world) or pairs_per_rank < 0:
        raise ValueError("bad shard")
    start = rank * pairs_per_rank
    return range(start, start + pairs_per_rank)


def shard_round_robin(n_items, rank, world):
    """Strong-scaling partition of one stream: item i goes to rank i % world."""
    return range(rank, n_items, world)


def gather_counts(counts, world, dist=None, out=None):
    """All-gather of the per-image keypoint-count vector.  `counts` is an int32 tensor on the
    rank's device; returns the [world * len(counts)] tensor (identity for world == 1)."""
    if world == 1 or dist is None:
        return counts
    import torch
    if out is None:
        out = torch.empty(world * counts.numel(), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(out, counts)
    return out
