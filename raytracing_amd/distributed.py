"""Multi-GPU sharding of one frame: each rank traces a disjoint range of the sqrt_spp^2 strata for every pixel and the
float accumulation buffers are summed onto rank 0 with one collective (RCCL over xGMI on GPUs; gloo in CPU tests).

The reference has no distributed code; this follows SURVEY.md 8(e): linear stratum index s = s_j*sqrt_spp + s_i
(camera.h:176-181), RNG keyed by (seed, pixel, s) so the summed image is independent of the number of ranks; the sum is
left unnormalised and scaled by 1/sqrt_spp^2 after the reduce (camera.h:184,215)."""
from __future__ import annotations


def split_range(n_strata: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, disjoint, exhaustive stratum ranges (sizes differ by at most one: 961 = 8*120 + 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * n_strata) // world, ((rank + 1) * n_strata) // world


def frame_shards(n_frames: int, world: int, rank: int) -> list[int]:
    """Animation sequences shard by frame: frame f -> rank f mod world, no collective."""
    return list(range(rank, n_frames, world))


def reduce_accum(accum, dst: int = 0):
    """Sum the ranks' accumulation buffers onto `dst` (torch.distributed: nccl == RCCL on ROCm)."""
    import torch.distributed as dist
    if dist.is_initialized():  # (a one-rank group reduces onto itself: the same call path, nothing moves)
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
