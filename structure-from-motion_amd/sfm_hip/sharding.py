"""Hypothesis-space sharding across GPUs (one process per GPU, torch.distributed).

Hypotheses are independent given the (replicated) correspondences, so rank r scores the
contiguous id range shard_range(total, r, world).  The only exchange is the merge of the
per-rank winners: one all-reduce(max) over int64 keys (count << 32) | (0xFFFFFFFF - id), which
yields the largest count and, among equal counts, the lowest global id: the reference's
strict-'>' first-maximum rule (src/estimate_translation.py:73, src/ransac_algorithm.py:105).
With backend "nccl" the all-reduce is RCCL over xGMI; the payload is 8 bytes per problem, so
it is latency bound and all problems of a step are merged in one call.
"""
from __future__ import annotations


def shard_range(total, rank, world):
    """Contiguous [lo, hi) of `total` ids for `rank`; sizes differ by at most one."""
    q, r = divmod(int(total), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_key(count, index):
    if not (0 <= index <= 0xFFFFFFFF and 0 <= count < (1 << 31)):
        raise ValueError("count/index out of range for a packed key")
    return (int(count) << 32) | (0xFFFFFFFF - int(index))


def unpack_key(key):
    key = int(key)
    return key >> 32, 0xFFFFFFFF - (key & 0xFFFFFFFF)


def merge_best(keys):
    """In-place max-merge of an int64 tensor of packed keys over all ranks (no-op for 1 rank)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MAX)
    return keys
