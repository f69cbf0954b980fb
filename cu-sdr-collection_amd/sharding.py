"""Channel sharding for multi-GPU runs: tracking channels are independent
(tracking.m:133 loop body shares nothing but the read-only IF record), so the partition is by
channel, one process per GPU, no data-path collective."""
from __future__ import annotations


def shard_channels(n_channels: int, world_size: int, rank: int) -> list[int]:
    """Contiguous, balanced split of channel indices 0..n_channels-1 over `world_size` ranks:
    the first (n_channels % world_size) ranks take one extra channel."""
    if world_size < 1 or not 0 <= rank < world_size or n_channels < 0:
        raise ValueError("bad sharding arguments")
    base, extra = divmod(n_channels, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))
