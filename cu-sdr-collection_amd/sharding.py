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


def shard_prns(prn_list, world_size: int, rank: int) -> list:
    """Acquisition shards by PRN: every rank computes the PRN-independent signal spectra itself (they are
    hoisted out of the PRN loop and cost one forward FFT pass) and searches its round-robin share of the list, so ranks
    finish together whatever the list order; no collective on the data path."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad sharding arguments")
    return [p for i, p in enumerate(prn_list) if i % world_size == rank]


def merge_acq_results(per_rank):
    """Combine the acqResults structs of the ranks (each filled only at its own PRNs, zeros elsewhere): element-wise sum
    is exact because the shares are disjoint.  per_rank: list of objects with carrFreq / codePhase / peakMetric arrays."""
    import numpy as np
    from types import SimpleNamespace
    out = SimpleNamespace()
    for f in vars(per_rank[0]):
        setattr(out, f, np.sum([np.asarray(getattr(r, f)) for r in per_rank], axis=0))
    return out
