"""Channel sharding for multi-GPU runs: tracking channels are independent (tracking.m:133 loop body shares nothing
but the read-only IF record), so the partition is by channel, one process per GPU.  The ONE exchange step of the data path is the
hand-over of a band's raw record from the rank that read it to the other ranks that track channels of that band (broadcast_record /
distribute_band_records: RCCL broadcast of the GPU tensor, one sub-group per band); nothing is exchanged per epoch."""
from __future__ import annotations


# bench.py closed_loop_sweep (MI355X, round 4, 5-s records): microseconds per epoch of the HOST-closed loop - the mode north_star
# names - against the channels on ONE GPU; an epoch is 1 ms of signal, so x real time = 1000 / value.  Up to the knee an epoch is one
# channel's own latency chain (descriptor over PCIe, correlate, all-gather of the team's partial sums, records back) whatever the
# number of channels, so more GPUs buy nothing; beyond it the teams shrink (the persistent grid must stay resident) and the single
# closer thread fills up, and the time grows - sub-linearly: 8x the channels cost 2.4x the time.
CLOSED_LOOP_US_PER_EPOCH = {
    "GPS_L1CA": {12: 6.7, 24: 7.6, 48: 8.6, 96: 13.1, 192: 16.3},      # 18 Msps, host-closed (round 6: gpurun_out/r06*/bench.json); device-closed: 5.0, 7.0, 6.3, 7.5, 9.9
    "GPS_L5C": {16: 9.7, 32: 10.1, 64: 13.1},                           # 50 Msps, data + pilot; device-closed: 8.7, 8.7, 11.1
}
# the largest measured count whose epoch time is within 1.25x of the smallest count's (`knee_channels` of the sweep); signals that
# were not swept take the entry of their kernel class: "fast" (<= 1 table transition per 16-sample chunk) or "lane"
CLOSED_LOOP_KNEE = {"GPS_L1CA": 24, "GPS_L5C": 32, "fast": 24, "lane": 32}
# kernel class of the closed loop per package (gc_block_lowrate_level at the package's default front end): table entries per sample
# step * R - at most one table transition per 8-sample chunk (7 * step * R < 1) runs the fast kernel: the 0.511 / 1.023 / 2.046-Mcps
# codes at 12 - 18 Msps, BOC(1,1) half-chip tables included; the 10.23-Mcps codes and the three-arm channels with a derived BOC(6,1)
# arm run the lane kernel
SIGNAL_CLASS = {"GPS_L1CA": "fast", "GLO_GL1": "fast", "GLO_GL2": "fast", "GPS_L2C": "fast", "BDS_B1I": "fast", "GAL_E1C": "fast", "BDS_B1C_NB": "fast",
                "GAL_E1C_CBOC": "lane", "BDS_B1C_WB": "lane", "GPS_L5C": "lane", "GAL_E5a": "lane", "GAL_E5b": "lane", "BDS_B2a": "lane", "BDS_B3I": "lane"}

def expected_us_per_epoch(n_channels: int, signal: str = "GPS_L1CA") -> float:
    """Host-closed loop of `n_channels` on one GPU: the sweep's measurement, interpolated linearly in the channel count (held flat
    below the first point, extended with the last segment's slope above the last)."""
    cls = signal if signal in CLOSED_LOOP_US_PER_EPOCH else ("GPS_L1CA" if SIGNAL_CLASS.get(signal, "lane") == "fast" else "GPS_L5C")
    pts = sorted(CLOSED_LOOP_US_PER_EPOCH[cls].items())
    if n_channels <= pts[0][0]:
        return pts[0][1]
    for (n0, t0), (n1, t1) in zip(pts, pts[1:]):
        if n_channels <= n1:
            return t0 + (t1 - t0) * (n_channels - n0) / (n1 - n0)
    (n0, t0), (n1, t1) = pts[-2], pts[-1]
    return t1 + (t1 - t0) * (n_channels - n1) / (n1 - n0)


def recommended_world_size(n_channels: int, signal: str = "GPS_L1CA", max_gpus: int = 8, min_x_realtime: float | None = None,
                           epoch_ms: float = 1.0) -> int:
    """north_star: "shard across the 8 GPUs of one node ... only when the channel count warrants it" (the reference's channel loop,
    tracking.m:133, is serial).  Default: 1 while `n_channels` is at or below the signal's knee (CLOSED_LOOP_KNEE), then the fewest
    GPUs that bring every rank's share back under it.  With `min_x_realtime`: the fewest GPUs whose share of the channels the
    sweep's table (expected_us_per_epoch) runs at that many times real time or faster (`epoch_ms`: the package's block length)."""
    if n_channels < 0 or max_gpus < 1:
        raise ValueError("bad arguments")
    if min_x_realtime is not None:
        for n in range(1, max_gpus + 1):
            if epoch_ms * 1e3 / expected_us_per_epoch(-(-n_channels // n), signal) >= min_x_realtime:
                return n
        return max_gpus
    knee = CLOSED_LOOP_KNEE.get(signal)
    if knee is None:
        knee = CLOSED_LOOP_KNEE[SIGNAL_CLASS.get(signal, "lane")]
    return max(1, min(max_gpus, -(-n_channels // knee)))


def shard_channels(n_channels: int, world_size: int, rank: int) -> list[int]:
    """Contiguous, balanced split of channel indices 0..n_channels-1 over `world_size` ranks:
    the first (n_channels % world_size) ranks take one extra channel."""
    if world_size < 1 or not 0 <= rank < world_size or n_channels < 0:
        raise ValueError("bad sharding arguments")
    base, extra = divmod(n_channels, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def shard_bands(band_channels: dict, world_size: int) -> list:
    """BASELINE config 5 (all-constellation mix, 64 channels on 8 GPUs): channels of several IF
    records (bands) placed so that every rank holds the same number of channels (+-1: the shares of `shard_channels`) while each
    band's record lives on as few GPUs as possible -- a rank needs the record of every band it tracks, and a band shared by k
    ranks costs k - 1 peer copies (`broadcast_record`) and k x its size in HBM.  `band_channels`: {band: number of channels}.
    Returns plan[rank] = [(band, channel index within the band), ...].

    Bands are placed largest first (ties by name).  A band takes whole EMPTY ranks for as many full shares as it has, and its
    remainder goes to the rank with the least free room that still holds it (best fit); only when no rank does is it cut, over the
    ranks with the most room.  A band of n channels therefore sits on ceil(n / share) ranks whenever the ranks' room allows it
    (config 5: the 18-channel L1 band on 3 GPUs, the 16-channel L5 band on 2, four of the six 5-channel bands on 1)."""
    if world_size < 1 or any(n < 0 for n in band_channels.values()):
        raise ValueError("bad sharding arguments")
    total = sum(band_channels.values())
    free = [len(shard_channels(total, world_size, r)) for r in range(world_size)]
    cap = list(free)
    plan = [[] for _ in range(world_size)]

    def put(band, first, count, rank):
        plan[rank].extend((band, first + i) for i in range(count))
        free[rank] -= count

    for band, n in sorted(band_channels.items(), key=lambda kv: (-kv[1], kv[0])):
        at = 0
        for r in range(world_size):                       # full shares on ranks that are still empty
            if n - at >= cap[r] > 0 and free[r] == cap[r]:
                put(band, at, cap[r], r)
                at += cap[r]
        while at < n:
            rest = n - at
            fits = [r for r in range(world_size) if free[r] >= rest]
            if fits:                                      # best fit: the least room that still holds the remainder
                r = min(fits, key=lambda q: (free[q], q))
                put(band, at, rest, r)
                at = n
            else:                                         # no rank holds it: the most room first
                r = max(range(world_size), key=lambda q: (free[q], -q))
                if free[r] == 0:
                    raise AssertionError("shard_bands: the shares do not add up")
                take = free[r]
                put(band, at, take, r)
                at += take
    return plan


def band_ranks(plan) -> dict:
    """{band: sorted ranks that track at least one of its channels}; the first rank of each list reads the band's file and is
    the `src` of its `broadcast_record` (process group = that list of ranks)."""
    out: dict = {}
    for r, items in enumerate(plan):
        for b, _ in items:
            if r not in out.setdefault(b, []):
                out[b].append(r)
    return out


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


def broadcast_record(record, src: int = 0, device=None, group=None, via_host: bool = False):
    """The one exchange step of the sharded path: the rank that read the IF file hands the raw record to the
    ranks that track other channels of the same band.  `record`: a 1-D torch int8 / int16 tensor on rank `src` (ignored
    elsewhere).  `src` is a GLOBAL rank; `group` the process group of the ranks that share the band (None: every rank) - only
    its members call this function.  With tensors on the GPUs and the group's backend "nccl" this is one RCCL broadcast over
    xGMI (288 GB of HBM per GPU: the whole record in one piece, no chunking); on CPU tensors (gloo) the same call is the test
    double.  Returns the record on every member; hand it to the engine without a copy:
        t = broadcast_record(t, device=f"cuda:{local_rank}");  engine.attach_if(t.data_ptr(), t.numel() // 2)
    Process order matters when torch drives the GPU in the same process as the engine: torch ships its own copy of the HIP
    runtime, so initialise torch's CUDA side (torch.cuda.set_device) BEFORE the first Engine() -- with the engine first, torch
    finds no device (tests/test_gpu_correlator.py::test_broadcast_record_is_adopted_without_a_copy runs in that order).
    `via_host`: the record crosses as a CPU tensor (the group's CPU backend) and is uploaded to `device` on arrival - for ranks
    that share ONE GPU (RCCL refuses two ranks on a device: functional checks on a 1-GPU box) or a group without a GPU backend.
    """
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    meta = torch.zeros(2, dtype=torch.int64)            # a CPU tensor: the group's CPU backend carries it whatever its GPU backend is
    if rank == src:
        if record.dim() != 1 or record.dtype not in (torch.int8, torch.int16):
            raise ValueError("broadcast_record: a 1-D int8 or int16 tensor is expected")
        meta[0], meta[1] = int(record.numel()), 1 if record.dtype == torch.int8 else 2
    try:
        dist.broadcast(meta, src=src, group=group)
    except RuntimeError:                                 # a group with a GPU backend only ("nccl"): the same on every member
        if device is None:
            raise
        m = meta.to(device)
        dist.broadcast(m, src=src, group=group)
        meta = m.cpu()
    n = int(meta[0])
    dtype = torch.int8 if int(meta[1]) == 1 else torch.int16
    wire = "cpu" if (via_host or device is None) else device
    if rank == src:
        home = record if device is None else record.to(device)       # what this rank keeps (no copy when it is there already)
        t = home if wire != "cpu" else record.to("cpu")
    else:
        home = None
        t = torch.empty(n, dtype=dtype, device=wire)
    dist.broadcast(t, src=src, group=group)
    if rank == src:
        return home
    return t if device is None else t.to(device)


def distribute_band_records(plan, read_record, device=None, via_host: bool = False, timings: dict | None = None):
    """BASELINE config 5, the hand-over end to end: `plan` = shard_bands(...) (identical on every rank).  For every band the
    first rank that tracks one of its channels reads the file (`read_record(band)` -> 1-D int8 / int16 tensor) and broadcasts
    it to the other ranks of the band - one process group per band, so ranks that do not track the band neither take part
    nor spend HBM on it.  Returns {band: tensor} for the bands of THIS rank.  Every rank must call it (torch.distributed
    creates groups collectively, in the same order everywhere).  `timings`, if given, receives {band: (seconds of the
    broadcast as this rank saw it, bytes)} for the bands that travelled."""
    import time
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    ranks = band_ranks(plan)
    groups = {}
    for band in sorted(ranks):                          # new_group is collective over the WORLD: same order on every rank
        groups[band] = dist.new_group(ranks=ranks[band]) if len(ranks[band]) > 1 else None
    out = {}
    for band in sorted(ranks):
        members = ranks[band]
        if rank not in members:
            continue
        src = members[0]
        rec = read_record(band) if rank == src else None
        if len(members) == 1:
            out[band] = rec if device is None else rec.to(device)
        else:
            t0 = time.perf_counter()
            out[band] = broadcast_record(rec, src=src, device=device, group=groups[band], via_host=via_host)
            if out[band].is_cuda:
                torch.cuda.synchronize(out[band].device)
            if timings is not None:
                timings[band] = (time.perf_counter() - t0, int(out[band].numel()) * out[band].element_size())
    return out


def launch_ranks(argv, n: int, env_extra=None, rendezvous_timeout_s: float = 240.0, grace_s: float = 5.0, total_timeout_s: float | None = None,
                 on_line=None) -> dict:
    """One process per GPU without an external launcher, unable to hang: starts `argv` n times (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT set, GC_RENDEZVOUS_TIMEOUT_S = the init_process_group timeout the ranks should use), polls
    ALL children, and on the first non-zero exit (or when `total_timeout_s` passes) terminates the rest - a rank that dies at
    gc_create / init_process_group would otherwise leave the others inside a collective until its timeout.  Rank 0's stdout is this
    process's stdout (the result line), every rank's stderr (and the other ranks' stdout) is relayed line by line with a
    "[rank r] " prefix and its tail kept.
    Returns {"rc": 0 | the failing rank's code, "failed_rank": r | None, "reason": str | None, "stderr_tail": {rank: last lines},
    "seconds": wall time}."""
    import collections
    import os
    import socket
    import subprocess
    import sys
    import threading
    import time
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tails = {r: collections.deque(maxlen=30) for r in range(n)}
    procs, pumps = [], []

    def pump(r, stream):
        for raw in iter(stream.readline, b""):
            line = raw.decode("utf-8", "replace").rstrip("\n")
            tails[r].append(line)
            if on_line is not None:
                on_line(r, line)
            else:
                print(f"[rank {r}] {line}", file=sys.stderr, flush=True)
        stream.close()

    t0 = time.time()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GC_RENDEZVOUS_TIMEOUT_S=str(int(rendezvous_timeout_s)), GC_RANK_LAUNCHER="launch_ranks", **(env_extra or {}))
        p = subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        procs.append(p)
        for stream in ([p.stderr] if r == 0 else [p.stderr, p.stdout]):
            th = threading.Thread(target=pump, args=(r, stream), daemon=True)
            th.start()
            pumps.append(th)
    failed, reason, rc = None, None, 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and failed is None:
                failed, rc, reason = r, (code if code > 0 else 128 - code), f"rank {r} exited with code {code}"
        if failed is None and total_timeout_s is not None and time.time() - t0 > total_timeout_s and alive:
            failed, rc, reason = min(alive), 124, f"no result after {total_timeout_s:.0f} s"
        if failed is not None and alive:
            for r in alive:                                   # the survivors sit in a collective that will never complete
                try:
                    os.killpg(procs[r].pid, 15)               # each child is its own session: exact process group, no pattern
                except (ProcessLookupError, PermissionError):
                    pass
            t1 = time.time()
            while any(procs[r].poll() is None for r in alive) and time.time() - t1 < grace_s:
                time.sleep(0.05)
            for r in alive:
                if procs[r].poll() is None:
                    try:
                        os.killpg(procs[r].pid, 9)
                    except (ProcessLookupError, PermissionError):
                        pass
                procs[r].wait()
            alive.clear()
        if alive:
            time.sleep(0.05)
    for th in pumps:
        th.join(timeout=2.0)
    return {"rc": rc, "failed_rank": failed, "reason": reason, "stderr_tail": {r: list(t) for r, t in tails.items()}, "seconds": time.time() - t0}
