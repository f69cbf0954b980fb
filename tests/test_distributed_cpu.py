"""The N > 1 control plane of bench.py on CPU: world_size-2 gloo — channel sharding is disjoint and
complete, the barrier + MAX-over-ranks timing reduction works, and rank 0 alone reports."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from cu_sdr_collection_amd.sharding import broadcast_record, merge_acq_results, shard_channels, shard_prns
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    mine = shard_channels(13, world, rank)
    # every rank times its own (fake) work; the job time is the MAX over ranks
    elapsed = 0.5 + rank
    dist.barrier()
    t = torch.tensor([elapsed, float(len(mine))], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    # acquisition shards by PRN: each rank fills its own entries of acqResults, rank 0 merges
    import numpy as np
    from types import SimpleNamespace
    prns = list(range(1, 33))
    acq = SimpleNamespace(carrFreq=np.zeros(32), codePhase=np.zeros(32), peakMetric=np.zeros(32))
    for prn in shard_prns(prns, world, rank):
        acq.carrFreq[prn - 1], acq.codePhase[prn - 1], acq.peakMetric[prn - 1] = 1000.0 + prn, prn * 7, 3.0 + 0.1 * prn
    parts = [None] * world
    dist.all_gather_object(parts, acq)
    merged = merge_acq_results(parts)
    # the one exchange step: the rank that read the file broadcasts the raw record (RCCL over xGMI on the GPUs, gloo here)
    rec = torch.arange(-100, 100, dtype=torch.int8).repeat(50) if rank == 0 else None
    rec = broadcast_record(rec, src=0)
    q.put((rank, float(t[0]), gathered, merged.codePhase.tolist(), merged.peakMetric.tolist(), (int(rec.numel()), int(rec.to(torch.int64).sum()), int(rec[137]))))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_two_control_plane():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tmax, gathered, code_phase, metric, rec in res:
        assert rec == (10000, -5000, 37)
        assert code_phase == [prn * 7 for prn in range(1, 33)]
        assert metric == [3.0 + 0.1 * prn for prn in range(1, 33)]
        assert tmax == 1.5  # MAX over ranks
        assert sorted(gathered[0] + gathered[1]) == list(range(13))
        assert not set(gathered[0]) & set(gathered[1])


def _band_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from cu_sdr_collection_amd.sharding import band_ranks, distribute_band_records, shard_bands
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    plan = shard_bands({"L1": 5, "L5": 2, "E5b": 1}, world)        # every rank computes the same plan
    reads = []

    def read_record(band):                                          # stands for fopen + fread of the band's IF file
        reads.append(band)
        g = torch.Generator().manual_seed(sum(map(ord, band)))
        return torch.randint(-128, 128, (2000 + 100 * len(band),), dtype=torch.int8, generator=g)

    recs = distribute_band_records(plan, read_record)
    mine = sorted({b for b, _ in plan[rank]})
    q.put((rank, plan[rank], band_ranks(plan), reads, {b: (int(t.numel()), int(t.to(torch.int64).sum())) for b, t in recs.items()}, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_band_plan_end_to_end():
    """BASELINE config 5's hand-over on CPU: shard_bands -> one process group per band -> the band's first rank reads and
    broadcasts -> every rank ends up with exactly the records of the bands it tracks (a band spanning both ranks is read
    once and broadcast; single-rank bands are never sent anywhere)."""
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    def expect(band):
        g = torch.Generator().manual_seed(sum(map(ord, band)))
        t = torch.randint(-128, 128, (2000 + 100 * len(band),), dtype=torch.int8, generator=g)
        return int(t.numel()), int(t.to(torch.int64).sum())

    (r0, plan0, ranks0, reads0, recs0, mine0), (r1, plan1, ranks1, reads1, recs1, mine1) = res
    assert len(plan0) == len(plan1) == 4 and ranks0 == ranks1 == {"L1": [0, 1], "L5": [1], "E5b": [1]}
    assert mine0 == ["L1"] and mine1 == ["E5b", "L1", "L5"]
    assert reads0 == ["L1"] and sorted(reads1) == ["E5b", "L5"]      # L1 is read once, on rank 0, and travels
    assert set(recs0) == {"L1"} and set(recs1) == {"L1", "L5", "E5b"}
    for recs in (recs0, recs1):
        for b, v in recs.items():
            assert v == expect(b)


_RANK_SCRIPT = r'''
import datetime, os, sys, time
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1]
if mode == "die_before_rendezvous" and rank == 1:
    print("rank 1: gc_create failed (simulated)", file=sys.stderr, flush=True)
    sys.exit(7)
dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(os.environ["GC_RENDEZVOUS_TIMEOUT_S"])))
if mode == "die_in_collective" and rank == 1:
    print("rank 1: out of memory (simulated)", file=sys.stderr, flush=True)
    os._exit(9)
x = torch.ones(1)
if mode != "ok":
    time.sleep(0.5)
dist.all_reduce(x)            # rank 0 would wait here for the default timeout if nobody stopped it
if rank == 0:
    print('{"value": %d}' % int(x[0]), flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ["die_before_rendezvous", "die_in_collective"])
def test_launcher_ends_with_the_failing_ranks_report_when_a_rank_dies(mode, tmp_path):
    """bench.py --gpus N starts its ranks through sharding.launch_ranks: a rank that dies (before the rendezvous, or while the
    others sit in a collective) takes the rest down within seconds, and the caller learns which rank, its exit code and the
    tail of its stderr - instead of a hang until gloo's timeout with no result line."""
    import time
    sys.path.insert(0, ROOT)
    from cu_sdr_collection_amd.sharding import launch_ranks
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    lines = []
    t0 = time.time()
    out = launch_ranks([sys.executable, str(script), mode], 2, rendezvous_timeout_s=120.0, on_line=lambda r, l: lines.append((r, l)))
    took = time.time() - t0
    assert took < 60.0, took                                     # far below the 120-s rendezvous / collective timeout
    assert out["failed_rank"] == 1 and out["rc"] in (7, 9)
    assert "simulated" in " ".join(out["stderr_tail"][1])
    assert any(r == 1 and "simulated" in l for r, l in lines)     # relayed with the rank it came from


def test_launcher_returns_zero_and_rank_zero_prints_when_all_ranks_finish(tmp_path, capfd):
    sys.path.insert(0, ROOT)
    from cu_sdr_collection_amd.sharding import launch_ranks
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    out = launch_ranks([sys.executable, str(script), "ok"], 2, rendezvous_timeout_s=120.0)
    assert out["rc"] == 0 and out["failed_rank"] is None
    assert '{"value": 2}' in capfd.readouterr().out


def test_launcher_total_timeout_stops_ranks_that_never_finish(tmp_path):
    sys.path.insert(0, ROOT)
    from cu_sdr_collection_amd.sharding import launch_ranks
    script = tmp_path / "sleep.py"
    script.write_text("import time\ntime.sleep(600)\n")
    out = launch_ranks([sys.executable, str(script)], 2, total_timeout_s=2.0, grace_s=1.0, on_line=lambda r, l: None)
    assert out["rc"] == 124 and out["seconds"] < 30.0 and "no result" in out["reason"]
