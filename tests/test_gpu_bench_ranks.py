"""bench.py's N > 1 paths end to end on ONE GPU (GC_BENCH_DEVICE=0 puts every rank on device 0; the eight-GPU run is the driver's).

BASELINE configs[4] at its stated shape - 64 channels, all twelve signals, eight band records, eight ranks - and the sharded
GPS L1 C/A line: band plan -> the band's first rank holds the record -> sharding.distribute_band_records / broadcast_record ->
gc_attach_if -> one tracking job per package -> replay.  RCCL needs one GPU per rank, so on this box the records cross through
the hosts (gloo); on the driver's 8-GPU node the same code takes backend nccl (Ranks.rccl)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=1500, env_extra=None, drop=()):
    """Runs bench.py, checks the printed line (ONE line, < 4 KB, the contract's keys) and returns the long form from --detail."""
    import tempfile
    env = dict(os.environ, GC_BENCH_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", *drop):
        env.pop(k, None)
    env.update(env_extra or {})
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--detail", detail], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0], r.stdout[-3000:]
        line = json.loads(lines[0])
        assert len(lines[0]) < 4096 and line["roofline"]["frac"] > 0 and line["value"] > 0 and "workload" in line["config"], lines[0]
        full = json.load(open(detail))
    assert full["value"] == line["value"] and full["ms_per_step"] == line["ms_per_step"] and full["roofline"]["frac"] == line["roofline"]["frac"]
    return full


def test_config5_sixty_four_channels_of_twelve_signals_on_eight_ranks():
    res = _bench("--config", "mix", "--gpus", "8", "--mix-seconds", "2", "--steps", "2", "--warmup", "1", "--spot-check")
    assert res["n_gpus"] == 8 and res["channels"] == 64 and len(res["ranks"]) == 8
    assert res["signals"] == sorted(["GPS_L1CA", "GAL_E1C", "BDS_B1C_NB", "GPS_L5C", "GAL_E5a", "BDS_B2a", "BDS_B1I", "GAL_E5b", "BDS_B3I",
                                     "GPS_L2C", "GLO_GL1", "GLO_GL2"])
    assert set(res["config"]["band_reader_ranks"]) == {"L1", "L5", "B1I", "E5b", "B3I", "L2", "GLO_L1", "GLO_L2"}
    assert res["channels_locked"] == 64 and res["channels_locked_device_loop"] == 64
    travelled = set()
    for r in res["ranks"]:
        assert sum(j["channels"] for j in r["jobs"]) == 8                       # every rank tracks eight channels
        assert r["handover"] is not None
        travelled |= set(r["handover"]["bands_received_or_sent_here"])
        for j in r["jobs"]:
            assert j["channels_locked"] == j["channels"], (r["rank"], j)
            assert j["oracle_spot_check_max_dev_rel_sum_abs_x"] < 2e-6, (r["rank"], j)     # the float64 oracle at identical descriptors
            assert j["replay_vs_closed_loop_max_dev"] < 2e-5, (r["rank"], j)               # batched replay = the loop's own records
    # the bands whose channels span two ranks crossed the wire; single-rank bands never did
    spanning = {b for b, ranks in res["config"]["band_reader_ranks"].items() if len(ranks) > 1}
    assert travelled == spanning and {"L1", "L5"} <= spanning
    assert res["value"] > 0 and res["roofline"]["frac"] > 0


def test_sharded_l1ca_line_hands_the_record_over_and_shards_acquisition_by_prn():
    res = _bench("--gpus", "2", "--seconds", "4", "--steps", "2", "--warmup", "1", "--no-cpu")
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    h = res["handover"]
    assert h["bytes"] >= 2 * 4 * 18000000 and h["seconds"] > 0
    assert [r["channels_locked"] for r in res["ranks"]] == [12, 12]
    assert set(res["ranks"][0]["prns"]) != set(res["ranks"][1]["prns"])         # the ranks track different channels of the one record
    for r in res["ranks"]:
        assert r["replay_vs_closed_loop_max_dev"] < 2e-5
    a = res["acquisition"]
    assert a["prns_per_rank"] == 16 and a["all_scene_prns_found"] and a["code_phases_within_two_samples_of_the_scene"]
    # A/B: without the hand-over every rank synthesises its own copy - same scene, same results
    res2 = _bench("--gpus", "2", "--seconds", "4", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-handover")
    assert "handover" not in res2 and [r["channels_locked"] for r in res2["ranks"]] == [12, 12]
    assert res2["acquisition"]["acquired"] == a["acquired"]


def test_rccl_process_group_carries_the_hand_over_at_world_size_one():
    """The process group the driver's 8-GPU run uses - "cpu:gloo,cuda:nccl", RCCL probe, broadcast of the GPU tensor, gc_attach_if -
    with the one rank a 1-GPU box allows (GC_BENCH_FORCE_DIST): the record's hand-over runs over backend nccl."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    res = _bench("--gpus", "1", "--config", "l1ca", "--seconds", "4", "--steps", "2", "--warmup", "1", "--no-cpu", timeout=900,
                 env_extra={"GC_BENCH_FORCE_DIST": "1", "MASTER_PORT": port}, drop=("GC_BENCH_DEVICE",))
    assert res["handover"]["backend"].startswith("nccl"), res["handover"]
    assert res["closed_loop"]["channels_locked"] == 12 and res["replay_vs_closed_loop_max_dev"] < 2e-5


@pytest.mark.gpu
def test_more_ranks_than_devices_fails_fast_with_the_error_line():
    """VERDICT r5 #8a: `bench.py --gpus 8` on a box with fewer GPUs ends in seconds with the ONE JSON line (`error`, value null) and a
    non-zero status - without an external launcher and as rank 0 of one (torch.distributed.run sets RANK / WORLD_SIZE), before torch
    is imported or a rank is started."""
    import json, subprocess, sys, time
    import cu_sdr_collection_amd as P
    seen = P.device_count()
    assert seen >= 1
    n = seen + 7
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("GC_BENCH_DEVICE", "RANK", "LOCAL_RANK", "WORLD_SIZE")}
    for extra in ({}, {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"}):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], env={**env, **extra},
                           capture_output=True, text=True, timeout=120)
        dt = time.perf_counter() - t0
        assert r.returncode != 0 and dt < 10.0, (r.returncode, dt, r.stderr[-500:])
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert d["value"] is None and d["n_gpus"] == n and d["n_devices_seen"] == seen and "device" in d["error"]
