#!/usr/bin/env python3
"""bench.py — IF Msamples/s through the tracking correlators; x real-time @ 12-ch GPS L1 C/A.

Workload (BASELINE.json configs[1]): 12 GPS L1 C/A channels, 1-ms coherent
E/P/L correlators, 60 s of synthetic int8 I/Q IF at 18 Msps (2.16 GB resident in HBM), C/N0 45 dB-Hz,
Doppler U(-5,5) kHz, seed 20241008+2.

  1. the record is synthesised in HBM (csrc/synth.hip);
  2. closed-loop tracking (gc_track: discriminators and loop filters on the host, the correlator in a persistent
     host-fed kernel) produces the per-epoch state -> reported as x real-time (closed loop);
  3. a "step" = ONE batched replay pass of the hot path over the whole record: all
     channels x epochs blocks (720 000) in one launch, descriptors and IF resident in HBM.
     `value` = channel-samples through the correlators per second (whole job, all ranks).

N > 1 (one rank per GPU, launched by torch.distributed.run): channels shard across GPUs with no
data-path collective — every rank tracks its own 12 channels on its own copy of the record — so
scaling is "weak"; the control plane (barrier + max over ranks) is torch.distributed/gloo.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def _spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without torch.distributed.run: start N copies of this script, one rank per GPU
    (LOCAL_RANK selects the device; GC_BENCH_DEVICE pins every rank to one device on a 1-GPU box), rendezvous on
    127.0.0.1.  Rank 0 prints the result line; the other ranks' stdout goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=60.0, help="length of the IF record")
    ap.add_argument("--channels", type=int, default=12)
    ap.add_argument("--cpu-epochs", type=int, default=4000, help="epochs per channel timed on the CPU baseline (4000: ~13 s of one core)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus))     # no external launcher: one process per GPU, started here
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         "they must agree (n_gpus in the result line is the number of ranks that ran)")
    dist = None
    if world > 1 or os.environ.get("GC_BENCH_FORCE_DIST"):
        import torch.distributed as dist  # control plane only: gloo on CPU tensors
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import cu_sdr_collection_amd as P
    from cu_sdr_collection_amd import _lib as L
    from cu_sdr_collection_amd.receiver import track_params

    S = P.initSettings()
    fs = S.samplingFreq
    nch = args.channels
    n_samples = int(round(args.seconds * fs))
    n_epochs = int(args.seconds * 1000) - 2
    S.msToProcess = n_epochs
    S.numberOfChannels = nch

    # one GPU per rank; GC_BENCH_DEVICE pins every rank to one device (functional test of the N > 1 path on a 1-GPU box)
    eng = P.Engine(int(os.environ.get("GC_BENCH_DEVICE", local_rank)))
    dev_name, cus = eng.device_info()
    sats = P.synth.scene(nch, 20241008 + 2, fs)
    t0 = time.time()
    P.synth.generate_if_gpu(eng, sats, n_samples, fs, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023,
                            seed=20241008 + 2)
    t_synth = time.time() - t0
    eng.set_sampling_freq(fs)

    # channel table as preRun would hand it over (truth + a 3 Hz acquisition residual)
    inits = []
    for i, s in enumerate(sats):
        eng.set_channel(i, [P.codes.padded_table(P.codes.generateCAcode(s.prn))])
        inits.append(L.gc_channel_init(channel=i, prn=s.prn, acquired_freq=S.IF + s.doppler + 3.0,
                                       code_freq=S.codeFreqBasis, code_phase=int(np.ceil(s.code_phase_samples)) + 1))

    # ---- closed loop: produces the state every replay block needs -------------------------------
    p = track_params(S)
    t0 = time.time()
    fields, done, st = eng.track(p, inits)
    t_closed = time.time() - t0
    if st != 0 or int(done.min()) != n_epochs:
        raise RuntimeError(f"closed-loop tracking stopped early: status {st}, epochs {done}")
    locked = np.mean(np.abs(fields["I_P"][:, 1000:]), axis=1) > 3 * np.mean(np.abs(fields["Q_P"][:, 1000:]), axis=1)
    blks = np.ceil((S.codeLength - fields["remCodePhase"]) / (fields["codeFreq"] / fs)).astype(np.int64)
    chan_samples = int(blks.sum())
    closed_msps = chan_samples / t_closed / 1e6
    # the same loop closed on the device (one persistent cooperative launch, include/gnsscorr.h gc_track_device)
    t0 = time.time()
    dfields, ddone, dst = eng.track(p, inits, device_loop=True)
    t_dev = time.time() - t0
    dev_loop = None
    if dst == 0 and int(ddone.min()) == n_epochs:
        dev_loop = {"corr_msps": round(chan_samples / t_dev / 1e6, 1), "x_realtime": round(chan_samples / t_dev / 1e6 / nch / (fs / 1e6), 2),
                    "us_per_epoch": round(t_dev / n_epochs * 1e6, 2),
                    "same_block_geometry_as_host_loop": bool(np.array_equal(dfields["absoluteSample"], fields["absoluteSample"])),
                    # two closed loops part ways by one sample of block boundary at a knife edge of ceil() sooner or later
                    # (DESIGN.md 4.3b); until then they cut identical blocks
                    "epochs_with_the_host_loops_block_geometry": int(np.argmax(np.any(dfields["absoluteSample"] != fields["absoluteSample"], axis=0)))
                    if np.any(dfields["absoluteSample"] != fields["absoluteSample"]) else int(n_epochs),
                    "max_carr_freq_dev_hz": float(np.max(np.abs(dfields["carrFreq"] - fields["carrFreq"])))}

    # ---- replay descriptors, epoch-major so the channels of one epoch sit next to each other ----
    nb = nch * n_epochs
    blocks = eng.make_blocks(nb)
    # fill through a structured numpy view (720k ctypes attribute writes would take seconds)
    dt = np.dtype([("channel", "<i4"), ("blksize", "<i4"), ("first_sample", "<i8"), ("rem_code_phase", "<f8"),
                   ("code_phase_step", "<f8"), ("el_spacing", "<f8"), ("carr_freq", "<f8"),
                   ("rem_carr_phase", "<f8"), ("table_offset", "<i4", (3,)), ("reserved", "<i4")])
    assert dt.itemsize == 72
    view = np.frombuffer(blocks, dtype=dt)
    for k in range(nch):
        sl = slice(k, nb, nch)
        view["channel"][sl] = k
        view["blksize"][sl] = blks[k]
        view["first_sample"][sl] = fields["absoluteSample"][k].astype(np.int64)
        view["rem_code_phase"][sl] = fields["remCodePhase"][k]
        view["code_phase_step"][sl] = fields["codeFreq"][k] / fs
        view["el_spacing"][sl] = S.dllCorrelatorSpacing
        view["carr_freq"][sl] = fields["carrFreq"][k]
        view["rem_carr_phase"][sl] = fields["remCarrPhase"][k]
    eng.replay_prepare(blocks)

    # ---- warm-up, then exactly K timed steps -----------------------------------------------------
    for _ in range(args.warmup):
        eng.replay_launch()
    eng.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(args.steps):
        eng.replay_launch()
    kernel_ms_total = eng.timer_stop()  # hipEvents on the launch stream; also synchronises
    eng.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, kernel_ms_total], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_total = float(t[0]), float(t[1])

    # replay must reproduce the closed-loop outputs (same kernel, different split count)
    out = eng.replay_fetch()[:, 0, :]
    rec = np.stack([fields[f].T.reshape(-1) for f in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")], axis=1)
    scale = 2.0 * 18000 * 28.0
    replay_dev = float(np.max(np.abs(out - rec)) / scale)

    ms_per_step = elapsed * 1e3 / args.steps
    total_chan_samples = chan_samples * world
    value = total_chan_samples * args.steps / elapsed / 1e6
    kernel_ms = kernel_ms_total / args.steps
    algo_bytes = 2.0 * chan_samples  # int8 I/Q: 2 bytes per channel-sample
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9

    # HBM bytes per launch of the replay kernel, measured offline with rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE on this exact command (profiles/r01/traffic.json, gfx950 x2 read correction applied)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01", "traffic.json")))
        if tj.get("blocks_per_launch") == nb:
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    result = {
        "metric": "IF Msamples/s through tracking correlators; x real-time @ 12-ch GPS L1 C/A",
        "value": round(value, 1),
        "unit": "Msamples/s (channel-samples, all GPUs)",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 accumulate over int8 I/Q samples; f64/64-bit fixed-point code+carrier phase",
        "data": "synthetic",
        "config": {"workload": f"GPS L1 C/A, {nch} channels/GPU, 1-ms E/P/L correlators, {args.seconds:g} s of int8 I/Q IF at 18 Msps "
                               f"({n_samples * 2 / 1e9:.2f} GB in HBM), batched replay of {nb} blocks per step",
                   "channels_per_gpu": nch, "epochs": n_epochs, "blocks_per_step": nb,
                   "parallelism": f"channels sharded over {world} GPU(s), no data-path collective"},
        "if_msps_per_gpu": round(value / world / nch, 1),
        "x_realtime_replay": round(value / world / nch / (fs / 1e6), 1),
        "closed_loop": {"corr_msps": round(closed_msps, 1), "x_realtime": round(closed_msps / nch / (fs / 1e6), 2),
                        "us_per_epoch": round(t_closed / n_epochs * 1e6, 2), "channels_locked": int(locked.sum())},
        "closed_loop_device": dev_loop,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "kernel": "corr_epl_fast_kernel<ARMS=1, I8_IQ, SPL=16>", "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes_per_launch": algo_bytes},
        "replay_vs_closed_loop_max_dev": replay_dev,
        "device": dev_name, "compute_units": cus, "synth_s": round(t_synth, 2),
    }

    # ---- acquisition (the other half of the hot path), reference defaults: 32 PRNs x 29 bins x 20 ms ----
    from cu_sdr_collection_amd.receiver import acquisition as gpu_acquisition
    Sa = P.initSettings()
    gpu_acquisition(eng, Sa)  # warm-up (plans, twiddles, scratch)
    t0 = time.perf_counter()
    acq = gpu_acquisition(eng, Sa)
    t_acq = time.perf_counter() - t0
    found = sorted(int(i) + 1 for i in np.nonzero(acq.carrFreq)[0])
    result["acquisition"] = {"seconds": round(t_acq, 4), "prns_searched": 32, "bins": 29, "non_coh_ms": 20,
                             "fft_size": 36000, "acquired": found,
                             "all_scene_prns_found": sorted(s.prn for s in sats) == found}

    # ---- CPU baseline: the oracle's C restatement of tracking.m, closed loop, 1 core ------------
    if rank == 0 and world == 1 and not args.no_cpu:
        from types import SimpleNamespace

        from oracle import c_oracle as CO
        CO.build(force=True)
        cpu_epochs = min(args.cpu_epochs, n_epochs)
        n_cpu = int((cpu_epochs + 3) * 1e-3 * fs)
        iq = eng.read_if(0, min(n_cpu, n_samples))
        Sc = P.initSettings()
        Sc.msToProcess = cpu_epochs
        ch = [SimpleNamespace(PRN=i.prn, acquiredFreq=i.acquired_freq, codePhase=i.code_phase, status="T") for i in inits]
        t0 = time.perf_counter()
        ref, cdone, aborted = CO.track_l1ca(iq, ch, Sc)
        t_cpu = time.perf_counter() - t0
        cpu_samples = float(np.sum(np.ceil((Sc.codeLength - ref["remCodePhase"]) / (ref["codeFreq"] / fs))))
        cpu_msps = cpu_samples / t_cpu / 1e6
        # Parity at identical descriptors: the CPU loop's own recorded per-epoch state (tracking.m:212-216,249,277,314,332)
        # replayed through the GPU correlator, all 12 x cpu_epochs blocks, against the CPU loop's sums.  (The two CLOSED
        # loops cannot be compared sample for sample for long: float32 partial sums make their NCO states differ by ~1e-8
        # chip after a few thousand epochs, enough to put a sample that sits on a chip edge on the other side, and
        # eventually to tip ceil((L - rem)/step) at a knife edge - DESIGN.md 4.3b; the first such epoch is reported.)
        cb = eng.make_blocks(nch * cpu_epochs)
        cv = np.frombuffer(cb, dtype=dt)
        for k in range(nch):
            sl = slice(k, nch * cpu_epochs, nch)
            cv["channel"][sl] = k
            cv["blksize"][sl] = np.ceil((Sc.codeLength - ref["remCodePhase"][k]) / (ref["codeFreq"][k] / fs)).astype(np.int64)
            cv["first_sample"][sl] = ref["absoluteSample"][k].astype(np.int64)
            cv["rem_code_phase"][sl] = ref["remCodePhase"][k]
            cv["code_phase_step"][sl] = ref["codeFreq"][k] / fs
            cv["el_spacing"][sl] = Sc.dllCorrelatorSpacing
            cv["carr_freq"][sl] = ref["carrFreq"][k]
            cv["rem_carr_phase"][sl] = ref["remCarrPhase"][k]
        got = eng.correlate(cb)[:, 0, :]
        want = np.stack([ref[f].T.reshape(-1) for f in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")], axis=1)
        dev = float(np.max(np.abs(got - want))) / scale
        same = ref["absoluteSample"] == fields["absoluteSample"][:, :cpu_epochs]
        n_same = [int(np.argmin(r)) if not r.all() else cpu_epochs for r in same]
        result["cpu_baseline"] = {
            "value": round(cpu_msps, 2), "unit": "Msamples/s (channel-samples)", "cores": 1, "kind": "port",
            "sample": f"oracle/gnss_oracle.c (float64 restatement of tracking.m:133-368, gcc -O3), closed loop, "
                      f"{nch} channels x {cpu_epochs} epochs of the same record ({t_cpu:.1f} s of CPU)",
            "x_realtime": round(cpu_msps / nch / (fs / 1e6), 4),
            "gpu_replay_of_cpu_state_max_dev": dev,
            "closed_loops_cut_the_same_blocks_for_epochs": int(min(n_same)),
        }
    if rank == 0:
        print(json.dumps(result), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
