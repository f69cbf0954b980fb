#!/usr/bin/env python3
"""Timing of concurrent multi-package tracking (gc_track_multi) against the same calls run one after the other, on a
GPU-synthesised L1-band record with GPS L1 C/A + Galileo E1 B/C + BDS B1C channels.  GC_PERSIST_COOP=0/1 selects plain /
cooperative launches of the persistent kernels (gc_internal.h)."""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cu_sdr_collection_amd as P  # noqa: E402
from cu_sdr_collection_amd.settings import initSettings_BDS_B1C, initSettings_GAL_E1C  # noqa: E402
from cu_sdr_collection_amd.synth import SatSpec, SignalGroup, generate_if_mix_gpu  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    fs = 18e6
    rng = np.random.default_rng(1)

    def sats(prns, period, cn0):
        return [SatSpec(prn=p, doppler=float(rng.uniform(-3e3, 3e3)), code_phase_samples=float(rng.uniform(0, period)),
                        carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=cn0) for p in prns]

    l1, e1, b1c = sats((3, 17, 28), 18000, 46.0), sats((4, 19, 31), 72000, 47.0), sats((8, 41), 180000, 47.0)
    groups = [SignalGroup(l1, P.codes.generateCAcode, 1.023e6, 1023),
              SignalGroup(e1, P.codes.generateE1Bcode, 2.046e6, 8184, bit_periods=1, pilot_fn=P.codes.generateE1Ccode),
              SignalGroup(b1c, P.codes.generateDataBOC11, 2.046e6, 20460, bit_periods=1, pilot_fn=P.codes.generatePilotBOC11, pilot_phase=np.pi / 2)]
    ms = int(seconds * 1000) - 30
    S1 = P.initSettings()
    S2, S3 = initSettings_GAL_E1C(), initSettings_BDS_B1C()
    for S, n in ((S1, 3), (S2, 3), (S3, 2)):
        S.msToProcess, S.numberOfChannels = ms, n

    def chans(S, sv, code_freq=False):
        out = []
        for s in sv:
            f = S.IF + s.doppler + 2.0
            c = SimpleNamespace(PRN=s.prn, acquiredFreq=f, status="T", codePhase=int(np.ceil(s.code_phase_samples)) + 1)
            if code_freq:
                c.codeFreq = S.codeFreqBasis + (f - S.IF) / S.carrFreqBasis * S.codeFreqBasis
            out.append(c)
        return out

    e1_, e2_, e3_ = P.Engine(0), P.Engine(0), P.Engine(0)
    generate_if_mix_gpu(e1_, groups, int(seconds * fs), fs, 20e3, seed=7)
    e1_.set_sampling_freq(fs)
    e2_.share_if(e1_)
    e3_.share_if(e1_)
    calls = [(e1_, chans(S1, l1), S1, "GPS_L1CA"), (e2_, chans(S2, e1), S2, "GAL_E1C"), (e3_, chans(S3, b1c, True), S3, "BDS_B1C_NB")]
    for dev in (False, True):
        P.receiver.tracking_multi(calls, device_loop=dev)   # warm-up
        t0 = time.perf_counter()
        res = P.receiver.tracking_multi(calls, device_loop=dev)
        t_multi = time.perf_counter() - t0
        t_seq = []
        for c in calls:
            t0 = time.perf_counter()
            try:
                P.tracking(*c, device_loop=dev)
            except P.GnssCorrError:
                P.tracking(*c)
            t_seq.append(time.perf_counter() - t0)
        locked = [[bool(np.mean(np.abs(t.I_P[20:])) > 2 * np.mean(np.abs(t.Q_P[20:]))) and t.status == "T" for t in tr] for tr, _ in res]
        print(json.dumps({"device_loop": dev, "seconds": seconds, "coop_env": os.environ.get("GC_PERSIST_COOP"), "t_multi_s": round(t_multi, 4),
                          "t_sequential_s": [round(t, 4) for t in t_seq], "sum_sequential_s": round(sum(t_seq), 4),
                          "x_realtime_multi": round((ms / 1000) / t_multi, 1), "locked": locked}), flush=True)


if __name__ == "__main__":
    main()
