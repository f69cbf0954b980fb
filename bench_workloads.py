"""Synthetic workloads of BASELINE.json's configs for bench.py: per-package signal models, band records, closed loops and
batched replays through the C-ABI.  Bench plumbing only (not part of the product package); nothing here touches oracle/."""
from __future__ import annotations

import time
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np

BLOCK_DT = np.dtype([("channel", "<i4"), ("blksize", "<i4"), ("first_sample", "<i8"), ("rem_code_phase", "<f8"),
                     ("code_phase_step", "<f8"), ("el_spacing", "<f8"), ("carr_freq", "<f8"), ("rem_carr_phase", "<f8"),
                     ("table_offset", "<i4", (3,)), ("reserved", "<i4")])
assert BLOCK_DT.itemsize == 72


@dataclass
class Package:
    """How one reference package's signal is synthesised and tracked."""
    signal: str                    # cu_sdr_collection_amd.signals.SIGNALS key
    settings_fn: str               # cu_sdr_collection_amd.settings mirror of the package's initSettings.m
    overrides: dict                # settings changes (pilot tracking on, narrow correlator ...)
    code: object                   # P -> (prn -> chips)
    code_len: int
    rate_mult: float = 1.0         # chips of `code` per settings.codeFreqBasis chip (2: BOC(1,1) half-chip tables)
    bit_periods: int = 20
    pilot: object = None           # P -> (prn -> chips) of the pilot component
    pilot_phase: float = 0.0
    carrier_ratio: float = 1540.0
    prns: tuple = tuple(range(1, 33))
    glonass: bool = False          # satellites are frequency numbers K; IF offset K*freqSpacing; Q,I sample order
    l2c_phase: bool = False        # channels carry CLCodePhase


def _l2c_combined(P):
    def f(prn):
        cm, cl = P.codes.generateCMcode(prn).astype(np.int8), P.codes.generateCLcode(prn).astype(np.int8)
        return np.tile(cm, 75) + cl
    return f


PACKAGES = {
    "GPS_L1CA": Package("GPS_L1CA", "initSettings", {}, lambda P: P.codes.generateCAcode, 1023),
    "GAL_E1C": Package("GAL_E1C", "initSettings_GAL_E1C", dict(pilotTRKflag=1), lambda P: P.codes.generateE1Bcode, 8184, 2.0, 1,
                       lambda P: P.codes.generateE1Ccode, 0.0, prns=tuple(range(1, 37))),
    # BASELINE config 3 as worded: the E1-C pilot tracked with its CBOC(6,1,1/11) replica (an extension: the reference's package
    # stops at BOC(1,1)); narrow correlator so that spacing * 2 * 6 < 1 table entry
    "GAL_E1C_CBOC": Package("GAL_E1C_CBOC", "initSettings_GAL_E1C", dict(pilotTRKflag=1, dllCorrelatorSpacing=0.05), lambda P: P.codes.generateE1Bcode, 8184, 2.0, 1,
                            lambda P: P.codes.generateE1Ccode, 0.0, prns=tuple(range(1, 37))),
    "BDS_B1C_NB": Package("BDS_B1C_NB", "initSettings_BDS_B1C", dict(pilotTRKflag=1), lambda P: P.codes.generateDataBOC11, 20460, 2.0, 1,
                          lambda P: P.codes.generatePilotBOC11, np.pi / 2, prns=tuple(range(19, 47))),
    "BDS_B1I": Package("BDS_B1I", "initSettings_BDS_B1I", {}, lambda P: P.codes.generateCAcode53, 2046, carrier_ratio=1526.0, prns=tuple(range(6, 38))),
    "GPS_L5C": Package("GPS_L5C", "initSettings_GPS_L5C", dict(pilotTRKflag=1), lambda P: P.codes.generateL5Icode, 10230, 1.0, 10,
                       lambda P: P.codes.generateL5Qcode, np.pi / 2, 1150.0),
    "GAL_E5a": Package("GAL_E5a", "initSettings_GAL_E5a", dict(pilotTRKflag=1), lambda P: (lambda prn: P.codes.generateE5aIcode(prn, 1)), 10230, 1.0, 20,
                       lambda P: (lambda prn: P.codes.generateE5aQcode(prn, 1)), np.pi / 2, 1150.0, prns=tuple(range(1, 37))),
    "BDS_B2a": Package("BDS_B2a", "initSettings_BDS_B2a", dict(pilotTRKflag=1), lambda P: P.codes.generateB2aDataCode, 10230, 1.0, 5,
                       lambda P: P.codes.generateB2aPilotCode, np.pi / 2, 1150.0, prns=tuple(range(19, 47))),
    "GAL_E5b": Package("GAL_E5b", "initSettings_GAL_E5b", dict(pilotTRKflag=1), lambda P: (lambda prn: P.codes.generateE5bIcode(prn, 1)), 10230, 1.0, 4,
                       lambda P: (lambda prn: P.codes.generateE5bQcode(prn, 1)), np.pi / 2, 1180.0, prns=tuple(range(1, 37))),
    "BDS_B3I": Package("BDS_B3I", "initSettings_BDS_B3I", {}, lambda P: P.codes.generateB3Icode, 10230, 1.0, 20, carrier_ratio=1240.0, prns=tuple(range(6, 60))),
    "GPS_L2C": Package("GPS_L2C", "initSettings_GPS_L2C", dict(pilotTRKflag=1), _l2c_combined, 20460 * 75, 2.0, 1, carrier_ratio=1200.0, l2c_phase=True),
    "GLO_GL1": Package("GLO_GL1", "initSettings_GLO_GL1", {}, lambda P: (lambda k: P.codes.generateGLOcode()), 511, 1.0, 10, carrier_ratio=3135.0,
                       prns=tuple(range(-7, 7)), glonass=True),
    "GLO_GL2": Package("GLO_GL2", "initSettings_GLO_GL2", {}, lambda P: (lambda k: P.codes.generateGLOcode()), 511, 1.0, 10, carrier_ratio=2438.0,
                       prns=tuple(range(-7, 7)), glonass=True),
}

# BASELINE config 5: 64 channels of the twelve signals, grouped by the IF record (band) they are tracked from
MIX_BANDS = {
    "L1": [("GPS_L1CA", 8), ("GAL_E1C", 6), ("BDS_B1C_NB", 4)],
    "L5": [("GPS_L5C", 6), ("GAL_E5a", 5), ("BDS_B2a", 5)],
    "B1I": [("BDS_B1I", 5)], "E5b": [("GAL_E5b", 5)], "B3I": [("BDS_B3I", 5)], "L2": [("GPS_L2C", 5)],
    "GLO_L1": [("GLO_GL1", 5)], "GLO_L2": [("GLO_GL2", 5)],
}


def settings_for(P, pkg: Package, **extra):
    from cu_sdr_collection_amd import settings as SET
    S = getattr(SET, pkg.settings_fn)()
    for k, v in {**pkg.overrides, **extra}.items():
        setattr(S, k, v)
    return S


def make_sats(P, pkg: Package, S, n: int, seed: int, cn0: float = 46.0, doppler_max: float = 4e3):
    rng = np.random.default_rng(seed)
    ids = rng.choice(np.array(pkg.prns), size=n, replace=False)
    period = S.samplingFreq * S.intTime
    return [P.synth.SatSpec(prn=int(p), doppler=float(rng.uniform(-doppler_max, doppler_max)), code_phase_samples=float(rng.uniform(0, period)),
                            carrier_phase=float(rng.uniform(0, 2 * np.pi)), cn0_dbhz=cn0) for p in ids]


def signal_group(P, pkg: Package, S, sats):
    from cu_sdr_collection_amd.synth import SignalGroup
    groups = []
    if pkg.glonass:      # one group per frequency number: the FDMA offset is the group's intermediate frequency
        for s in sats:
            groups.append(SignalGroup([s], pkg.code(P), pkg.rate_mult * S.codeFreqBasis, pkg.code_len, pkg.bit_periods, None, 0.0, pkg.carrier_ratio,
                                      intermediate_freq=S.IF + s.prn * S.freqSpacing))
        return groups
    return [SignalGroup(list(sats), pkg.code(P), pkg.rate_mult * S.codeFreqBasis, pkg.code_len, pkg.bit_periods,
                        pkg.pilot(P) if pkg.pilot else None, pkg.pilot_phase, pkg.carrier_ratio)]


@dataclass
class Job:
    """One tracking() call: a package's channels on one engine (context)."""
    name: str
    pkg: Package
    S: object
    sats: list
    engine: object
    params: object = None
    inits: list = None
    fields: dict = None            # raw gc_track records [nch, n_epochs]
    done: object = None
    blks: object = None            # block sizes [nch, n_epochs]
    phase0: list = None


def prepare_job(P, job: Job, n_epochs: int):
    from cu_sdr_collection_amd import _lib as L
    from cu_sdr_collection_amd import signals
    from cu_sdr_collection_amd.receiver import track_params
    S, pkg = job.S, job.pkg
    spec = signals.SIGNALS[pkg.signal]
    S.msToProcess = int(round(n_epochs * S.intTime * 1000))
    S.numberOfChannels = len(job.sats)
    p = track_params(S, pkg.signal)
    assert p.n_epochs == n_epochs, (p.n_epochs, n_epochs)
    inits, phases = [], []
    for i, s in enumerate(job.sats):
        job.engine.set_channel(i, spec.tables(s.prn, S), index_scale=spec.index_scale, arm_mult=spec.arm_mult, windows=spec.windows)
        f = S.IF + s.doppler + 2.0 + (s.prn * S.freqSpacing if pkg.glonass else 0.0)
        cf = S.codeFreqBasis + (f - S.IF) / S.carrFreqBasis * S.codeFreqBasis if spec.code_freq_from_channel else S.codeFreqBasis
        if spec.doubled_code:
            cf = 2 * S.codeFreqBasis
        cp = int(np.ceil(s.code_phase_samples)) + 1
        ph = 0
        if pkg.l2c_phase:
            cp -= 1                      # GPS_L2C tracking.m:153 seeks to skipNumberOfBytes + codePhase
            ph = 1                       # the combined CM/CL code of the generator starts in CL segment 1 at code_phase_samples
        phases.append(ph)
        inits.append(L.gc_channel_init(channel=i, prn=int(s.prn), acquired_freq=f, code_freq=cf, code_phase=cp, table_phase=ph))
    job.params, job.inits, job.phase0 = p, inits, phases
    return job


def run_closed_loops(P, jobs, device_loop: bool):
    """All jobs' tracking loops at once (gc_track_multi); returns the wall time.  Raw records go into job.fields."""
    t0 = time.perf_counter()
    res = P.Engine.track_multi([(j.engine, j.params, j.inits) for j in jobs], device_loop=device_loop)
    dt = time.perf_counter() - t0
    out = []
    for j, (fields, done, st) in zip(jobs, res):
        if st != 0 or int(done.min()) != j.params.n_epochs:
            raise RuntimeError(f"{j.name}: closed loop stopped early (status {st}, epochs {done})")
        out.append(fields)
    return dt, out


def keep_records(job: Job, fields):
    p = job.params
    job.fields = fields
    fs = p.sampling_freq
    job.blks = np.ceil((p.code_length - fields["remCodePhase"]) / (fields["codeFreq"] / fs)).astype(np.int64)


def locked(job: Job, skip_frac: float = 0.25):
    n0 = int(job.params.n_epochs * skip_frac)
    f = job.fields
    e = np.hypot(f["I_E"][:, n0:], f["Q_E"][:, n0:]).mean(axis=1)
    pr = np.hypot(f["I_P"][:, n0:], f["Q_P"][:, n0:]).mean(axis=1)
    return int(np.sum(pr > 1.15 * e))


def replay_blocks(job: Job):
    """Epoch-major descriptor list (the channels of an epoch next to each other) from the recorded per-epoch state
    (tracking.m:212-216,249,277,314,332)."""
    p, f = job.params, job.fields
    nch, n_ep = job.blks.shape
    nb = nch * n_ep
    blocks = job.engine.make_blocks(nb)
    v = np.frombuffer(blocks, dtype=BLOCK_DT)
    for k in range(nch):
        sl = slice(k, nb, nch)
        v["channel"][sl] = k
        v["blksize"][sl] = job.blks[k]
        v["first_sample"][sl] = f["absoluteSample"][k].astype(np.int64)
        v["rem_code_phase"][sl] = f["remCodePhase"][k]
        v["code_phase_step"][sl] = f["codeFreq"][k] / p.sampling_freq
        v["el_spacing"][sl] = p.el_spacing
        v["carr_freq"][sl] = f["carrFreq"][k]
        v["rem_carr_phase"][sl] = f["remCarrPhase"][k]
        if p.table_phase_count > 0 and job.phase0[k] > 0:
            ph = (job.phase0[k] - 1 + np.arange(n_ep)) % p.table_phase_count          # GPS_L2C tracking.m:261,357-360
            v["table_offset"][sl, 1] = (int(p.code_length) * ph).astype(np.int32)
    return blocks, v


def warm_engine(P, job: Job):
    """A second context on the job's record with the first quarter of its replay list prepared: what clocks_up launches.  Its
    grid differs from the measured replay's, so a profiler's per-kernel statistics of the measured launches stay undiluted."""
    import copy
    eng = P.Engine(job.engine.device_id)
    eng.share_if(job.engine)
    eng.set_sampling_freq(job.params.sampling_freq)
    prepare_job(P, Job(job.name + ":warm", job.pkg, copy.copy(job.S), job.sats, eng), job.params.n_epochs)
    _, v = replay_blocks(job)
    nch = job.blks.shape[0]
    n = max(nch, (job.blks.shape[1] // 4) * nch)
    sub = eng.make_blocks(n)
    np.frombuffer(sub, dtype=BLOCK_DT)[:] = v[:n]
    eng.replay_prepare(sub)
    return eng


def clocks_up(warm, ms: float):
    """Untimed launches on a warm_engine until `ms` milliseconds have passed: after a mostly idle phase (synthesis, the closed
    loops' persistent kernels) the device needs ~35 ms of work to reach its clocks (4.4 ms per config-2 pass falling to 3.6)."""
    if warm is None or ms <= 0:
        return
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        warm.replay_launch()
        warm.synchronize()


def time_replay(job: Job, steps: int, warmup: int, warm=None, prewarm_ms: float = 40.0):
    """K launches of the batched replay of one job; returns (ms per launch from hipEvents on the launch stream, max deviation of
    the replayed sums from the closed loop's own records, in units of full scale)."""
    eng = job.engine
    blocks, _ = replay_blocks(job)
    eng.replay_prepare(blocks)
    clocks_up(warm, prewarm_ms)
    for _ in range(warmup):
        eng.replay_launch()
    eng.synchronize()
    eng.timer_start()
    for _ in range(steps):
        eng.replay_launch()
    ms = eng.timer_stop() / steps
    out = eng.replay_fetch()
    f = job.fields
    arms = 2 if job.params.pilot_combine in (1, 2, 3) else 1     # modes 4 / 5 record the FOLDED pilot, not arm 1 as correlated
    names = ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")
    rec = np.stack([f[n].T.reshape(-1) for n in names], axis=1)
    comp = 2.0
    scale = comp * float(job.blks.mean()) * 28.0
    dev = float(np.max(np.abs(out[:, 0, :] - rec)) / scale)
    if arms == 2:
        recp = np.stack([f["Pilot_" + n].T.reshape(-1) for n in names], axis=1)
        dev = max(dev, float(np.max(np.abs(out[:, 1, :] - recp)) / scale))
    return ms, dev, eng.last_kernel()


def time_replays_together(jobs, steps: int, warmup: int):
    """The replays of several jobs of ONE record (one engine = one context and stream each), every pass launching all of them
    before the next pass: milliseconds per pass of the whole band, host clock around `steps` passes between two synchronisations
    (events of one stream do not bracket the others).  Each job's list was prepared by time_replay before."""
    import time
    for _ in range(max(1, warmup)):
        for j in jobs:
            j.engine.replay_launch()
    for j in jobs:
        j.engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for j in jobs:
            j.engine.replay_launch()
    for j in jobs:
        j.engine.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


KERNEL_NAMES = {0: "corr_epl_lane_kernel", 1: "corr_epl_fast_kernel (one-wave workgroups)", 2: "corr_epl_fast_kernel (four waves, int8-pair tables)",
                3: "corr_epl_fast_kernel (four waves, float tables)", 4: "corr_epl_multi_kernel (up to 2 / 4 transitions per 16-sample chunk)",
                5: "corr_epl_cboc_kernel (BOC(1,1) arms by transitions, BOC(6,1) arm by a per-sample sign)", -1: "corr_epl_mixed_kernel"}


def band_scene(P, parts, fs: float, seed: int, cn0: float = 46.0):
    """The satellites of one band record, the same on every rank: parts = [(package name, number of channels)] ->
    ([(Package, settings, sats)] in the same order, the synthesiser's signal groups)."""
    groups, out = [], []
    for k, (name, n) in enumerate(parts):
        pkg = PACKAGES[name]
        S = settings_for(P, pkg)
        S.samplingFreq = fs
        sats = make_sats(P, pkg, S, n, seed + 17 * k, cn0=cn0)
        groups += signal_group(P, pkg, S, sats)
        out.append((pkg, S, sats))
    return out, groups


def band_is_qi(parts) -> bool:
    return any(PACKAGES[name].glonass for name, _ in parts)


def make_band(P, engine, parts, seconds: float, fs: float, intermediate_freq: float, seed: int, dtype=np.int8, cn0: float = 46.0,
              attached: bool = False):
    """Synthesises one band record in the engine's HBM (attached=True: into the caller-owned buffer the engine already reads);
    returns band_scene's [(Package, settings, sats)]."""
    out, groups = band_scene(P, parts, fs, seed, cn0)
    P.synth.generate_if_mix_gpu(engine, groups, int(round(seconds * fs)), fs, intermediate_freq, seed, dtype=dtype,
                                qi_order=band_is_qi(parts), attached=attached)
    engine.set_sampling_freq(fs)
    return out
