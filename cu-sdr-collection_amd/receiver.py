"""Host-side mirror of the reference's operator interface for the hot path (GPS L1 C/A package):

    acqResults            = acquisition(longSignal, settings)       GPS/GPS_L1CA/include/acquisition.m:1
    channel               = preRun(acqResults, settings)            GPS/GPS_L1CA/include/preRun.m:1
    [trackResults, chan]  = tracking(fid, channel, settings)        GPS/GPS_L1CA/include/tracking.m:1

Same names, argument meaning, struct fields and error behaviour; the heavy lifting is done by
libgnsscorr.so on the GPU (no CPU path exists here).  `fid` is an Engine whose IF buffer plays
the role of the open file; `longSignal` is identified by its position in that buffer.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

from . import _lib as L
from .settings import skip_samples
from . import codes
from .engine import Engine


def _round(x: float) -> int:
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


# ---------------------------------------------------------------------------------------------
# C/N0 estimator (host side, Common/CNoVSM.m:38-47)
# ---------------------------------------------------------------------------------------------
def CNoVSM(I, Q, T):
    Z = np.asarray(I) ** 2 + np.asarray(Q) ** 2
    Zm = np.mean(Z)
    Zv = np.var(Z, ddof=1)
    Pav = np.sqrt(complex(Zm ** 2 - Zv))
    Nv = 0.5 * (Zm - Pav)
    return float(10 * np.log10(abs((1 / T) * Pav / (2 * Nv))))


def Calc_CNo_PLD(trackResults, settings, loopCnt, straight_pilot: bool = False):
    """[CNo, PllDetector] = Calc_CNo_PLD(trackResults, settings, loopCnt) of BDS/B2a and BDS/B1C (host side, like CNoVSM):
    variance-summing C/N0 of the data arm, of the pilot arm and of their sum over the last settings.CNoInterval epochs, and
    the narrow-band PLL lock detector NBD/NBP of each arm with the data bits wiped by sign (B2a Calc_CNo_PLD.m:35-97).  The
    pilot prompt pair is read swapped (the pilot is tracked in quadrature) except for the B1C wide-band loop
    (`straight_pilot`; pilotTRKflag == 2 in BDS/B1C/include/Calc_CNo_PLD.m).  CNo[1], PllDetector[1] stay 0 without a pilot."""
    n = int(settings.CNoInterval)
    T = settings.intTime
    sl = slice(loopCnt - n, loopCnt)

    def arm(I, Q):
        Z = I ** 2 + Q ** 2
        Zm = np.mean(Z)
        Zv = np.var(Z, ddof=1)
        Pav = np.sqrt(complex(Zm ** 2 - Zv))          # MATLAB's sqrt of a negative number is complex; abs() below
        Nv = 0.5 * (Zm - Pav)
        lin = abs((1 / T) * Pav / (2 * Nv))
        wiped = np.sum(I[I > 0]) - np.sum(I[I < 0])
        nbp = wiped ** 2 + np.sum(Q) ** 2
        nbd = wiped ** 2 - np.sum(Q) ** 2
        return lin, nbd / nbp

    CNo = np.zeros(3)
    PllDetector = np.zeros(2)
    data, PllDetector[0] = arm(np.asarray(trackResults.I_P[sl], dtype=np.float64), np.asarray(trackResults.Q_P[sl], dtype=np.float64))
    CNo[0] = 10 * np.log10(data)
    pilot = 0.0
    if getattr(settings, "pilotTRKflag", 0) in (1, 2) and hasattr(trackResults, "Pilot_I_P"):
        pi = np.asarray(trackResults.Pilot_I_P[sl], dtype=np.float64)
        pq = np.asarray(trackResults.Pilot_Q_P[sl], dtype=np.float64)
        pilot, PllDetector[1] = arm(pi, pq) if straight_pilot else arm(pq, pi)
        CNo[1] = 10 * np.log10(pilot)
    with np.errstate(divide="ignore"):
        CNo[2] = 10 * np.log10(data + pilot)
    return CNo, PllDetector


# ---------------------------------------------------------------------------------------------
# acquisition
# ---------------------------------------------------------------------------------------------
def _acq_params(settings, first_sample: int) -> L.gc_acq_params:
    p = L.gc_acq_params()
    p.sampling_freq = settings.samplingFreq
    p.code_freq_basis = settings.codeFreqBasis
    p.code_length = settings.codeLength
    p.intermediate_freq = settings.IF
    p.search_band = settings.acqSearchBand
    p.search_step = settings.acqSearchStep
    p.non_coh_time = int(settings.acqNonCohTime)
    p.first_sample = int(first_sample)
    return p


def acquisition(engine: Engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """acqResults = acquisition(longSignal, settings) with longSignal = the IF buffer from
    `first_sample` (default settings.skipNumberOfBytes, as postProcessing.m:74-96 reads it).

    With settings.resamplingflag == 1 and samplingFreq above settings.resamplingThreshold the search runs on the conditioned
    signal of acquisition.m:46-111 (zero-phase FIR(700) band-pass + band-pass-sampling decimation, Engine.acq_condition) and the
    results are mapped back to the record's sampling rate and IF (:264-276).  `n_long` = length(longSignal) for that case
    (default: max(42, acqNonCohTime + 2) code periods, postProcessing.m:82-83, or what the buffer holds).
    """
    import copy
    import math
    if first_sample is None:
        first_sample = skip_samples(settings)
    flag = getattr(settings, "resamplingflag", getattr(settings, "resamplingFlag", 0))
    resampled = settings.samplingFreq > settings.resamplingThreshold and flag == 1
    S = settings
    if resampled:
        if n_long is None:
            n_long = min(int(engine.if_buffer()[1]) - int(first_sample), max(42, int(settings.acqNonCohTime) + 2) * codes.samplesPerCode(settings))
        bw = settings.codeFreqBasis * 2 + 0.5e6                      # acquisition.m:58
        new_fs, new_if, _ = engine.acq_condition(settings.samplingFreq, settings.IF, bw, first_sample, n_long)
        S = copy.copy(settings)
        S.samplingFreq, S.IF = new_fs, new_if                        # :81,95
    prns = list(S.acqSatelliteList)
    acq = SimpleNamespace(carrFreq=np.zeros(32), codePhase=np.zeros(32), peakMetric=np.zeros(32))
    src, first = (1, 0) if resampled else engine.acq_input(first_sample, settings.samplingFreq, n_long)[:2]   # int16 / Q-I / real records: float copy
    p = _acq_params(S, first)
    p.source = src
    tables = np.stack([codes.makeCaTable(prn, S) for prn in prns])
    res = engine.acquire_coarse(p, tables)
    found = []
    for prn, r in zip(prns, res):
        acq.peakMetric[prn - 1] = r.peak_metric                      # acquisition.m:200
        if r.peak_metric > S.acqThreshold:                           # :206
            found.append((prn, r))
    if found:                                                        # the fine stage of every detection in one launch
        f = engine.acquire_fine_l1ca_batch(p, np.stack([codes.generateCAcode(prn) for prn, _ in found]),
                                           [r.code_phase for _, r in found], [r.coarse_freq for _, r in found])
        for (prn, r), fk in zip(found, f):
            acq.carrFreq[prn - 1] = fk                               # :254-260
            acq.codePhase[prn - 1] = r.code_phase                    # :256
            if resampled:                                            # :264-276: back to the record's rate and IF
                acq.codePhase[prn - 1] = math.floor((r.code_phase - 1) / S.samplingFreq * settings.samplingFreq) + 1
                if S.IF >= S.samplingFreq / 2:
                    doppler = (S.samplingFreq - S.IF) - fk
                else:
                    doppler = fk - S.IF
                acq.carrFreq[prn - 1] = doppler + settings.IF
    return acq


# ---------------------------------------------------------------------------------------------
# preRun
# ---------------------------------------------------------------------------------------------
def preRun(acqResults, settings, signal: str = "GPS_L1CA"):
    """channel = preRun(acqResults, settings) of package `signal` (include/preRun.m of each package): the strongest
    detections first, one struct per channel.  Per-package fields: `codeFreq` for the packages whose tracking.m starts the code
    NCO from it (GPS_L5C preRun.m:69-71: codeFreqBasis + (acquiredFreq - IF)/carrFreqBasis*codeFreqBasis; also GAL_E5a / E5b,
    BDS B1C / B2a / B3I), `CLCodePhase` for GPS L2C with the pilot on (GPS_L2C preRun.m:70-72), `K = index - 8` instead of
    `PRN` for GLONASS (GLO_GL1 preRun.m:66).  Indices run over the whole acqResults arrays, whatever their length (32, 50, 63
    PRNs; 14 frequency numbers at K + 8)."""
    from . import signals
    spec = signals.SIGNALS[signal]
    n_ch = int(settings.numberOfChannels)
    glo = spec.id_field == "K"
    channel = []
    for _ in range(n_ch):
        ch = SimpleNamespace(acquiredFreq=0.0, codePhase=0, status="-")
        setattr(ch, spec.id_field, 0)
        if spec.code_freq_from_channel:
            ch.codeFreq = 0.0
        channel.append(ch)
    order = np.argsort(-np.asarray(acqResults.peakMetric, dtype=np.float64), kind="stable")   # preRun.m:60 (sort ... 'descend' is stable)
    n_found = int(np.sum(np.asarray(acqResults.carrFreq) != 0))
    for ii in range(min(n_ch, n_found)):                                      # :65
        p = int(order[ii])
        ch = channel[ii]
        setattr(ch, spec.id_field, p + 1 - 8 if glo else p + 1)
        ch.acquiredFreq = float(acqResults.carrFreq[p])
        ch.codePhase = int(acqResults.codePhase[p])
        if spec.code_freq_from_channel:
            ch.codeFreq = settings.codeFreqBasis + (ch.acquiredFreq - settings.IF) / settings.carrFreqBasis * settings.codeFreqBasis
        if spec.doubled_code and getattr(settings, "pilotTRKflag", 0):
            ch.CLCodePhase = int(acqResults.CLCodePhase[p])
        ch.status = "T"
    return channel


# ---------------------------------------------------------------------------------------------
# tracking
# ---------------------------------------------------------------------------------------------
_REC_FIELDS = ("absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L",
               "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt", "remCodePhase", "remCarrPhase")
_PILOT_FIELDS = ("Pilot_I_E", "Pilot_Q_E", "Pilot_I_P", "Pilot_Q_P", "Pilot_I_L", "Pilot_Q_L")
# tracking.m:47-86: these eight are created with inf(1, n), the others with zeros(1, n); epochs a channel never reaches keep them
_INF_FIELDS = ("codeFreq", "carrFreq", "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt", "remCodePhase", "remCarrPhase")
# "reference": trackResults carries exactly the Pilot_* fields the package's tracking.m records (signals.SignalSpec.recorded_pilot);
# "all": all six pilot sums whenever a pilot arm is correlated (the reference computes the early / late ones but drops them)
DEFAULT_PILOT_FIELDS = "reference"


def _recorded_pilot_fields(spec, pilot: bool, mode: str | None):
    if not pilot:
        return ()
    mode = mode or DEFAULT_PILOT_FIELDS
    if mode == "all" or spec.recorded_pilot == "all":
        return _PILOT_FIELDS
    return ("Pilot_I_P", "Pilot_Q_P") if spec.recorded_pilot == "prompt" else ()


def track_params(settings, signal: str = "GPS_L1CA") -> L.gc_track_params:
    from . import signals
    spec = signals.SIGNALS[signal]
    p = L.gc_track_params()
    p.sampling_freq = settings.samplingFreq
    p.code_freq_basis = settings.codeFreqBasis
    p.code_length = settings.codeLength
    p.el_spacing = settings.dllCorrelatorSpacing
    p.int_time = settings.intTime
    p.dll_noise_bw = settings.dllNoiseBandwidth
    p.dll_damping = settings.dllDampingRatio
    p.pll_noise_bw = settings.pllNoiseBandwidth
    p.pll_damping = settings.pllDampingRatio
    cno = getattr(settings, "CNo", None)
    if cno is not None and not int(getattr(settings, "CNoInterval", 0)) and int(getattr(cno, "VSMinterval", 0)) > 1:
        p.cno_interval = int(cno.VSMinterval)          # tracking.m:351-358: CNoVSM every VSMinterval epochs, inside the loop
        p.cno_acc_time = float(cno.accTime)
    p.pll_kind = spec.pll_kind
    flag = getattr(settings, "pilotTRKflag", 0)
    # BDS/B1C/include/postProcessing.m:69-74: pilotTRKflag 1 runs NB_tracking, 2 runs WB_tracking (both track the pilot)
    pilot = spec.pilot_combine if (flag == 1 or (flag == 2 and signal == "BDS_B1C_WB")) else 0
    p.pilot_combine = pilot
    if int(getattr(settings, "CNoInterval", 0)) > 1:
        # BDS/B2a tracking.m:409-432, B1C NB_tracking.m:397-418, WB_tracking.m:441-461: Calc_CNo_PLD every CNoInterval epochs; the
        # pilot prompt pair is read swapped (pilotTRKflag == 1, Calc_CNo_PLD.m:72-75) or straight (== 2, the B1C wide-band loop)
        p.cno_interval = int(settings.CNoInterval)
        p.cno_acc_time = float(settings.intTime)
        p.cno_mode = L.GC_CNO_PLD if not pilot else (L.GC_CNO_PLD_PILOT if signal.endswith("_WB") else L.GC_CNO_PLD_PILOT_SWAPPED)
    if spec.pll_kind == L.GC_PLL_3_STATE:
        p.pf3, p.pf2, p.pf1 = signals.calcLoopCoefCarr(settings, spec.coef_variant)
    if pilot:
        for name in ("pll_weight", "dll_weight"):
            w = getattr(spec, name)
            if callable(w):
                w = w(settings)
            if w is not None:
                getattr(p, name)[0], getattr(p, name)[1] = float(w[0]), float(w[1])
        if spec.dll_scale_spacing:
            p.dll_scale = 1.0 - settings.dllCorrelatorSpacing
    # tracking.m:145-153: dataAdaptCoeff*(skipNumberOfBytes + codePhase-1) bytes of schar components, or
    # dataAdaptCoeff*(skipNumberOfBytes + (codePhase-1)*2) bytes of int16 components = skipNumberOfBytes/2 + codePhase-1 samples
    if str(getattr(settings, "dataType", "schar")) == "int16" and not spec.int16_branch:
        raise NotImplementedError(f"{signal}: the reference's tracking.m has no int16 branch (its fseek assumes one byte per "
                                  "component and would start at half the code phase); convert the record to schar")
    p.skip_samples = skip_samples(settings)
    p.n_epochs = signals.epochs_to_process(settings)
    if spec.doubled_code:
        # GPS_L2C/include/tracking.m:107-109: spacing and code length in units of the RZ-doubled code; :153 seeks to
        # skipNumberOfBytes + codePhase WITHOUT the usual -1; :261,357-360 CL window bookkeeping
        p.el_spacing = settings.dllCorrelatorSpacing * 2
        p.code_length = settings.codeLength * 2
        p.code_freq_basis = settings.codeFreqBasis * 2
        p.skip_samples = skip_samples(settings) + 1
        p.table_phase_count = 75 if pilot else 0
    return p


def _tracking_prepare(fid: Engine, channel, settings, signal: str, pilot_fields: str | None = None):
    """The part of tracking() in front of the loops: result structs (tracking.m:47-86), code tables (:156-158), per-channel
    start state (:145-170).  Returns a job record for _tracking_finish."""
    from . import signals
    spec = signals.SIGNALS[signal]
    if settings.fileType not in (1, 2) or settings.dataType not in ("schar", "int8", "int16"):
        raise ValueError("tracking(): settings.fileType must be 1 (real) or 2 (I/Q), settings.dataType 'schar' or 'int16'")
    # fileType 1 (real samples, tracking.m:126-130,232-236): the record must have been loaded with layout GC_REAL
    n_ep = signals.epochs_to_process(settings)
    p = track_params(settings, signal)
    pilot = p.pilot_combine != 0
    rec_pilot = _recorded_pilot_fields(spec, pilot, pilot_fields)
    results = []
    active = []
    for i, ch in enumerate(channel):
        tr = SimpleNamespace(status="-", PRN=0)
        for f in _REC_FIELDS + rec_pilot:
            setattr(tr, f, np.full(n_ep, np.inf) if f in _INF_FIELDS else np.zeros(n_ep))
        tr.CNo = SimpleNamespace(VSMValue=[], VSMIndex=[])
        pld_n = int(getattr(settings, "CNoInterval", 0))
        if pld_n:   # BDS/B2a tracking.m:85-92, B1C NB_tracking.m:92-98: created for every channel, zeros until a record is due
            combined = "B2a_CNo" if signal.startswith("BDS_B2a") else "B1C_CNo"
            for f in ("DataCNo", "DataPLD") + (("PilotCNo", "PilotPLD", combined) if pilot else ()):
                setattr(tr, f, np.zeros(n_ep // pld_n))
        results.append(tr)
        sat = getattr(ch, spec.id_field, getattr(ch, "PRN", 0))
        if (ch.status != "-") if spec.id_field == "K" else (sat != 0):                # tracking.m:136 / GLO_GL1 tracking.m:138
            tr.PRN = sat                                                               # :138 / GLO :141
            fid.set_channel(i, spec.tables(sat, settings), index_scale=spec.index_scale, arm_mult=spec.arm_mult,
                            windows=spec.windows)
            active.append(i)
    inits = []
    for i in active:
        ch = channel[i]
        cf = ch.codeFreq if spec.code_freq_from_channel else settings.codeFreqBasis
        if spec.doubled_code:
            cf = settings.codeFreqBasis * 2                                           # GPS_L2C tracking.m:171
        inits.append(L.gc_channel_init(channel=i, prn=int(results[i].PRN), acquired_freq=ch.acquiredFreq,
                                       code_freq=cf, code_phase=int(ch.codePhase),
                                       table_phase=int(getattr(ch, "CLCodePhase", 0)) if (spec.doubled_code and pilot) else 0))
    return SimpleNamespace(fid=fid, channel=channel, settings=settings, signal=signal, spec=spec, n_ep=n_ep, p=p, pilot=pilot,
                           rec_pilot=rec_pilot, results=results, active=active, inits=inits)


def _tracking_finish(job, fields, done, status):
    """The part of tracking() behind the loops: records into the trackResults structs, C/N0 (tracking.m:351-358), status."""
    settings, signal, spec, pilot, n_ep, results, channel = job.settings, job.signal, job.spec, job.pilot, job.n_ep, job.results, job.channel
    # B2a / B1C estimate C/N0 with Calc_CNo_PLD every settings.CNoInterval epochs (below); the other packages with CNoVSM
    pld = int(getattr(settings, "CNoInterval", 0))
    cno = None if pld else getattr(settings, "CNo", None)
    vsm = int(cno.VSMinterval) if cno is not None else 0
    for k, i in enumerate(job.active):
        tr = results[i]
        n_done = int(done[k])
        for f in _REC_FIELDS + job.rec_pilot:
            getattr(tr, f)[:n_done] = fields[f][k][:n_done]          # epochs never reached keep their inf / 0 (tracking.m:47-86)
        if spec.doubled_code:
            # GPS_L2C tracking.m:226,250,376,382-383: what the reference RECORDS is in single-code units, and
            # absoluteSample is pushed back by the code-phase remainder expressed in samples
            step = tr.codeFreq[:n_done] / settings.samplingFreq
            tr.absoluteSample[:n_done] = tr.absoluteSample[:n_done] + 1 - tr.remCodePhase[:n_done] / step
            for f in ("remCodePhase", "codeFreq", "dllDiscr", "dllDiscrFilt"):
                getattr(tr, f)[:n_done] /= 2
        lib_cno = fields.get("CNoVSM")                                                # computed inside the loop (gc_track_params.cno_interval)
        for loop in (range(vsm, n_done + 1, vsm) if vsm else ()):                     # tracking.m:351-358
            if lib_cno is not None:
                tr.CNo.VSMValue.append(float(lib_cno[k][loop // vsm - 1]))
            else:
                tr.CNo.VSMValue.append(CNoVSM(tr.I_P[loop - vsm:loop], tr.Q_P[loop - vsm:loop], settings.CNo.accTime))
            tr.CNo.VSMIndex.append(loop)
        if pld:
            # BDS/B2a/include/tracking.m:85-92,191-192,409-432 (B1C NB_tracking.m:92-98,397-418, WB_tracking.m:443-461):
            # every CNoInterval epochs, the estimate averaged 0.5/0.5 with the previous one (zeros before the first)
            combined = "B2a_CNo" if signal.startswith("BDS_B2a") else "B1C_CNo"
            prev = np.zeros(3)
            lib_pld = fields.get("CNoPLD")                                            # [nch, nk, 5], evaluated by the library (gc_cno_mode)
            for loop in range(pld, n_done + 1, pld):
                kk = loop // pld - 1
                if lib_pld is not None:
                    v = lib_pld[k][kk]
                    tr.DataCNo[kk], tr.DataPLD[kk] = v[0], v[3]
                    if pilot:
                        tr.PilotCNo[kk], tr.PilotPLD[kk] = v[1], v[4]
                        getattr(tr, combined)[kk] = v[2]
                    continue
                c, d = Calc_CNo_PLD(tr, settings, loop, straight_pilot=signal.endswith("_WB"))
                tr.DataCNo[kk] = c[0] * 0.5 + prev[0] * 0.5
                tr.DataPLD[kk] = d[0]
                if pilot:
                    tr.PilotCNo[kk] = c[1] * 0.5 + prev[1] * 0.5
                    getattr(tr, combined)[kk] = c[2] * 0.5 + prev[2] * 0.5
                    tr.PilotPLD[kk] = d[1]
                prev = c
        if n_done == n_ep:
            tr.status = channel[i].status                                             # tracking.m:365
    if status == L.GC_E_RANGE:
        print("Not able to read the specified number of samples  for tracking, exiting!")
    return results, channel


def tracking(fid: Engine, channel, settings, signal: str = "GPS_L1CA", device_loop: bool = False, pilot_fields: str | None = None):
    """[trackResults, channel] = tracking(fid, channel, settings) — `signal` selects the reference
    package whose tracking.m is mirrored ("GPS_L1CA": GPS/GPS_L1CA/include/tracking.m;
    "GAL_E1C": GAL/GAL_E1C/include/tracking.m, data + pilot arms, BOC(1,1) half-chip tables; ... signals.SIGNALS).

    Returns (trackResults, channel).  On a short read the reference prints a message and
    returns what it has (tracking.m:241-245); here the partially filled results are returned
    the same way and `trackResults[i].status` stays '-' for channels that did not finish.
    """
    job = _tracking_prepare(fid, channel, settings, signal, pilot_fields)
    if not job.active:
        return job.results, channel
    fields, done, status = fid.track(job.p, job.inits, device_loop=device_loop)   # device_loop: gc_track_device (include/gnsscorr.h)
    return _tracking_finish(job, fields, done, status)


def tracking_file(fid: Engine, path: str, channel, settings, window_samples: int, signal: str = "GPS_L1CA", pilot_fields: str | None = None):
    """tracking(fid, channel, settings) on a record FILE that need not fit the device: at most 2 * window_samples samples are
    resident at any time (include/gnsscorr.h gc_track_file: two alternating device windows, the next one read and uploaded
    while the current one is tracked).  The reference freads block by block (tracking.m:226-245) and so handles any file
    length; results are identical to tracking() on the fully loaded record.  settings.fileType / dataType / the package's
    sample order say how the file is laid out, as in postProcessing.m:59-96."""
    job = _tracking_prepare(fid, channel, settings, signal, pilot_fields)
    if not job.active:
        return job.results, channel
    dtype = L.GC_I16 if str(settings.dataType) == "int16" else L.GC_I8
    # GLONASS front ends deliver Q first (GLO_GL1/include/tracking.m:227)
    layout = L.GC_REAL if settings.fileType == 1 else (L.GC_QI if signal.startswith("GLO_") else L.GC_IQ)
    fid.set_sampling_freq(settings.samplingFreq)
    fields, done, status = fid.track_file(path, job.p, job.inits, int(window_samples), dtype=dtype, layout=layout)
    return _tracking_finish(job, fields, done, status)


def tracking_multi(calls, device_loop: bool = False, pilot_fields: str | None = None):
    """Several packages' tracking() at once (BASELINE config 5, include/gnsscorr.h gc_track_multi):
    calls = [(fid, channel, settings, signal), ...] with one Engine per call - engines that read the same record share it
    with Engine.share_if.  Returns [(trackResults, channel), ...] in call order, each exactly what tracking() returns."""
    jobs = [_tracking_prepare(*c, pilot_fields=pilot_fields) for c in calls]
    live = [j for j in jobs if j.active]
    got = Engine.track_multi([(j.fid, j.p, j.inits) for j in live], device_loop=device_loop) if live else []
    out = []
    it = iter(got)
    for j in jobs:
        out.append(_tracking_finish(j, *next(it)) if j.active else (j.results, j.channel))
    return out
