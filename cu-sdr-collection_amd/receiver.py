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


def acquisition(engine: Engine, settings, first_sample: int | None = None):
    """acqResults = acquisition(longSignal, settings) with longSignal = the IF buffer from
    `first_sample` (default settings.skipNumberOfBytes, as postProcessing.m:74-96 reads it).

    Only the resampling-off path (initSettings.m:93 default) is implemented; the optional
    FIR/decimation front end (acquisition.m:50-111) is out of scope.
    """
    if settings.samplingFreq > settings.resamplingThreshold and settings.resamplingflag == 1:
        raise NotImplementedError("acquisition resampling front end (acquisition.m:50-111) is out of scope")
    if first_sample is None:
        first_sample = int(settings.skipNumberOfBytes)
    prns = list(settings.acqSatelliteList)
    acq = SimpleNamespace(carrFreq=np.zeros(32), codePhase=np.zeros(32), peakMetric=np.zeros(32))
    p = _acq_params(settings, first_sample)
    tables = np.stack([codes.makeCaTable(prn, settings) for prn in prns])
    res = engine.acquire_coarse(p, tables)
    for prn, r in zip(prns, res):
        acq.peakMetric[prn - 1] = r.peak_metric                      # acquisition.m:200
        if r.peak_metric > settings.acqThreshold:                    # :206
            f = engine.acquire_fine_l1ca(p, codes.generateCAcode(prn), r.code_phase, r.coarse_freq)
            acq.carrFreq[prn - 1] = f                                # :254-260
            acq.codePhase[prn - 1] = r.code_phase                    # :256
    return acq


# ---------------------------------------------------------------------------------------------
# preRun
# ---------------------------------------------------------------------------------------------
def preRun(acqResults, settings):
    n_ch = int(settings.numberOfChannels)
    channel = [SimpleNamespace(PRN=0, acquiredFreq=0.0, codePhase=0, status="-") for _ in range(n_ch)]
    order = np.argsort(-np.asarray(acqResults.peakMetric), kind="stable")   # preRun.m:60
    n_found = int(np.sum(np.asarray(acqResults.carrFreq) != 0))
    for ii in range(min(n_ch, n_found)):                                      # :65
        p = int(order[ii])
        channel[ii].PRN = p + 1
        channel[ii].acquiredFreq = float(acqResults.carrFreq[p])
        channel[ii].codePhase = int(acqResults.codePhase[p])
        channel[ii].status = "T"
    return channel


# ---------------------------------------------------------------------------------------------
# tracking
# ---------------------------------------------------------------------------------------------
_REC_FIELDS = ("absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L",
               "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt", "remCodePhase", "remCarrPhase")
_PILOT_FIELDS = ("Pilot_I_E", "Pilot_Q_E", "Pilot_I_P", "Pilot_Q_P", "Pilot_I_L", "Pilot_Q_L")


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
    p.pll_kind = spec.pll_kind
    pilot = spec.pilot_combine if getattr(settings, "pilotTRKflag", 0) == 1 else 0
    p.pilot_combine = pilot
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
    p.skip_samples = int(settings.skipNumberOfBytes)
    p.n_epochs = signals.epochs_to_process(settings)
    if spec.doubled_code:
        # GPS_L2C/include/tracking.m:107-109: spacing and code length in units of the RZ-doubled code; :153 seeks to
        # skipNumberOfBytes + codePhase WITHOUT the usual -1; :261,357-360 CL window bookkeeping
        p.el_spacing = settings.dllCorrelatorSpacing * 2
        p.code_length = settings.codeLength * 2
        p.code_freq_basis = settings.codeFreqBasis * 2
        p.skip_samples = int(settings.skipNumberOfBytes) + 1
        p.table_phase_count = 75 if pilot else 0
    return p


def tracking(fid: Engine, channel, settings, signal: str = "GPS_L1CA", device_loop: bool = False):
    """[trackResults, channel] = tracking(fid, channel, settings) — `signal` selects the reference
    package whose tracking.m is mirrored ("GPS_L1CA": GPS/GPS_L1CA/include/tracking.m;
    "GAL_E1C": GAL/GAL_E1C/include/tracking.m, data + pilot arms, BOC(1,1) half-chip tables).

    Returns (trackResults, channel).  On a short read the reference prints a message and
    returns what it has (tracking.m:241-245); here the partially filled results are returned
    the same way and `trackResults[i].status` stays '-' for channels that did not finish.
    """
    from . import signals
    spec = signals.SIGNALS[signal]
    if settings.fileType != 2 or settings.dataType not in ("schar", "int8", "int16"):
        raise NotImplementedError("tracking(): fileType 2 (I/Q) schar/int16 input only in this build")
    n_ep = signals.epochs_to_process(settings)
    p = track_params(settings, signal)
    pilot = p.pilot_combine != 0
    results = []
    active = []
    for i, ch in enumerate(channel):
        tr = SimpleNamespace(status="-", PRN=0)
        for f in _REC_FIELDS + (_PILOT_FIELDS if pilot else ()):
            setattr(tr, f, np.zeros(n_ep))
        tr.CNo = SimpleNamespace(VSMValue=[], VSMIndex=[])
        results.append(tr)
        if ch.PRN != 0:
            tr.PRN = ch.PRN
            fid.set_channel(i, spec.tables(ch.PRN, settings), index_scale=spec.index_scale, arm_mult=spec.arm_mult,
                            windows=spec.windows)
            active.append(i)
    if not active:
        return results, channel
    inits = []
    for i in active:
        ch = channel[i]
        cf = ch.codeFreq if spec.code_freq_from_channel else settings.codeFreqBasis
        if spec.doubled_code:
            cf = settings.codeFreqBasis * 2                                           # GPS_L2C tracking.m:171
        inits.append(L.gc_channel_init(channel=i, prn=ch.PRN, acquired_freq=ch.acquiredFreq,
                                       code_freq=cf, code_phase=int(ch.codePhase),
                                       table_phase=int(getattr(ch, "CLCodePhase", 0)) if (spec.doubled_code and pilot) else 0))
    fields, done, status = fid.track(p, inits, device_loop=device_loop)   # device_loop: gc_track_device (include/gnsscorr.h)
    # B2a / B1C estimate C/N0 with Calc_CNo_PLD every settings.CNoInterval epochs (below); the other packages with CNoVSM
    pld = int(getattr(settings, "CNoInterval", 0))
    cno = None if pld else getattr(settings, "CNo", None)
    vsm = int(cno.VSMinterval) if cno is not None else 0
    for k, i in enumerate(active):
        tr = results[i]
        for f in _REC_FIELDS + (_PILOT_FIELDS if pilot else ()):
            getattr(tr, f)[:] = fields[f][k]
        n_done = int(done[k])
        if spec.doubled_code:
            # GPS_L2C tracking.m:226,250,376,382-383: what the reference RECORDS is in single-code units, and
            # absoluteSample is pushed back by the code-phase remainder expressed in samples
            step = tr.codeFreq[:n_done] / settings.samplingFreq
            tr.absoluteSample[:n_done] = tr.absoluteSample[:n_done] + 1 - tr.remCodePhase[:n_done] / step
            for f in ("remCodePhase", "codeFreq", "dllDiscr", "dllDiscrFilt"):
                getattr(tr, f)[:n_done] /= 2
        for loop in (range(vsm, n_done + 1, vsm) if vsm else ()):                     # tracking.m:351-358
            tr.CNo.VSMValue.append(CNoVSM(tr.I_P[loop - vsm:loop], tr.Q_P[loop - vsm:loop], settings.CNo.accTime))
            tr.CNo.VSMIndex.append(loop)
        if pld:
            # BDS/B2a/include/tracking.m:85-92,191-192,409-432 (B1C NB_tracking.m:92-98,397-418, WB_tracking.m:443-461):
            # every CNoInterval epochs, the estimate averaged 0.5/0.5 with the previous one (zeros before the first)
            nrec = n_ep // pld
            combined = "B2a_CNo" if signal.startswith("BDS_B2a") else "B1C_CNo"
            names = ("DataCNo", "DataPLD") + (("PilotCNo", "PilotPLD", combined) if pilot else ())
            for f in names:
                setattr(tr, f, np.zeros(nrec))
            prev = np.zeros(3)
            for loop in range(pld, n_done + 1, pld):
                c, d = Calc_CNo_PLD(tr, settings, loop, straight_pilot=signal.endswith("_WB"))
                k = loop // pld - 1
                tr.DataCNo[k] = c[0] * 0.5 + prev[0] * 0.5
                tr.DataPLD[k] = d[0]
                if pilot:
                    tr.PilotCNo[k] = c[1] * 0.5 + prev[1] * 0.5
                    getattr(tr, combined)[k] = c[2] * 0.5 + prev[2] * 0.5
                    tr.PilotPLD[k] = d[1]
                prev = c
        if n_done == n_ep:
            tr.status = channel[i].status                                             # tracking.m:365
    if status == L.GC_E_RANGE:
        print("Not able to read the specified number of samples  for tracking, exiting!")
    return results, channel
