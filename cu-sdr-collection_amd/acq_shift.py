"""Acquisition of the packages that realise Doppler bins as circular shifts of ONE signal spectrum
(SURVEY.md §8a row A5): BDS B1I (BDS/B1I/include/acquisition.m), GPS L2C (GPS/GPS_L2C/include/acquisition.m) and
BDS B1C (BDS/B1C/include/acquisition.m).  The transforms run on the GPU (gc_acq_shift_*: signal spectra once, one
shifted product + inverse FFT per PRN, carrier and bin); the per-row maxima come back and the reference's selection
rules — which row wins, second-peak exclusion ranges, thresholds — are restated here on the host, line by line."""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

from . import _lib as L
from . import codes


def _round(x: float) -> int:
    return int(math.floor(x + 0.5))


def _sampled(code: np.ndarray, n: int, ts: float, tc: float, start_at_one: bool, last: int | None, first_one: bool = False):
    """code(ceil(ts*(k)/tc)) for k = 1..n (start_at_one) or 0..n-1, MATLAB 1-based indices into `code`."""
    k = np.arange(1, n + 1) if start_at_one else np.arange(0, n)
    idx = np.ceil(ts * k / tc).astype(np.int64)
    if first_one:
        idx[0] = 1
    if last is not None:
        idx[-1] = last
    return code[idx - 1]


def _second_peak(corr: np.ndarray, code_phase: int, exclude: int, period: int) -> float:
    """max of corr over one code period outside +-exclude samples of the peak (1-based code_phase), with the
    reference's three range cases (BDS/B1I acquisition.m:141-156, GPS_L2C acquisition.m:77-91)."""
    e1, e2 = code_phase - exclude, code_phase + exclude
    if e1 < 2:
        rng = np.arange(e2, period + e1 + 1)
    elif e2 >= period:
        rng = np.arange(e2 - period + 1, e1 + 1)
    else:
        rng = np.concatenate([np.arange(1, e1 + 1), np.arange(e2, period + 1)])
    return float(np.max(corr[rng - 1]))


# ---------------------------------------------------------------------------------------------
# BDS B1I
# ---------------------------------------------------------------------------------------------
def _b1i_step_size(settings, freq_resolution: float, nblocks: int) -> float:
    ss = getattr(settings, "stepSize", None)
    if ss is None or (isinstance(ss, (list, tuple)) and not ss):                    # acquisition.m:46-47
        return 0.5 / (nblocks * settings.codeLength / settings.codeFreqBasis)
    if ss == freq_resolution:                                                      # :48-49
        return ss
    steps = np.arange(1, freq_resolution / 2 + 1e-9, 0.25)                          # :51-59
    steps = steps[np.remainder(freq_resolution, steps) == 0]
    diff = steps - ss
    m = int(np.argmin(np.abs(diff)))
    return float(steps[m - 1] if diff[m] > 0 else steps[m])


def acquisition_B1I(engine, settings, first_sample: int | None = None):
    """acqResults = acquisition(longSignal, settings) of BDS/B1I/include/acquisition.m (resampling off)."""
    if first_sample is None:
        first_sample = int(settings.skipNumberOfBytes)
    ncodes, nblocks = 2, 4                                                         # :34-35
    fs = settings.samplingFreq
    spb = _round(fs / (settings.codeFreqBasis / (nblocks * settings.codeLength)))  # :36-37 samplesPerBlock
    ts = 1.0 / fs
    freq_res = fs / spb                                                            # :43
    nbins = _round(settings.acqSearchBand * 1e3 / freq_res) + 1                    # :44
    step = _b1i_step_size(settings, freq_res, nblocks)
    nshifts = int(freq_res / step)                                                 # :61
    spc2 = _round(fs / (settings.codeFreqBasis / (ncodes * settings.codeLength)))  # makeCaTableDMA.m
    init_freq = settings.IF + (settings.acqSearchBand / 2) * 1000                  # :66
    p = L.gc_acq_shift_params(sampling_freq=fs, carrier_f0=init_freq, carrier_step=freq_res / nshifts, first_sample=first_sample,
                              n=spb, n_signals=2, n_carriers=nshifts, n_bins=nbins, n_arms_max=1)
    engine.acq_shift_prepare(p)
    acq = SimpleNamespace(carrFreq=np.zeros(58), codePhase=np.zeros(58), peakMetric=np.zeros(58))
    chip = _round(fs / settings.codeFreqBasis)                                     # :139
    for prn in settings.acqSatelliteList:
        ca = codes.generateCAcode53(prn).astype(np.int8)
        table = _sampled(np.concatenate([ca, ca]), spc2, ts, 1.0 / settings.codeFreqBasis, True, ncodes * 2046)
        local = np.concatenate([table, np.zeros(spb // ncodes, dtype=np.int8)])     # :86
        rmax, _ = engine.acq_shift_search(local[None, :])
        rmax = rmax.reshape(nshifts, 2, nbins)
        prevmax, best, freq_shift, bin_idx = 0.0, None, 0, 0
        for it in range(nshifts):                                                  # :87-122, the sequential rule
            for b in range(nbins):
                if b == nbins - 1 and it > 0:
                    continue
                p1, p2 = float(rmax[it, 0, b]), float(rmax[it, 1, b])
                if p1 > prevmax or p2 > prevmax:
                    if p1 > p2:
                        prevmax, best = p1, (it, 0, b)
                    else:
                        prevmax, best = p2, (it, 1, b)
                    freq_shift, bin_idx = it + 1, b + 1
        if best is None:
            continue
        corr = engine.acq_shift_row((best[0] * 2 + best[1]) * nbins + best[2])
        code_phase = int(np.argmax(corr)) + 1                                      # :126
        max_peak = float(corr[code_phase - 1])
        second = _second_peak(corr, code_phase, chip, spb // nblocks)
        acq.peakMetric[prn - 1] = max_peak / second                                # :160
        if max_peak / second > settings.acqThreshold:                              # :163
            acq.codePhase[prn - 1] = code_phase
            acq.carrFreq[prn - 1] = init_freq - freq_res * (bin_idx - 1) + (freq_res / nshifts) * (freq_shift - 1)  # :168
    return acq
