"""Acquisition of the packages that follow GPS L1 C/A's scheme (one FFT pair per
Doppler bin and 1-code hop, data + pilot replicas summed, peak/sigPower metric) and differ in the fine-frequency
stage: GPS L5 (Neuman-Hofman search over 20 codes, GPS_L5C/include/acquisition.m), Galileo E5a (100-chip secondary
code search, 5-Hz bins, GAL_E5a/include/acquisition.m), BDS B2a (non-coherent data + pilot, BDS/B2a/include/
acquisition.m).  Coarse search and the per-code sums of the fine stage run on the GPU (gc_acquire_coarse_multi,
gc_acquire_fine_sums); the hypothesis search over 20-100 complex numbers per bin is restated here."""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

from . import _lib as L
from . import codes

NH20 = np.array([1, 1, 1, 1, 1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, -1, -1, -1, 1], dtype=np.float64)  # GPS_L5C acquisition.m:131


def _round(x: float) -> int:
    return int(math.floor(x + 0.5))


def make_table(code: np.ndarray, settings) -> np.ndarray:
    """makeL5ITable.m / makeE5aITable.m / makeB2aDataTable.m: code(ceil(ts*(1:spc)/tc)), last index = codeLength."""
    spc = _round(settings.samplingFreq / (settings.codeFreqBasis / settings.codeLength))
    idx = np.ceil((1.0 / settings.samplingFreq) * np.arange(1, spc + 1) / (1.0 / settings.codeFreqBasis)).astype(np.int64)
    idx[-1] = int(settings.codeLength)
    return code[idx - 1]


def _circular_code_search(sums: np.ndarray, sec: np.ndarray) -> float:
    """max over the len(sec) circular shifts of |sum(sumPerCode .* circshift(sec, k))| (GPS_L5C acquisition.m:243-248)."""
    best = 0.0
    s = sec.copy()
    for _ in range(sec.shape[0]):
        best = max(best, abs(np.sum(sums * s)))
        s = np.roll(s, 1)
    return best


def _family_a(engine, settings, first_sample, coarse_codes, fine_codes, ncodes, fine_step, combine, n_results=32):
    from .receiver import _acq_params
    if first_sample is None:
        first_sample = int(settings.skipNumberOfBytes)
    prns = list(settings.acqSatelliteList)
    acq = SimpleNamespace(carrFreq=np.zeros(n_results), codePhase=np.zeros(n_results), peakMetric=np.zeros(n_results))
    p = _acq_params(settings, first_sample)
    tables = np.stack([np.stack([make_table(c, settings) for c in coarse_codes(prn)]) for prn in prns])   # [nprn, narms, spc]
    res = engine.acquire_coarse(p, tables)
    spc = tables.shape[-1]
    nfine = _round(settings.acqSearchStep / fine_step) + 1
    for prn, r in zip(prns, res):
        acq.peakMetric[prn - 1] = r.peak_metric
        if r.peak_metric > settings.acqThreshold:
            fp = L.gc_fine_params(sampling_freq=settings.samplingFreq, code_freq=settings.codeFreqBasis,
                                  f0=r.coarse_freq + settings.acqSearchStep / 2, fstep=fine_step,
                                  first_sample=first_sample + r.code_phase - 1, spc=spc, ncodes=ncodes, nbins=nfine,
                                  code_len=int(settings.codeLength), index_offset=1)
            sums = [engine.acquire_fine_sums(fp, c) for c in fine_codes(prn)]       # each [nfine, ncodes]
            fine = np.array([combine(prn, [s[k] for s in sums]) for k in range(nfine)])
            f = fp.f0 - fine_step * int(np.argmax(fine))
            acq.carrFreq[prn - 1] = f if f != 0 else 1
            acq.codePhase[prn - 1] = r.code_phase
    return acq


def acquisition_L5(engine, settings, first_sample: int | None = None):
    """GPS/GPS_L5C/include/acquisition.m: I5 + Q5 coarse search, fine stage on the Q5 pilot over 20 codes with the
    20-bit Neuman-Hofman code tried at every circular shift (:228-252)."""
    return _family_a(engine, settings, first_sample,
                     lambda prn: [codes.generateL5Icode(prn), codes.generateL5Qcode(prn)],
                     lambda prn: [codes.generateL5Qcode(prn)], 20, 25.0,
                     lambda prn, s: _circular_code_search(s[0], NH20))


def acquisition_E5a(engine, settings, first_sample: int | None = None):
    """GAL/GAL_E5a/include/acquisition.m: E5a-I + E5a-Q primary codes in the coarse search, fine stage on the pilot over
    100 codes in 5-Hz bins with the PRN's CS100 secondary code at every circular shift."""
    return _family_a(engine, settings, first_sample,
                     lambda prn: [codes.generateE5aIcode(prn, 1), codes.generateE5aQcode(prn, 1)],
                     lambda prn: [codes.generateE5aQcode(prn, 1)], 100, 5.0,
                     lambda prn, s: _circular_code_search(s[0], codes.generateE5aQ_secondary(prn).astype(np.float64)), n_results=36)


def acquisition_B2a(engine, settings, first_sample: int | None = None):
    """BDS/B2a/include/acquisition.m: data + pilot coarse search; fine stage = sum over max(10, acqNonCohTime) codes of
    |per-code sum| of both components (no secondary-code hypothesis needed)."""
    ncodes = max(10, int(settings.acqNonCohTime))                                    # :156
    return _family_a(engine, settings, first_sample,
                     lambda prn: [codes.generateB2aDataCode(prn), codes.generateB2aPilotCode(prn)],
                     lambda prn: [codes.generateB2aDataCode(prn), codes.generateB2aPilotCode(prn)], ncodes, 25.0,
                     lambda prn, s: float(np.sum(np.abs(s[0])) + np.sum(np.abs(s[1]))), n_results=63)
