"""Acquisition of the packages that follow GPS L1 C/A's scheme (one FFT pair per
Doppler bin and 1-code hop, data + pilot replicas summed, peak/sigPower metric) and differ in the fine-frequency
stage: GPS L5 (Neuman-Hofman search over 20 codes, GPS_L5C/include/acquisition.m), Galileo E5a (100-chip secondary
code search, 5-Hz bins, GAL_E5a/include/acquisition.m), BDS B2a (non-coherent data + pilot, BDS/B2a/include/
acquisition.m).  Coarse search and the per-code sums of the fine stage run on the GPU (gc_acquire_coarse_multi,
gc_acquire_fine_sums); the hypothesis search over 20-100 complex numbers per bin is restated here."""
from __future__ import annotations

import functools
import math
from types import SimpleNamespace

import numpy as np

from . import _lib as L
from .settings import skip_samples
from . import codes

NH20 = np.array([1, 1, 1, 1, 1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, -1, -1, -1, 1], dtype=np.float64)  # GPS_L5C acquisition.m:131


def _round(x: float) -> int:
    return int(math.floor(x + 0.5))


def make_table(code: np.ndarray, settings) -> np.ndarray:
    """makeL5ITable.m / makeE5aITable.m / makeB2aDataTable.m: code(ceil(ts*(1:spc)/tc)), last index = codeLength."""
    return code[_table_index(float(settings.samplingFreq), float(settings.codeFreqBasis), int(settings.codeLength), False)]


@functools.lru_cache(maxsize=32)
def _table_index(fs: float, fc: float, code_length: int, boc: bool) -> np.ndarray:
    """codeValueIndex of the make*Table.m files (0-based), the same for every PRN and arm of a front end: kept per (fs, fc, L)."""
    spc = _round(fs / (fc / code_length))
    if boc:
        idx = np.ceil((1.0 / fs) * np.arange(1, spc + 1) / (1.0 / fc / 2)).astype(np.int64)
        idx[-1] = code_length * 2
        idx[0] = 1
    else:
        idx = np.ceil((1.0 / fs) * np.arange(1, spc + 1) / (1.0 / fc)).astype(np.int64)
        idx[-1] = code_length
    idx -= 1
    idx.setflags(write=False)
    return idx


@functools.lru_cache(maxsize=64)
def _rolled(sec_bytes: bytes, n: int) -> np.ndarray:
    """Row k = circshift(sec, k): the secondary code at every circular shift, built once per code."""
    sec = np.frombuffer(sec_bytes, dtype=np.float64, count=n)
    m = np.stack([np.roll(sec, k) for k in range(n)])
    m.setflags(write=False)
    return m


def _circular_code_search(sums: np.ndarray, sec: np.ndarray):
    """max over the len(sec) circular shifts of |sum(sumPerCode .* circshift(sec, k))| (GPS_L5C acquisition.m:243-248); all shifts
    in one product with the rolled-code matrix (the loop of 100 np.roll per fine bin was 0.27 of Galileo E5a's 0.28 s per search).
    `sums`: [ncodes] -> float, or [nbins, ncodes] (every fine bin at once) -> [nbins]."""
    sec = np.ascontiguousarray(sec, dtype=np.float64)
    m = _rolled(sec.tobytes(), sec.shape[0])
    if sums.ndim == 2:
        return np.max(np.abs(sums @ m.T), axis=1)
    return float(np.max(np.abs(m @ sums)))


def _split_code_search(sums: np.ndarray, sec: np.ndarray):
    """GAL_E1C acquisition.m:237-245 / BDS B3I acquisition.m:262-270: the secondary code aligned, then every
    circular shift k = 1..len-1 with the sum SPLIT at the possible data-bit edge: |sum(1:k)| + |sum(k+1:end)|.
    `sums`: [ncodes] -> float, or [nbins, ncodes] -> [nbins]."""
    sec = np.ascontiguousarray(sec, dtype=np.float64)
    n = sec.shape[0]
    one = sums.ndim == 1
    x = sums[None, :] if one else sums
    cs = np.cumsum(_rolled(sec.tobytes(), n)[None, :, :] * x[:, None, :], axis=2)    # cs[b, k, j] = sum(t_k[:j + 1]) of bin b
    total = cs[:, :, -1]
    k = np.arange(1, n)
    head = cs[:, k, k - 1]
    out = np.maximum(np.abs(total[:, 0]), np.max(np.abs(head) + np.abs(total[:, 1:] - head), axis=1))
    return float(out[0]) if one else out


def _family_a(engine, settings, first_sample, coarse_codes, fine_codes, ncodes, fine_step, combine, n_results=32,
              table_fn=None, fine_code_freq=None, fine_code_len=None, index_offset=1, bandwidth=None, n_long=None,
              band_margin=0.0, mirror=True):
    """`bandwidth`: BW of the package's input-conditioning block (acquisition.m:46-111 of each package: 2*fc + 0.5 MHz for the
    10.23-Mcps packages, 20.46 / 20.552 MHz for Galileo E5 / E1, 9 MHz for GLONASS), taken when settings.resamplingflag asks for it;
    `n_long` = length(longSignal) for that case; `band_margin`: the 0.002 by which GPS_L5C / BDS_B2a widen fir1's normalised band edges
    (GPS_L5C acquisition.m:69); `mirror`: whether the package maps an IF in the upper half of the new Nyquist band back as
    newFs - IF (GPS_L5C :304-309, BDS_B2a :301-306) or always as carrFreq - IF (GAL_E5a :292, GAL_E5b :240, GAL_E1C :281, BDS_B3I :295)."""
    import copy
    from .receiver import _acq_params
    if first_sample is None:
        first_sample = skip_samples(settings)
    original = settings
    flag = getattr(settings, "resamplingflag", getattr(settings, "resamplingFlag", 0))
    resampled = settings.samplingFreq > settings.resamplingThreshold and flag == 1
    if resampled:
        if bandwidth is None:
            raise NotImplementedError("this package's acquisition has no input-conditioning block")
        if n_long is None:
            n_long = int(engine.if_buffer()[1]) - int(first_sample)
        new_fs, new_if, _ = engine.acq_condition(settings.samplingFreq, settings.IF, bandwidth, first_sample, n_long,
                                                 band_margin=band_margin)
        settings = copy.copy(settings)
        settings.samplingFreq, settings.IF = new_fs, new_if
        first_sample = 0
    src = 1
    if not resampled:
        src, first_sample, _ = engine.acq_input(first_sample, settings.samplingFreq, n_long)    # int16 / Q-I / real records: float copy
    prns = list(settings.acqSatelliteList)
    acq = SimpleNamespace(carrFreq=np.zeros(n_results), codePhase=np.zeros(n_results), peakMetric=np.zeros(n_results))
    p = _acq_params(settings, first_sample)
    p.source = src
    table_fn = table_fn or make_table
    tables = np.stack([np.stack([table_fn(c, settings) for c in coarse_codes(prn)]) for prn in prns])   # [nprn, narms, spc]
    res = engine.acquire_coarse(p, tables)
    spc = tables.shape[-1]
    nfine = _round(settings.acqSearchStep / fine_step) + 1 if fine_step else 0
    found = []
    for prn, r in zip(prns, res):
        acq.peakMetric[prn - 1] = r.peak_metric
        if r.peak_metric > settings.acqThreshold:
            acq.codePhase[prn - 1] = r.code_phase
            if not fine_step:                      # GAL_E5b acquisition.m:227: the coarse bin is the answer
                acq.carrFreq[prn - 1] = r.coarse_freq
                continue
            found.append((prn, r))
    if found:
        # the per-code sums of every detection (and of both code arms where the package has two) in ONE launch and one read-back: a
        # call per detection and arm was a launch + a synchronisation each, 0.15 - 0.3 ms of searches that take 3 - 6 ms
        fp = L.gc_fine_params(sampling_freq=settings.samplingFreq, code_freq=fine_code_freq or settings.codeFreqBasis,
                              f0=0.0, fstep=fine_step, first_sample=0, spc=spc, ncodes=ncodes, nbins=nfine,
                              code_len=int(fine_code_len or settings.codeLength), index_offset=index_offset, source=src)
        arms = [list(fine_codes(prn)) for prn, _ in found]
        codes_all = np.stack([c for a in arms for c in a])
        firsts = [first_sample + r.code_phase - 1 for (_, r), a in zip(found, arms) for _c in a]
        f0s = [r.coarse_freq + settings.acqSearchStep / 2 for (_, r), a in zip(found, arms) for _c in a]
        sums_all = engine.acquire_fine_sums_batch(fp, codes_all, firsts, f0s)        # [sum of arms, nfine, ncodes]
        k = 0
        for (prn, r), a in zip(found, arms):
            sums = [sums_all[k + i] for i in range(len(a))]                          # each [nfine, ncodes]
            k += len(a)
            fine = np.asarray(combine(prn, sums), dtype=np.float64)                  # the hypothesis search of every fine bin at once
            assert fine.shape == (nfine,), fine.shape
            f = (r.coarse_freq + settings.acqSearchStep / 2) - fine_step * int(np.argmax(fine))
            acq.carrFreq[prn - 1] = f if f != 0 else 1
    if resampled:                                  # back to the record's rate and IF (GPS_L5C acquisition.m:293-305)
        for prn in prns:
            if acq.carrFreq[prn - 1] != 0:
                acq.codePhase[prn - 1] = math.floor((acq.codePhase[prn - 1] - 1) / settings.samplingFreq * original.samplingFreq) + 1
                if mirror and settings.IF >= settings.samplingFreq / 2:
                    doppler = (settings.samplingFreq - settings.IF) - acq.carrFreq[prn - 1]
                else:
                    doppler = acq.carrFreq[prn - 1] - settings.IF
                acq.carrFreq[prn - 1] = doppler + original.IF
    return acq


def acquisition_L5(engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """GPS/GPS_L5C/include/acquisition.m: I5 + Q5 coarse search, fine stage on the Q5 pilot over 20 codes with the
    20-bit Neuman-Hofman code tried at every circular shift (:228-252)."""
    return _family_a(engine, settings, first_sample,
                     lambda prn: [codes.generateL5Icode(prn), codes.generateL5Qcode(prn)],
                     lambda prn: [codes.generateL5Qcode(prn)], 20, 25.0,
                     lambda prn, s: _circular_code_search(s[0], NH20), bandwidth=settings.codeFreqBasis * 2 + 0.5e6, n_long=n_long,   # BW: :62
                     band_margin=0.002)                                                                                                 # :69


def acquisition_E5a(engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """GAL/GAL_E5a/include/acquisition.m: E5a-I + E5a-Q primary codes in the coarse search, fine stage on the pilot over
    100 codes in 5-Hz bins with the PRN's CS100 secondary code at every circular shift."""
    return _family_a(engine, settings, first_sample,
                     lambda prn: [codes.generateE5aIcode(prn, 1), codes.generateE5aQcode(prn, 1)],
                     lambda prn: [codes.generateE5aQcode(prn, 1)], 100, 5.0,
                     lambda prn, s: _circular_code_search(s[0], codes.generateE5aQ_secondary(prn).astype(np.float64)), n_results=50,   # acquisition.m:139 zeros(1, 50)
                     bandwidth=20.46e6, n_long=n_long, mirror=False)                                                                   # BW: :57


def acquisition_B2a(engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """BDS/B2a/include/acquisition.m: data + pilot coarse search; fine stage = sum over max(10, acqNonCohTime) codes of
    |per-code sum| of both components (no secondary-code hypothesis needed)."""
    ncodes = max(10, int(settings.acqNonCohTime))                                    # :156
    return _family_a(engine, settings, first_sample,
                     lambda prn: [codes.generateB2aDataCode(prn), codes.generateB2aPilotCode(prn)],
                     lambda prn: [codes.generateB2aDataCode(prn), codes.generateB2aPilotCode(prn)], ncodes, 25.0,
                     lambda prn, s: np.sum(np.abs(s[0]), axis=-1) + np.sum(np.abs(s[1]), axis=-1),
                     n_results=int(max(settings.acqSatelliteList)),   # BDS/B2a acquisition.m:139 zeros(1, max(acqSatelliteList))
                     bandwidth=settings.codeFreqBasis * 2 + 0.5e6, n_long=n_long, band_margin=0.002)   # BW: :60, wp: :64


def acquisition_E5b(engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """GAL/GAL_E5b/include/acquisition.m: E5b-I + E5b-Q coarse search in 60-Hz bins and no fine stage (:227)."""
    return _family_a(engine, settings, first_sample,
                     lambda prn: [codes.generateE5bIcode(prn, 1), codes.generateE5bQcode(prn, 1)], None, 0, 0.0, None, n_results=50,   # acquisition.m:138 zeros(1, 50)
                     bandwidth=20.46e6, n_long=n_long, mirror=False)


def _b3i_combine(prn, s):
    """BDS/B3I/include/acquisition.m:252-271: GEO satellites (PRN 1-5, 59-63; 2-ms D2 symbols) — the better of the two
    pairings of adjacent codes; MEO/IGSO (PRN 6-58) — NH20 with the split-sum search."""
    x = s[0]                                         # [nbins, 20] (or [20])
    if 1 <= prn <= 5 or 59 <= prn <= 63:
        lead = x.shape[:-1]
        p1 = np.sum(np.abs(x.reshape(lead + (10, 2)).sum(axis=-1)), axis=-1)
        p2 = np.sum(np.abs(x[..., [0, 19]]), axis=-1) + np.sum(np.abs(x[..., 1:19].reshape(lead + (9, 2)).sum(axis=-1)), axis=-1)
        return np.maximum(p1, p2)
    return _split_code_search(x, NH20)


def acquisition_B3I(engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """BDS/B3I/include/acquisition.m: single-component coarse search, 20-code fine stage indexed (0 : 20*spc-1)."""
    return _family_a(engine, settings, first_sample, lambda prn: [codes.generateB3Icode(prn)], lambda prn: [codes.generateB3Icode(prn)],
                     20, 25.0, _b3i_combine, n_results=63, index_offset=0, bandwidth=settings.codeFreqBasis * 2 + 0.5e6, n_long=n_long,
                     mirror=False)


E1C_SECONDARY = np.array([1, 1, -1, -1, -1, 1, 1, 1, 1, 1, 1, 1, -1, 1, -1, 1, -1, -1, 1, -1, -1, 1, 1, -1, 1], dtype=np.float64)  # GAL_E1C acquisition.m:138


def _make_boc_table(code: np.ndarray, settings) -> np.ndarray:
    """makeE1BTable.m:43-55: half-chip code sampled at ceil(ts*(1:spc)/(tc/2)), first index forced to 1, last to 2L."""
    return code[_table_index(float(settings.samplingFreq), float(settings.codeFreqBasis), int(settings.codeLength), True)]


def acquisition_E1C(engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """GAL/GAL_E1C/include/acquisition.m: E1-B + E1-C BOC(1,1) replicas (4-ms codes, 144 000-point transforms), fine
    stage on the pilot over 25 codes in 10-Hz bins with the 25-chip secondary code and the split-sum search."""
    return _family_a(engine, settings, first_sample,
                     lambda prn: [codes.generateE1Bcode(prn), codes.generateE1Ccode(prn)], lambda prn: [codes.generateE1Ccode(prn)],
                     25, 10.0, lambda prn, s: _split_code_search(s[0], E1C_SECONDARY), n_results=50, table_fn=_make_boc_table,
                     fine_code_freq=settings.codeFreqBasis * 2, fine_code_len=int(settings.codeLength) * 2, index_offset=0,
                     bandwidth=20.552e6, n_long=n_long, mirror=False)


# ---------------------------------------------------------------------------------------------
# GLONASS L1OF (FDMA: one code, one search band per frequency number K)
# ---------------------------------------------------------------------------------------------
def _matlab_colon(a: float, d: float, b: float) -> np.ndarray:
    """a:d:b as MATLAB builds it for non-integer steps: forwards from a, backwards from the snapped end point, mean
    in the middle — floor() of these values decides the sampled GLONASS code (generateCAcode.m:113-118), and
    k*511e3/fs hits exact integers every 12 000 samples."""
    n = int(math.floor((b - a) / d + 0.5))
    tol = 2.0 * np.finfo(np.float64).eps * max(abs(a), abs(b))
    if a + n * d - b > tol:
        n -= 1
    c = a + n * d
    if c - b > -tol:
        c = b
    out = np.empty(n + 1)
    k = np.arange(0, n // 2 + 1)
    out[k] = a + k.astype(np.float64) * d
    out[n - k] = c - k.astype(np.float64) * d
    if n % 2 == 0:
        out[n // 2] = (a + c) / 2.0
    return out


@functools.lru_cache(maxsize=8)
def _glonass_sampled_code(samp_freq: float, num_samples: int) -> np.ndarray:
    code = codes.generateGLOcode()
    step = 511e3 / samp_freq
    s = np.floor(_matlab_colon(0.0, step, num_samples * step - step)).astype(np.int64)
    out = code[np.remainder(s, 511)]
    out.setflags(write=False)
    return out


def glonass_sampled_code(samp_freq: float, num_samples: int) -> np.ndarray:
    """GLO/GLO_GL1/include/generateCAcode.m:112-118 with PRN 0: the 511-chip code sampled at floor(k*511e3/fs).  A constant of the
    rate and the length: kept per (rate, length) - the 40-code replica of the fine stage is 480 000 entries through MATLAB's colon
    rule, a millisecond and a half of every call."""
    return _glonass_sampled_code(float(samp_freq), int(num_samples))


def acquisition_GLO(engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """GLO/GLO_GL1/include/acquisition.m:120-200: for every frequency number K the L1CA scheme around
    IF - freqSpacing*K with the common 511-chip code; fine stage over 40 codes in 25-Hz bins against the 10-ms meander:
    |sum(10 codes) - sum(next 10)| at 20 alignments.  Results are stored at the 1-based index K + 8 of 21-entry arrays (acquisition.m:138-142,200;
    preRun.m:66 reads them back as K = index - 8): element K + 7 here.
    With settings.resamplingflag the input-conditioning block (:50-119, BW = 9 MHz) runs first; the code phase is mapped back
    to the record's rate, the carrier frequency is NOT: the reference assigns the mapped value to a field it spells
    `carrFreqcarrFreq` (:284), which is what comes back here too."""
    import copy
    from .receiver import _acq_params
    if first_sample is None:
        first_sample = skip_samples(settings)
    original = settings
    resampled = settings.samplingFreq > settings.resamplingThreshold and getattr(settings, "resamplingflag", 0) == 1
    if resampled:
        if n_long is None:
            n_long = int(engine.if_buffer()[1]) - int(first_sample)
        new_fs, new_if, _ = engine.acq_condition(settings.samplingFreq, settings.IF, 9e6, first_sample, n_long)    # BW: :57
        settings = copy.copy(settings)
        settings.samplingFreq, settings.IF = new_fs, new_if
        first_sample = 0
    src = 1
    if not resampled:
        src, first_sample, _ = engine.acq_input(first_sample, settings.samplingFreq, n_long)
    spc = _round(settings.samplingFreq / (settings.codeFreqBasis / settings.codeLength))
    acq = SimpleNamespace(carrFreq=np.zeros(21), codePhase=np.zeros(21), peakMetric=np.zeros(21))
    table = glonass_sampled_code(settings.samplingFreq, spc)[None, :]
    code40 = glonass_sampled_code(settings.samplingFreq, spc * 40)
    nfine = _round(settings.acqSearchStep / 25) + 1
    # every frequency number in ONE coarse call where its carrier, IF - freqSpacing*K (:146-147), is a whole number of FFT bins away
    # from IF (562.5 / 437.5 kHz on the 2-ms blocks of the default front end: 1 125 / 875 bins): the rows share the signal spectra and
    # run on the two lanes instead of fourteen one-row calls one after the other; otherwise (GC_E_UNSUPPORTED) row by row as before
    Ks = list(settings.acqSatelliteList)
    coarse = None
    if len(Ks) > 1:
        p_all = _acq_params(settings, first_sample)
        p_all.source = src
        try:
            coarse = engine.acquire_coarse(p_all, np.repeat(table, len(Ks), axis=0), freq_offset=[-settings.freqSpacing * K for K in Ks])
        except L.GnssCorrError as e:
            if e.status != L.GC_E_UNSUPPORTED:
                raise
    found = []
    for ik, K in enumerate(Ks):
        p = _acq_params(settings, first_sample)
        p.source = src
        p.intermediate_freq = settings.IF - settings.freqSpacing * K               # :146-147
        r = coarse[ik] if coarse is not None else engine.acquire_coarse(p, table)[0]
        acq.peakMetric[K + 7] = r.peak_metric
        if r.peak_metric > settings.acqThreshold:
            found.append((K, r))
    if found:
        # the 40-code replica is a sequence sampled by the reference's own rule (one entry per sample: gc_fine_params.code_freq = 0);
        # per-code sums for the 21 bins of every detection in one launch on the GPU, the 20 meander alignments of 40 complex numbers here
        fp = L.gc_fine_params(sampling_freq=settings.samplingFreq, code_freq=0.0, f0=0.0, fstep=25.0, first_sample=0, spc=spc, ncodes=40,
                              nbins=nfine, code_len=40 * spc, index_offset=0, source=src)
        f0s = [r.coarse_freq + settings.acqSearchStep / 2 for _, r in found]
        sums_all = engine.acquire_fine_sums_batch(fp, np.repeat(code40[None, :], len(found), axis=0),
                                                  [first_sample + r.code_phase - 1 for _, r in found], f0s)             # [ndet, nfine, 40]
        for (K, r), f0, sums in zip(found, f0s, sums_all):
            cs = np.concatenate([np.zeros((sums.shape[0], 1), dtype=sums.dtype), np.cumsum(sums, axis=1)], axis=1)         # cs[:, j] = sum(s[:j])
            c = np.arange(20)
            fine = np.max(np.abs(2.0 * cs[:, c + 10] - cs[:, c] - cs[:, c + 20]), axis=1)      # :180-185 |sum(10 codes) - sum(next 10)| at 20 alignments
            acq.carrFreq[K + 7] = float(f0 - 25.0 * int(np.argmax(fine)))
            acq.codePhase[K + 7] = r.code_phase
            if acq.carrFreq[K + 7] == 0:                                                                               # :263-265
                acq.carrFreq[K + 7] = 1
            if resampled:                                                                                              # :267-285
                acq.codePhase[K + 7] = math.floor((r.code_phase - 1) / settings.samplingFreq * original.samplingFreq) + 1
                if settings.IF >= settings.samplingFreq / 2:
                    doppler = (settings.samplingFreq - settings.IF) - acq.carrFreq[K + 7]
                else:
                    doppler = acq.carrFreq[K + 7] - settings.IF
                mapped = getattr(acq, "carrFreqcarrFreq", np.zeros(0))                   # :284: a new field, grown by the assignment
                if mapped.shape[0] < K + 8:
                    mapped = np.concatenate([mapped, np.zeros(K + 8 - mapped.shape[0])])
                mapped[K + 7] = doppler + original.IF
                acq.carrFreqcarrFreq = mapped
    return acq
