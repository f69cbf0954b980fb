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
from .settings import skip_samples
from . import codes


def _round(x: float) -> int:
    return int(math.floor(x + 0.5))


_INDEX_CACHE: dict = {}


def _sample_index(n: int, ts: float, tc: float, start_at_one: bool, last: int | None, first_one: bool = False) -> np.ndarray:
    """0-based ceil(ts*(k)/tc) - 1 for k = 1..n (start_at_one) or 0..n-1, with the make*Table.m end fixes.  The index vector depends
    on the rates only, not on the PRN: kept per (n, ts, tc, ...) - the reference recomputes it in every make*Table call."""
    key = (n, ts, tc, start_at_one, last, first_one)
    idx = _INDEX_CACHE.get(key)
    if idx is None:
        k = np.arange(1, n + 1) if start_at_one else np.arange(0, n)
        idx = np.ceil(ts * k / tc).astype(np.int64)
        if first_one:
            idx[0] = 1
        if last is not None:
            idx[-1] = last
        idx -= 1
        idx.setflags(write=False)
        if len(_INDEX_CACHE) > 32:
            _INDEX_CACHE.clear()
        _INDEX_CACHE[key] = idx
    return idx


def _sampled(code: np.ndarray, n: int, ts: float, tc: float, start_at_one: bool, last: int | None, first_one: bool = False):
    """code(ceil(ts*(k)/tc)) for k = 1..n (start_at_one) or 0..n-1, MATLAB 1-based indices into `code`."""
    return code[_sample_index(n, ts, tc, start_at_one, last, first_one)]


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


def _first_maximum(v: np.ndarray):
    """The reference walks the (carrier, bin) grid in order and keeps a value only if it EXCEEDS the largest so far, starting from 0
    (BDS/B1I acquisition.m:87-122, GPS_L2C acquisition.m:46-66), skipping the last bin of every carrier but the first: the winner
    is the first position, in that order, of the largest value - if that is above 0.  v: [carriers, bins]; returns (carrier, bin)
    0-based or None."""
    w = np.array(v, dtype=np.float64)
    w[1:, -1] = -np.inf
    i = int(np.argmax(w))                                                          # first occurrence of the maximum, row-major = scan order
    if not w.flat[i] > 0.0:
        return None
    return divmod(i, w.shape[1])


def _batched(engine, chips, index, weights, rule, exclude=0, period=1):
    """The whole PRN list in one library call (gc_acq_shift_search_batch): chips int8 [nprn, narms, chips per code] and the ONE index
    vector that samples every code (the zero padding up to the block length happens on the device).  None: GC_ACQ_SHIFT_PER_PRN=1 (the
    PRN loop with its two read-backs per PRN, kept for A/B runs and as the path of block lengths without specialised passes) or the
    library said so."""
    import os
    if os.environ.get("GC_ACQ_SHIFT_PER_PRN"):
        return None
    return engine.acq_shift_search_batch(chips, weights, rule, exclude, period, sample_index=index)


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
        first_sample = skip_samples(settings)
    src, first_sample, _ = engine.acq_input(first_sample, settings.samplingFreq)   # int16 / Q-I / real records: float copy on the device
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
                              n=spb, n_signals=2, n_carriers=nshifts, n_bins=nbins, n_arms_max=1, source=src)
    engine.acq_shift_prepare(p)
    acq = SimpleNamespace(carrFreq=np.zeros(58), codePhase=np.zeros(58), peakMetric=np.zeros(58))
    chip = _round(fs / settings.codeFreqBasis)                                     # :139
    prns = list(settings.acqSatelliteList)

    def local_code(prn):
        ca = codes.generateCAcode53(prn).astype(np.int8)
        table = _sampled(np.concatenate([ca, ca]), spc2, ts, 1.0 / settings.codeFreqBasis, True, ncodes * 2046)
        return np.concatenate([table, np.zeros(spb // ncodes, dtype=np.int8)])     # :86

    def record(prn, row, code_phase, max_peak, second):
        carrier, rest = divmod(row, 2 * nbins)
        freq_shift, bin_idx = carrier + 1, rest % nbins + 1
        acq.peakMetric[prn - 1] = max_peak / second                                # :160
        if max_peak / second > settings.acqThreshold:                              # :163
            acq.codePhase[prn - 1] = code_phase
            acq.carrFreq[prn - 1] = init_freq - freq_res * (bin_idx - 1) + (freq_res / nshifts) * (freq_shift - 1)  # :168

    picks = _batched(engine, np.stack([np.tile(codes.generateCAcode53(prn).astype(np.int8), 2)[None, :] for prn in prns]),
                     _sample_index(spc2, ts, 1.0 / settings.codeFreqBasis, True, ncodes * 2046), None, L.GC_SHIFT_PICK_SEQUENTIAL_PAIRS, chip, spb // nblocks)
    if picks is not None:
        # :87-122 (which (carrier, block, bin) wins), :126 (first maximum of that row) and :141-156 (second peak) inside the call
        for prn, pk in zip(prns, picks):
            if pk.row >= 0:
                record(prn, pk.row, pk.code_phase + 1, float(pk.peak), float(pk.second_peak))
        return acq
    for prn in prns:
        rmax, _ = engine.acq_shift_search(local_code(prn)[None, :])
        rmax = rmax.reshape(nshifts, 2, nbins)
        # :87-122, the sequential rule: a (carrier, bin) is taken when one of its two blocks' maxima exceeds the largest so far, and
        # then the first block only if it is the larger of the two
        p1, p2 = rmax[:, 0, :], rmax[:, 1, :]
        win = _first_maximum(np.maximum(p1, p2))
        if win is None:
            continue
        row = (win[0] * 2 + (0 if p1[win] > p2[win] else 1)) * nbins + win[1]
        corr = engine.acq_shift_row(row)
        code_phase = int(np.argmax(corr)) + 1                                      # :126
        record(prn, row, code_phase, float(corr[code_phase - 1]), _second_peak(corr, code_phase, chip, spb // nblocks))
    return acq


# ---------------------------------------------------------------------------------------------
# GPS L2C
# ---------------------------------------------------------------------------------------------
def acquisition_L2C(engine, settings, first_sample: int | None = None):
    """acqResults = acquisition(longSignal, settings) of GPS/GPS_L2C/include/acquisition.m: CM search over a
    40-ms block (Nblocks = 2), then — with pilotTRKflag — which of the 75 CL segments the CM period found lies in."""
    if first_sample is None:
        first_sample = skip_samples(settings)
    src, first_sample, _ = engine.acq_input(first_sample, settings.samplingFreq)   # int16 / Q-I / real records: float copy on the device
    nblocks = 2                                                                    # :13
    fs = settings.samplingFreq
    spc = _round(fs / (settings.codeFreqBasis / settings.codeLength))              # :14-15
    chip = _round(fs / settings.codeFreqBasis)                                     # :16 samplesPerChip
    spb = spc * nblocks                                                            # :17
    ts = 1.0 / fs
    freq_res = fs / spb                                                            # :22
    nbins = _round(settings.acqSearchBand * 1e3 / freq_res) + 1                    # :23
    nshifts = int(freq_res / settings.acqStep)                                     # :25
    init_freq = settings.IF + (settings.acqSearchBand / 2) * 1000                  # :33
    p = L.gc_acq_shift_params(sampling_freq=fs, carrier_f0=init_freq, carrier_step=-(freq_res / nshifts), first_sample=first_sample,
                              n=spb, n_signals=1, n_carriers=nshifts, n_bins=nbins, n_arms_max=1, source=src)
    engine.acq_shift_prepare(p)
    acq = SimpleNamespace(carrFreq=np.zeros(32), codePhase=np.zeros(32), peakMetric=np.zeros(32), CLCodePhase=np.zeros(0))
    tc = 1.0 / (settings.codeFreqBasis * 2)
    prns = list(settings.acqSatelliteList)

    def local_code(prn):
        cm = codes.generateCMcode(prn, int(settings.codeLength))
        table = _sampled(cm, spc, ts, tc, False, int(settings.codeLength) * 2, first_one=True)   # makeCMTable.m
        return np.concatenate([table, np.zeros(spc, dtype=np.int8)])                # :44

    # :46-66 (which (carrier, bin) wins), :72 (first maximum of that row) and :77-91 (second peak): inside the one call for all PRNs,
    # or PRN by PRN with the row maxima and the winning row read back each time
    picks = _batched(engine, np.stack([codes.generateCMcode(prn, int(settings.codeLength)).astype(np.int8)[None, :] for prn in prns]),
                     _sample_index(spc, ts, tc, False, int(settings.codeLength) * 2, first_one=True), None, L.GC_SHIFT_PICK_SEQUENTIAL, chip, spb // nblocks)
    for k, prn in enumerate(prns):
        if picks is not None:
            if picks[k].row < 0:
                continue
            best = divmod(int(picks[k].row), nbins)
            code_phase, max_peak, second = int(picks[k].code_phase) + 1, float(picks[k].peak), float(picks[k].second_peak)
        else:
            rmax, _ = engine.acq_shift_search(local_code(prn)[None, :])
            best = _first_maximum(rmax.reshape(nshifts, nbins))                    # :46-66
            if best is None:
                continue
            corr = engine.acq_shift_row(best[0] * nbins + best[1])
            code_phase = int(np.argmax(corr)) + 1                                  # :72
            max_peak = float(corr[code_phase - 1])
            second = _second_peak(corr, code_phase, chip, spb // nblocks)
        freq_shift, bin_idx = best[0] + 1, best[1] + 1
        acq.peakMetric[prn - 1] = max_peak / second                                # :94
        if max_peak / second > settings.acqThreshold:                              # :97
            f = init_freq - freq_res * (bin_idx - 1) - (freq_res / nshifts) * (freq_shift - 1)   # :101
            acq.carrFreq[prn - 1] = f
            acq.codePhase[prn - 1] = code_phase
            if getattr(settings, "pilotTRKflag", 0) == 1:                          # :140-166, 75 short correlations
                # sig - mean(sig), wiped with the carrier and with each of the 75 CL segments sampled like the CM table: one
                # launch, the segments as 75 replicas of one entry per sample (gc_fine_params.code_freq = 0)
                mean, _ = engine.acq_signal_stats(first_sample + code_phase - 1, spc, source=src)
                cl = codes.generateCLcode(prn, int(settings.CLCodeLength))
                idx = np.ceil(ts * np.arange(spc) / tc).astype(np.int64)
                idx[0] = 1
                idx[-1] = int(settings.codeLength) * (1 if settings.acqCohT <= 10 else 2)
                windows = np.stack([cl[idx - 1 + int(settings.codeLength) * 2 * ind] for ind in range(75)]).astype(np.int8)
                fp = L.gc_fine_params(sampling_freq=fs, code_freq=0.0, f0=f, fstep=0.0, first_sample=first_sample + code_phase - 1, spc=spc,
                                      ncodes=1, nbins=1, code_len=spc, index_offset=0, source=src, dc_re=mean.real, dc_im=mean.imag)
                power = np.abs(engine.acquire_fine_sums_batch(fp, windows, np.full(75, fp.first_sample), np.full(75, f))[:, 0, 0])
                if acq.CLCodePhase.shape[0] < prn:       # the field is created by this assignment and grows with it (GPS_L2C acquisition.m:165):
                    acq.CLCodePhase = np.concatenate([acq.CLCodePhase, np.zeros(prn - acq.CLCodePhase.shape[0])])   # numel = highest PRN found
                acq.CLCodePhase[prn - 1] = int(np.argmax(power)) + 1
    return acq


# ---------------------------------------------------------------------------------------------
# BDS B1C
# ---------------------------------------------------------------------------------------------
def _b1c_table(code: np.ndarray, settings, spc: int) -> np.ndarray:
    """makeDataTable.m / makePilotTable.m: BOC(1,1) half-chip code sampled at ceil(ts*(1:spc)/tc), first index
    forced to 1, last to 2*codeLength."""
    return _sampled(code, spc, 1.0 / settings.samplingFreq, 1.0 / settings.codeFreqBasis / 2, True,
                    int(settings.codeLength) * 2, first_one=True)


def _acquisition_B1C_conditioned(engine, original, first_sample: int, n_long: int | None):
    """BDS/B1C/include/acquisition.m with settings.resamplingflag: the conditioning block (:50-122: BW = 9 MHz, band edges
    widened by 0.002) runs first and every length below follows the new rate, so the (10 + acqCohT)-ms transform is no longer a
    size the radix plan takes.  The search then runs carrier by carrier (gc_acq_params.block_len / code_samples / n_bins /
    arm_weight): circshift(IQfreqDom, k) is the carrier moved by k*fs/N, and the N-point circular correlation with the replica's
    samplesXmsLen samples is the linear one of the block followed by a repeat of its first samplesXmsLen samples.  Fine stage,
    sigPower and the end-of-record rule as in acquisition_B1C; code phase and frequency mapped back as :280-296."""
    import copy
    if n_long is None:
        n_long = int(engine.if_buffer()[1]) - int(first_sample)
    new_fs, new_if, n_cond = engine.acq_condition(original.samplingFreq, original.IF, 9e6, first_sample, n_long, band_margin=0.002)   # :61-69
    settings = copy.copy(original)
    settings.samplingFreq, settings.IF = new_fs, new_if
    fs = new_fs
    spc = _round(fs / (settings.codeFreqBasis / settings.codeLength))
    xlen = _round(spc / 10 * settings.acqCohT)
    n = _round(spc / 10 * (10 + settings.acqCohT))
    nbins = _round(settings.acqSearchBand * 2 / settings.acqStep) + 1
    pilot = getattr(settings, "pilotACQflag", 0) == 1
    fine_step = 25
    nfine = _round(settings.acqStep / 25) * 2 + 1
    init_freq = settings.IF + settings.acqSearchBand
    # search_step: one position of circshift (:170-176); selFreq below is labelled with acqStep like the reference's
    p = L.gc_acq_params(sampling_freq=fs, code_freq_basis=settings.codeFreqBasis, code_length=settings.codeLength, intermediate_freq=settings.IF,
                        search_band=settings.acqSearchBand, search_step=fs / n, non_coh_time=1, source=1, first_sample=0)
    p.block_len, p.code_samples, p.n_bins = n, xlen, nbins
    if pilot:
        p.arm_weight[0], p.arm_weight[1] = math.sqrt(11) / math.sqrt(40), math.sqrt(29) / math.sqrt(40)   # :186-187
    nmax = max(settings.acqSatelliteList)
    acq = SimpleNamespace(carrFreq=np.zeros(nmax), codePhase=np.zeros(nmax), peakMetric=np.zeros(nmax))
    prns = list(settings.acqSatelliteList)
    dtabs = {prn: _b1c_table(codes.generateDataBOC11(prn), settings, spc) for prn in prns}
    ptabs = {prn: _b1c_table(codes.generatePilotBOC11(prn), settings, spc) for prn in prns} if pilot else {}
    tables = np.stack([np.stack([dtabs[prn][:xlen]] + ([ptabs[prn][:xlen]] if pilot else [])) for prn in prns])      # [nprn, narms, xlen]
    res = engine.acquire_coarse(p, tables)
    for prn, r in zip(prns, res):
        sel_freq = init_freq - (r.coarse_bin - 1) * settings.acqStep               # :194
        code_phase = int(r.code_phase)
        acq.peakMetric[prn - 1] = r.peak_metric                                    # :199 (sigPower over samplesXmsLen samples, one hop)
        if code_phase + spc - 1 > n_cond:                                          # :232-234
            code_phase -= spc
        if acq.peakMetric[prn - 1] > settings.acqThreshold:
            fp = L.gc_fine_params(sampling_freq=fs, code_freq=0.0, f0=sel_freq + settings.acqStep, fstep=float(fine_step),
                                  first_sample=code_phase - 1, spc=spc, ncodes=1, nbins=nfine, code_len=spc, index_offset=0, source=1)
            tabs = np.stack([dtabs[prn], ptabs[prn]]) if pilot else dtabs[prn][None, :]
            s = np.abs(engine.acquire_fine_sums_batch(fp, tabs, np.full(len(tabs), fp.first_sample), np.full(len(tabs), fp.f0))[:, :, 0])
            fine = (s[0] * 11 + s[1] * 29) / 40 if pilot else s[0]
            f = float(fp.f0 - fine_step * int(np.argmax(fine)))
            f = f if f != 0 else 1                                                 # :253-255
            acq.codePhase[prn - 1] = math.floor((code_phase - 1) / fs * original.samplingFreq) + 1          # :280-284
            doppler = (fs - settings.IF) - f if settings.IF >= fs / 2 else f - settings.IF                  # :288-294
            acq.carrFreq[prn - 1] = doppler + original.IF                                                    # :296
    return acq


def acquisition_B1C(engine, settings, first_sample: int | None = None, n_long: int | None = None):
    """acqResults = acquisition(longSignal, settings) of BDS/B1C/include/acquisition.m (resampling off): one
    (10 + acqCohT)-ms spectrum, Doppler bins as circular shifts, data and pilot BOC(1,1) replicas combined
    sqrt(11):sqrt(29), GLRT-style metric peak/sigPower, then a 25-Hz fine search on one code period.
    n_long = length(longSignal) (the reference pulls the code phase back by one period if too close to its end)."""
    if first_sample is None:
        first_sample = skip_samples(settings)
    if settings.samplingFreq > settings.resamplingThreshold and getattr(settings, "resamplingflag", 0) == 1:
        return _acquisition_B1C_conditioned(engine, settings, first_sample, n_long)
    fs = settings.samplingFreq
    spc = _round(fs / (settings.codeFreqBasis / settings.codeLength))              # :108-109
    xlen = _round(spc / 10 * settings.acqCohT)                                      # :111 samplesXmsLen
    n = _round(spc / 10 * (10 + settings.acqCohT))                                  # :113 len10PlusXms
    ts = 1.0 / fs
    nbins = _round(settings.acqSearchBand * 2 / settings.acqStep) + 1              # :120
    pilot = getattr(settings, "pilotACQflag", 0) == 1
    fine_step = 25                                                                 # :129
    nfine = _round(settings.acqStep / 25) * 2 + 1                                  # :130
    src, first_sample, n_avail = engine.acq_input(first_sample, fs, n_long)         # int16 / Q-I / real records: float copy on the device
    if n_long is None:
        n_long = n_avail
    _, var = engine.acq_signal_stats(first_sample, xlen, source=src)               # sigPower, :138
    sig_power = math.sqrt(var * xlen)
    init_freq = settings.IF + settings.acqSearchBand                               # :141
    p = L.gc_acq_shift_params(sampling_freq=fs, carrier_f0=init_freq, carrier_step=0.0, first_sample=first_sample,
                              n=n, n_signals=1, n_carriers=1, n_bins=nbins, n_arms_max=2, source=src)
    engine.acq_shift_prepare(p)
    nmax = max(settings.acqSatelliteList)
    acq = SimpleNamespace(carrFreq=np.zeros(nmax), codePhase=np.zeros(nmax), peakMetric=np.zeros(nmax))
    prns = list(settings.acqSatelliteList)
    weights = [math.sqrt(11) / math.sqrt(40), math.sqrt(29) / math.sqrt(40)] if pilot else None   # :186-187
    table = lambda gen, prn: _b1c_table(gen(prn), settings, spc)                   # (makeDataTable.m / makePilotTable.m; sampled where it is needed)
    pad = np.zeros(n - xlen, dtype=np.int8)

    def local_codes(prn):                                                          # :155-156: [table(1:samplesXmsLen) zeros]
        gens = [codes.generateDataBOC11, codes.generatePilotBOC11] if pilot else [codes.generateDataBOC11]
        return np.stack([np.concatenate([table(g, prn)[:xlen], pad]) for g in gens])

    index = _sample_index(spc, 1.0 / fs, 1.0 / settings.codeFreqBasis / 2, True, int(settings.codeLength) * 2, first_one=True)[:xlen]
    picks = _batched(engine, np.stack([np.stack([codes.generateDataBOC11(prn)] + ([codes.generatePilotBOC11(prn)] if pilot else [])) for prn in prns]),
                     index, weights, L.GC_SHIFT_PICK_GLOBAL)
    for k, prn in enumerate(prns):
        if picks is not None:
            # :193 max(max(results,[],2)) and [peakSize, codePhase] = max(max(results)): the first row holding the largest row
            # maximum, the first column holding the global maximum
            bin_idx, peak, code_phase = int(picks[k].row) + 1, float(picks[k].peak), int(picks[k].code_phase) + 1
        else:
            rmax, rarg = engine.acq_shift_search(local_codes(prn), weights)
            bin_idx = int(np.argmax(rmax)) + 1                                     # :193
            peak = float(rmax.max())
            code_phase = int(rarg[rmax == rmax.max()].min()) + 1
        sel_freq = init_freq - (bin_idx - 1) * settings.acqStep                    # :194
        acq.peakMetric[prn - 1] = peak / sig_power                                 # :199
        if code_phase + spc - 1 > n_long:                                          # :232-234
            code_phase -= spc
        if acq.peakMetric[prn - 1] > settings.acqThreshold:
            # one code period against the sampled BOC tables at nfine carriers (:242-250; the reference does not remove the mean
            # here): the tables are replicas of one entry per sample (gc_fine_params.code_freq = 0), data and pilot in one launch
            fp = L.gc_fine_params(sampling_freq=fs, code_freq=0.0, f0=sel_freq + settings.acqStep, fstep=float(fine_step),
                                  first_sample=first_sample + code_phase - 1, spc=spc, ncodes=1, nbins=nfine, code_len=spc,
                                  index_offset=0, source=src)
            tabs = np.stack([table(codes.generateDataBOC11, prn)] + ([table(codes.generatePilotBOC11, prn)] if pilot else []))
            s = np.abs(engine.acquire_fine_sums_batch(fp, tabs, np.full(len(tabs), fp.first_sample), np.full(len(tabs), fp.f0))[:, :, 0])
            fine = (s[0] * 11 + s[1] * 29) / 40 if pilot else s[0]
            f = float(fp.f0 - fine_step * int(np.argmax(fine)))
            acq.carrFreq[prn - 1] = f if f != 0 else 1                             # :253-255
            acq.codePhase[prn - 1] = code_phase
    return acq
