"""CPU oracle (TEST INFRASTRUCTURE ONLY) — float64 NumPy restatement of the reference hot path.

This file restates, line by line, the arithmetic of the MATLAB reference
(gnsscusdr/CU-SDR-Collection @ 2024_10_08) for the acquisition + tracking hot path.
It is the *checker* for the HIP kernels; it is never imported by the product package
(`cu-sdr-collection_amd/`).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.

PARITY STATUS: pinned against the reference's own source text, executed - not against MATLAB itself.  The reference
ships no tests, golden vectors or fixtures and neither MATLAB nor Octave exists in the build container or on the GPU box.
Round 2 added oracle/mlab, a minimal interpreter for the MATLAB subset of the hot-path files; tests/golden/make_ref_vectors.py
runs the reference's tracking.m / NB_tracking.m / WB_tracking.m / acquisition.m / preRun.m / initSettings.m / generate*.m of
all twelve packages through it (read in place from /root/reference, build container only) and tests/test_ref_vectors.py
holds this file to the stored results: <= 1e-12 relative on every tracking record, identical acquisition indices.  The
interpreter's built-ins (colon, rem, var, max, sort, fft ...) are restatements of MATLAB's documented behaviour, so a
misreading of a BUILT-IN would be common to fixtures and oracle; a misreading of the REFERENCE is caught.  Also: public-ICD
known-answer tests for the code generators, analytic correlator identities, a second independent restatement in C
(`oracle/gnss_oracle.c`) that must agree to the last few ulps, closed-loop lock tests (tests/test_oracle_*.py).

All citations are file:line under /root/reference/.  MATLAB semantics modelled
explicitly: colon-operator construction, `ceil` on exact integers, `rem` keeping the sign
of the dividend, `round` half-away-from-zero, `max` returning the first maximum, `var`
normalised by N-1.
"""
from __future__ import annotations

import cmath
import math
import os
from types import SimpleNamespace

import numpy as np

TWO_PI = 2.0 * math.pi  # MATLAB `2 * pi`

# --------------------------------------------------------------------------------------
# MATLAB built-in semantics
# --------------------------------------------------------------------------------------


def matlab_round(x: float) -> float:
    """MATLAB round(): half away from zero (acquisition.m:116,124)."""
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def colon(a: float, d: float, b: float) -> np.ndarray:
    """MATLAB `a:d:b` for doubles (MathWorks' published colon algorithm).

    The vector is NOT a + k*d for every k: the first half is built forwards from `a`,
    the second half backwards from the (tolerance-snapped) end point `c`, and for an even
    number of intervals the middle element is (a+c)/2.  tracking.m:252-273 relies on it
    (remCodePhase inherits the end-point rule through tcode(blksize)).
    """
    a = float(a)
    d = float(d)
    b = float(b)
    if d == 0 or (d > 0 and a > b) or (d < 0 and a < b):
        return np.zeros(0)
    tol = 2.0 * np.finfo(np.float64).eps * max(abs(a), abs(b))
    sig = 1.0 if d > 0 else -1.0
    if a == math.floor(a) and d == 1.0:
        n = int(math.floor(b) - a)
    elif a == math.floor(a) and d == math.floor(d):
        n = int(math.floor((b - a) / d))
    else:
        n = int(matlab_round((b - a) / d))
        if sig * (a + n * d - b) > tol:
            n -= 1
    c = a + n * d
    if sig * (c - b) > -tol:
        c = b
    out = np.empty(n + 1, dtype=np.float64)
    k = np.arange(0, n // 2 + 1, dtype=np.float64)
    ki = np.arange(0, n // 2 + 1)
    out[ki] = a + k * d
    out[n - ki] = c - k * d
    if n % 2 == 0:
        out[n // 2] = (a + c) / 2.0
    return out


def matlab_rem(x: float, y: float) -> float:
    """MATLAB rem(x, y): result has the sign of x (tracking.m:283)."""
    return math.fmod(x, y)


# --------------------------------------------------------------------------------------
# C1 — code generators
# --------------------------------------------------------------------------------------

# G2 delay table, generateCAcode.m:42-50 (values are public IS-GPS-200 / SBAS data)
_G2S = [5, 6, 7, 8, 17, 18, 139, 140, 141, 251,
        252, 254, 255, 256, 257, 258, 469, 470, 471, 472,
        473, 474, 509, 512, 513, 514, 515, 516, 859, 860,
        861, 862,
        145, 175, 52, 21, 237, 235, 886, 657,
        634, 762, 355, 1012, 176, 603, 130, 359, 595, 68,
        386]


def generate_ca_code(prn: int) -> np.ndarray:
    """GPS C/A code, 1023 chips of +-1 (generateCAcode.m:42-90).

    Product form of the LFSRs: registers hold +-1, feedback = product of taps
    (G1 taps {3,10}: :65; G2 taps {2,3,6,8,9,10}: :80), all -1 initial state (:61,:76),
    G2 rotated right by g2s(PRN) (:87), CAcode = -(g1.*g2) (:90).
    """
    g2shift = _G2S[prn - 1]
    g1 = np.empty(1023)
    reg = -np.ones(10)
    for i in range(1023):
        g1[i] = reg[9]
        save = reg[2] * reg[9]
        reg[1:10] = reg[0:9].copy()
        reg[0] = save
    g2 = np.empty(1023)
    reg = -np.ones(10)
    for i in range(1023):
        g2[i] = reg[9]
        save = reg[1] * reg[2] * reg[5] * reg[7] * reg[8] * reg[9]
        reg[1:10] = reg[0:9].copy()
        reg[0] = save
    g2 = np.concatenate([g2[1023 - g2shift:], g2[:1023 - g2shift]])
    return -(g1 * g2)


def pad_code(code: np.ndarray) -> np.ndarray:
    """[c(end) c c(1)] (tracking.m:158; GAL_E1C tracking.m:143)."""
    return np.concatenate([[code[-1]], code, [code[0]]])


def samples_per_code(settings) -> int:
    """round(fs / (fc / L)) (acquisition.m:116-117; makeCaTable.m:43-44)."""
    return int(matlab_round(settings.samplingFreq / (settings.codeFreqBasis / settings.codeLength)))


def make_ca_table(prn: int, settings) -> np.ndarray:
    """C2 — sampled C/A code for acquisition (makeCaTable.m:43-67)."""
    spc = samples_per_code(settings)
    ts = 1.0 / settings.samplingFreq
    tc = 1.0 / settings.codeFreqBasis
    ca = generate_ca_code(prn)
    idx = np.ceil((ts * np.arange(1, spc + 1, dtype=np.float64)) / tc).astype(np.int64)
    idx[-1] = 1023
    return ca[idx - 1]


# --------------------------------------------------------------------------------------
# T6 helpers — loop coefficients and C/N0 (host side in the product, restated here)
# --------------------------------------------------------------------------------------


def calc_loop_coef(lbw: float, zeta: float, k: float):
    """Common/calcLoopCoef.m:41-45."""
    wn = lbw * 8 * zeta / (4 * zeta ** 2 + 1)
    tau1 = k / (wn * wn)
    tau2 = 2.0 * zeta / wn
    return tau1, tau2


def calc_loop_coef_carr(settings, variant: str = "a"):
    """Common/calcLoopCoefCarr.m — 3-state carrier filter coefficients.

    Two variants exist in the tree.  "a" (GPS_L5C/Common/calcLoopCoefCarr.m:41-56, shared by
    B1I, B2a, E1C, E5a, GLO, L2C): a3 = b3 = 2, Wn = 1.2*LBW.  "b" (BDS/B1C, B3I, GAL_E5b:
    :41-50): a3 = 1.1, b3 = 2.4, Wn = LBW/0.7845.  Returns (pf3, pf2, pf1) like the source.
    """
    lbw = settings.pllNoiseBandwidth
    t = settings.intTime
    if variant == "a":
        a3, b3 = 2, 2
        wn = 1.2 * lbw
    else:
        a3, b3 = 1.1, 2.4
        wn = lbw / 0.7845
    pf3 = wn ** 3 * t ** 2
    pf2 = a3 * wn ** 2 * t
    pf1 = b3 * wn
    return pf3, pf2, pf1


def cno_vsm(i_p: np.ndarray, q_p: np.ndarray, t: float) -> float:
    """Common/CNoVSM.m:38-47 (var normalised by N-1)."""
    z = i_p ** 2 + q_p ** 2
    zm = np.mean(z)
    zv = np.var(z, ddof=1)
    pav = np.sqrt(complex(zm ** 2 - zv))  # MATLAB sqrt of a negative returns complex
    nv = 0.5 * (zm - pav)
    return float(10 * np.log10(abs((1 / t) * pav / (2 * nv))))


# --------------------------------------------------------------------------------------
# T2-T4 — one correlator block
# --------------------------------------------------------------------------------------


def blksize_for(code_length: float, rem_code_phase: float, code_phase_step: float) -> int:
    """tracking.m:222."""
    return int(math.ceil((code_length - rem_code_phase) / code_phase_step))


def code_ramps(rem: float, step: float, d: float, n: int, r: float):
    """The three index ramps of tracking.m:252-270, scaled by R as in
    GAL_E1C/include/tracking.m:236-262.  Returned in source order early, late, prompt.

    For R == 1 the source has no multiplication; x*1.0 is exact so one expression serves.
    """
    t_e = colon((rem - d) * r, step * r, ((n - 1) * step + rem - d) * r)
    t_l = colon((rem + d) * r, step * r, ((n - 1) * step + rem + d) * r)
    t_p = colon(rem * r, step * r, ((n - 1) * step + rem) * r)
    return t_e, t_l, t_p


def correlate_block(raw: np.ndarray, tables, rem: float, step: float, d: float,
                    carr_freq: float, rem_carr: float, fs: float, code_length: float,
                    r: float = 1.0, arm_mult=None, table_offset=None):
    """One integrate-and-dump block (tracking.m:247-300).

    raw     complex128[N]  (data1 + 1i*data2, tracking.m:233-235; callers swap for GLONASS)
    tables  list of padded code tables, one per arm ([c(end) c c(1)], tracking.m:158)
    arm_mult per-arm extra multiplier applied to the ramp before ceil
            (6 for the BOC(6,1) arm, BDS/B1C/include/WB_tracking.m:293)
    table_offset per-arm index offset (GPS_L2C tracking.m:261)
    Returns (sums[arms,6] ordered I_E,Q_E,I_P,Q_P,I_L,Q_L, rem_code', rem_carr').
    """
    n = raw.shape[0]
    arms = len(tables)
    arm_mult = arm_mult or [1.0] * arms
    table_offset = table_offset or [0] * arms
    t_e, t_l, t_p = code_ramps(rem, step, d, n, r)
    assert t_e.shape[0] == n and t_l.shape[0] == n and t_p.shape[0] == n
    # tracking.m:273 / GAL_E1C tracking.m:268
    rem_code_new = (t_p[n - 1] / r + step) - code_length if r != 1.0 else (t_p[n - 1] + step) - code_length
    # tracking.m:280-287
    time = np.arange(0, n + 1, dtype=np.float64) / fs
    trigarg = ((carr_freq * 2.0 * math.pi) * time) + rem_carr
    rem_carr_new = matlab_rem(float(trigarg[n]), TWO_PI)
    carrsig = np.exp(-1j * trigarg[:n])
    mixed = carrsig * raw
    i_bb = mixed.real
    q_bb = mixed.imag
    sums = np.empty((arms, 6))
    for a in range(arms):
        tab = tables[a]
        m = arm_mult[a]
        off = table_offset[a]
        # MATLAB index ceil(t)+1 (1-based) == ceil(t) (0-based)
        k_e = np.ceil(t_e * m).astype(np.int64) + off
        k_l = np.ceil(t_l * m).astype(np.int64) + off
        k_p = np.ceil(t_p * m).astype(np.int64) + off
        c_e = tab[k_e]
        c_l = tab[k_l]
        c_p = tab[k_p]
        sums[a, 0] = np.sum(c_e * i_bb)
        sums[a, 1] = np.sum(c_e * q_bb)
        sums[a, 2] = np.sum(c_p * i_bb)
        sums[a, 3] = np.sum(c_p * q_bb)
        sums[a, 4] = np.sum(c_l * i_bb)
        sums[a, 5] = np.sum(c_l * q_bb)
    return sums, rem_code_new, rem_carr_new


def raw_from_if(if_bytes: np.ndarray, first_sample: int, n: int, file_type: int = 2,
                swap_iq: bool = False) -> np.ndarray:
    """fread + de-interleave (tracking.m:226-236).  `if_bytes` is the int8/int16 file
    content; first_sample counts complex samples (fileType 2) or real samples (fileType 1)."""
    if file_type == 1:
        return if_bytes[first_sample:first_sample + n].astype(np.float64).astype(np.complex128)
    seg = if_bytes[2 * first_sample:2 * (first_sample + n)].astype(np.float64)
    if swap_iq:  # GLO_GL1/include/tracking.m:227
        return seg[1::2] + 1j * seg[0::2]
    return seg[0::2] + 1j * seg[1::2]


# --------------------------------------------------------------------------------------
# tracking(fid, channel, settings) — GPS L1 C/A closed loop (tracking.m:47-371)
# --------------------------------------------------------------------------------------

TRACK_FIELDS = ("absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P",
                "Q_L", "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt", "remCodePhase",
                "remCarrPhase")


def first_sample(settings, code_phase, int16_branch: bool = True, minus_one: bool = True) -> int:
    """Where a channel's first fread starts, in samples (complex samples for fileType 2), from the reference's fseek:
    schar   fseek(fid, dataAdaptCoeff*(skipNumberOfBytes + codePhase-1))            tracking.m:150-152
    int16   fseek(fid, dataAdaptCoeff*(skipNumberOfBytes + (codePhase-1)*2))        tracking.m:145-148
    One sample is dataAdaptCoeff components of 1 (schar) or 2 (int16) bytes, so the int16 branch starts at sample
    skipNumberOfBytes/2 + codePhase-1 (the same place postProcessing.m:74 starts the acquisition buffer,
    dataAdaptCoeff*skipNumberOfBytes bytes in).  Only GPS_L1CA, GAL_E5a, GAL_E5b and BDS/B3I have the int16 branch
    (`int16_branch`); GPS_L2C seeks without the -1 (GPS_L2C/include/tracking.m:153, `minus_one` False)."""
    cp = int(code_phase) - (1 if minus_one else 0)
    skip = int(getattr(settings, "skipNumberOfBytes", 0))
    if str(getattr(settings, "dataType", "schar")) == "int16":
        if not int16_branch:
            raise ValueError("this package's tracking.m has no int16 branch: its fseek assumes one byte per component")
        if skip % 2:
            raise ValueError("int16 record: dataAdaptCoeff*skipNumberOfBytes bytes is not a whole number of samples")
        return skip // 2 + cp
    return skip + cp


def tracking_l1ca(if_bytes: np.ndarray, channel, settings, correlate=None):
    """Closed-loop restatement of GPS/GPS_L1CA/include/tracking.m.

    `if_bytes` stands for the file behind `fid` (int8 interleaved I/Q for fileType 2).
    `channel` is a list of objects with PRN, acquiredFreq, codePhase, status (preRun.m:44-73).
    `correlate` lets a test substitute the device correlator for lines 247-300 while keeping
    the host loop closure; default is the oracle's own correlate_block.
    Returns a list of SimpleNamespace trackResults (fields tracking.m:47-86).
    """
    code_periods = int(settings.msToProcess)
    d = settings.dllCorrelatorSpacing
    pdi_code = settings.intTime
    tau1code, tau2code = calc_loop_coef(settings.dllNoiseBandwidth, settings.dllDampingRatio, 1.0)
    pdi_carr = settings.intTime
    tau1carr, tau2carr = calc_loop_coef(settings.pllNoiseBandwidth, settings.pllDampingRatio, 0.25)
    adapt = 1 if settings.fileType == 1 else 2
    n_total = if_bytes.shape[0] // adapt
    results = []
    for _ in channel:  # trackResults = repmat(trackResults, 1, numberOfChannels), tracking.m:47-86
        tr = SimpleNamespace(status="-", PRN=0)
        for f in TRACK_FIELDS:
            setattr(tr, f, np.zeros(code_periods))
        tr.CNo = SimpleNamespace(VSMValue=[], VSMIndex=[])
        results.append(tr)
    for tr, ch in zip(results, channel):
        if ch.PRN == 0:
            continue
        tr.PRN = ch.PRN
        pos = first_sample(settings, ch.codePhase)  # tracking.m:145-153
        table = pad_code(generate_ca_code(ch.PRN))  # :156-158
        code_freq = settings.codeFreqBasis
        rem_code = 0.0
        carr_freq = ch.acquiredFreq
        carr_basis = ch.acquiredFreq
        rem_carr = 0.0
        old_code_nco = 0.0
        old_code_err = 0.0
        old_carr_nco = 0.0
        old_carr_err = 0.0
        aborted = False
        for loop in range(code_periods):
            tr.absoluteSample[loop] = pos  # :212-216
            step = code_freq / settings.samplingFreq  # :219
            n = blksize_for(settings.codeLength, rem_code, step)  # :222
            if pos + n > n_total:  # short read, :241-245
                aborted = True
                break
            tr.remCodePhase[loop] = rem_code  # :249
            tr.remCarrPhase[loop] = rem_carr  # :277
            if correlate is None:
                raw = raw_from_if(if_bytes, pos, n, settings.fileType)
                sums, rem_code_new, rem_carr_new = correlate_block(
                    raw, [table], rem_code, step, d, carr_freq, rem_carr,
                    settings.samplingFreq, settings.codeLength)
            else:
                sums, rem_code_new, rem_carr_new = correlate(
                    ch, pos, n, rem_code, step, d, carr_freq, rem_carr)
            pos += n
            i_e, q_e, i_p, q_p, i_l, q_l = (float(v) for v in sums[0])
            rem_code, rem_carr = rem_code_new, rem_carr_new
            # PLL, :305-317
            with np.errstate(divide="ignore", invalid="ignore"):
                carr_err = float(np.arctan(np.float64(q_p) / np.float64(i_p)) / (2.0 * math.pi))
            carr_nco = old_carr_nco + (tau2carr / tau1carr) * (carr_err - old_carr_err) \
                + carr_err * (pdi_carr / tau1carr)
            old_carr_nco = carr_nco
            old_carr_err = carr_err
            tr.carrFreq[loop] = carr_freq
            carr_freq = carr_basis + carr_nco
            # DLL, :322-335
            e_mag = math.sqrt(i_e * i_e + q_e * q_e)
            l_mag = math.sqrt(i_l * i_l + q_l * q_l)
            with np.errstate(divide="ignore", invalid="ignore"):
                code_err = float((np.float64(e_mag) - l_mag) / (np.float64(e_mag) + l_mag))
            code_nco = old_code_nco + (tau2code / tau1code) * (code_err - old_code_err) \
                + code_err * (pdi_code / tau1code)
            old_code_nco = code_nco
            old_code_err = code_err
            tr.codeFreq[loop] = code_freq
            code_freq = settings.codeFreqBasis - code_nco
            # record, :338-348
            tr.dllDiscr[loop] = code_err
            tr.dllDiscrFilt[loop] = code_nco
            tr.pllDiscr[loop] = carr_err
            tr.pllDiscrFilt[loop] = carr_nco
            tr.I_E[loop], tr.I_P[loop], tr.I_L[loop] = i_e, i_p, i_l
            tr.Q_E[loop], tr.Q_P[loop], tr.Q_L[loop] = q_e, q_p, q_l
            # C/N0, :351-358
            vsm = int(settings.CNo.VSMinterval)
            if (loop + 1) % vsm == 0:
                val = cno_vsm(tr.I_P[loop + 1 - vsm:loop + 1], tr.Q_P[loop + 1 - vsm:loop + 1],
                              settings.CNo.accTime)
                tr.CNo.VSMValue.append(val)
                tr.CNo.VSMIndex.append(loop + 1)
        if aborted:
            # the reference prints and returns from the whole function (:241-245)
            return results
        tr.status = ch.status  # :365
    return results


# --------------------------------------------------------------------------------------
# acquisition(longSignal, settings) — GPS L1 C/A (acquisition.m:113-260), resampling off
# --------------------------------------------------------------------------------------


def matlab_var(x: np.ndarray) -> float:
    """var() of a (complex) vector: sum |x-mean|^2 / (N-1)."""
    m = np.mean(x)
    return float(np.sum(np.abs(x - m) ** 2) / (x.shape[0] - 1))


def acquisition_coarse_results(long_signal: np.ndarray, prn: int, settings, tables=None) -> np.ndarray:
    """results[bin, tau] of acquisition.m:158-192 for one PRN (float64 FFTs).  `tables`: explicit list of
    sampled codes; more than one = the data+pilot sum of GPS_L5C/include/acquisition.m:210-216."""
    spc = samples_per_code(settings)
    ts = 1.0 / settings.samplingFreq
    phase_points = np.arange(0, spc * 2, dtype=np.float64) * 2 * math.pi * ts  # :122
    n_bins = int(matlab_round(settings.acqSearchBand * 2 / settings.acqSearchStep)) + 1  # :124
    if tables is None:
        tables = [make_ca_table(prn, settings)]
    code_fds = [np.conj(np.fft.fft(np.concatenate([t, np.zeros(spc)]))) for t in tables]  # :160-164
    results = np.zeros((n_bins, spc * 2))
    for b in range(n_bins):
        f = settings.IF + settings.acqSearchBand - settings.acqSearchStep * b  # :169-170
        sig_carr = np.exp(-1j * f * phase_points)  # :172
        for h in range(int(settings.acqNonCohTime)):
            sig = long_signal[h * spc:(h + 2) * spc]  # :177-178
            iq = sig_carr * sig
            iq_fd = np.fft.fft(iq)  # :183
            for code_fd in code_fds:
                results[b, :] += np.abs(np.fft.ifft(iq_fd * code_fd))  # :186-190
    return results


def acquisition_l1ca(long_signal: np.ndarray, settings, want_results: bool = False):
    """acqResults = acquisition(longSignal, settings) (acquisition.m:113-260).

    With settings.resamplingflag == 1 above the resampling threshold the search runs on the conditioned signal
    (acquisition_front_end, :46-111) and the results are mapped back (:264-276).
    Returns SimpleNamespace(carrFreq, codePhase, peakMetric) with 32 entries each and, for
    test use, coarseBin (1-based, 0 = not computed).
    """
    original = settings
    long_signal, settings = acquisition_front_end(long_signal, settings)
    resampled = settings is not original
    spc = samples_per_code(settings)
    ts = 1.0 / settings.samplingFreq
    n_bins = int(matlab_round(settings.acqSearchBand * 2 / settings.acqSearchStep)) + 1
    acq = SimpleNamespace(carrFreq=np.zeros(32), codePhase=np.zeros(32), peakMetric=np.zeros(32),
                          coarseBin=np.zeros(32, dtype=np.int64))
    fine_step = 25
    n_fine = int(matlab_round(settings.acqSearchStep / fine_step)) + 1  # :140
    fine_phase = np.arange(0, 40 * spc, dtype=np.float64) * 2 * math.pi * ts  # :148
    sig_power = math.sqrt(matlab_var(long_signal[:spc]) * spc)  # :151
    all_results = {}
    for prn in settings.acqSatelliteList:
        results = acquisition_coarse_results(long_signal, prn, settings)
        if want_results:
            all_results[prn] = results
        # :196-198 — max() returns the first maximum
        coarse_bin = int(np.argmax(np.max(results, axis=1)))
        col_max = np.max(results, axis=0)
        code_phase0 = int(np.argmax(col_max))
        peak = float(col_max[code_phase0])
        acq.peakMetric[prn - 1] = peak / sig_power / settings.acqNonCohTime  # :200
        acq.coarseBin[prn - 1] = coarse_bin + 1
        if acq.peakMetric[prn - 1] > settings.acqThreshold:  # :206
            code_phase = code_phase0 + 1  # 1-based sample index
            ca = generate_ca_code(prn)
            tc = 1.0 / settings.codeFreqBasis
            idx = np.floor((ts * np.arange(0, 40 * spc, dtype=np.float64)) / tc).astype(np.int64)  # :215
            ca40 = ca[idx % int(settings.codeLength)]  # :218
            sig40 = long_signal[code_phase - 1:code_phase - 1 + 40 * spc]  # :221
            fine_result = np.zeros(n_fine)
            fine_bins = np.zeros(n_fine)
            coarse_f = settings.IF + settings.acqSearchBand - settings.acqSearchStep * coarse_bin
            for fb in range(n_fine):
                fine_bins[fb] = coarse_f + settings.acqSearchStep / 2 - fine_step * fb  # :227-228
                carr = np.exp(-1j * fine_bins[fb] * fine_phase)  # :230
                bb = sig40 * ca40 * carr  # :232
                sum_per_code = bb.reshape(40, spc).sum(axis=1)  # :235-238
                max_power = 0.0
                for com in range(20):  # :243-248
                    max_power = max(max_power, abs(np.sum(sum_per_code[com:com + 20])))
                fine_result[fb] = max_power
            max_fin = int(np.argmax(fine_result))  # :253
            acq.carrFreq[prn - 1] = fine_bins[max_fin]
            acq.codePhase[prn - 1] = code_phase
            if acq.carrFreq[prn - 1] == 0:  # :258-260
                acq.carrFreq[prn - 1] = 1
            if resampled:  # :264-276
                acq.codePhase[prn - 1] = math.floor((code_phase - 1) / settings.samplingFreq * settings.oldFreq) + 1
                if settings.IF >= settings.samplingFreq / 2:
                    doppler = (settings.samplingFreq - settings.IF) - acq.carrFreq[prn - 1]
                else:
                    doppler = acq.carrFreq[prn - 1] - settings.IF
                acq.carrFreq[prn - 1] = doppler + settings.oldIF
    if want_results:
        return acq, all_results
    return acq


def pre_run(acq, settings):
    """channel = preRun(acqResults, settings) (preRun.m:44-73)."""
    n_ch = int(settings.numberOfChannels)
    channel = [SimpleNamespace(PRN=0, acquiredFreq=0.0, codePhase=0, status="-") for _ in range(n_ch)]
    # sort(..., 'descend') is stable: ties keep ascending index order
    order = np.argsort(-acq.peakMetric, kind="stable")
    n_found = int(np.sum(acq.carrFreq != 0))
    for ii in range(min(n_ch, n_found)):
        p = int(order[ii])
        channel[ii].PRN = p + 1
        channel[ii].acquiredFreq = float(acq.carrFreq[p])
        channel[ii].codePhase = int(acq.codePhase[p])
        channel[ii].status = "T"
    return channel


# --------------------------------------------------------------------------------------
# Galileo E1-B / E1-C codes (GAL/GAL_E1C/include/generateE1Bcode.m:40-67, generateE1Ccode.m)
# --------------------------------------------------------------------------------------
_E1_DATA = None


def _e1_bits(which: str, prn: int) -> np.ndarray:
    """Memory codes (no generator exists): packed copy of the reference's E1b.dat / E1c.dat,
    tests/golden/gal_e1_memory_codes.npz (made by tests/golden/make_e1_codes.py)."""
    global _E1_DATA
    if _E1_DATA is None:
        import os
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        _E1_DATA = np.load(os.path.join(here, "tests", "golden", "gal_e1_memory_codes.npz"))
    return np.unpackbits(_E1_DATA[which][prn - 1])[:4092].astype(np.float64)


def generate_e1_code(prn: int, component: str) -> np.ndarray:
    """8184 half-chips: primary code 1-2*bit (logic 1 -> -1, generateE1Bcode.m:56) times the
    BOC(1,1) sub-carrier [+1, -1] (:59-65).  component 'B' (data) or 'C' (pilot)."""
    raw = 1.0 - 2.0 * _e1_bits("E1b" if component == "B" else "E1c", prn)
    out = np.empty(2 * raw.shape[0])
    out[0::2] = raw
    out[1::2] = -raw
    return out


# --------------------------------------------------------------------------------------
# Generic closed loop: 3-state PLL, data + pilot arms, R-scaled tables
# (GAL/GAL_E1C/include/tracking.m:45-383; GPS/GPS_L5C/include/tracking.m:47-424)
# --------------------------------------------------------------------------------------


def tracking_generic(if_bytes: np.ndarray, channel, settings, spec, correlate=None):
    """spec: SimpleNamespace(
         tables(prn) -> list of padded tables (arm 0 = data, arm 1 = pilot),
         r            index scale (2 for the BOC(1,1) half-chip tables, GAL_E1C tracking.m:236),
         pll          '2nd' (L1CA) | '3state' (calcLoopCoefCarr variant spec.coef_variant),
         pilot_combine 0 none | 1 rotate pilot by exp(-1i*pi/2) then average (GPS_L5C tracking.m:336-348)
                       | 2 plain average (GAL_E1C tracking.m:303-311,326-331)
                       | 3 pilot in quadrature atan(-I/Q) (B1C NB_tracking.m:341) | 4 B1C wide-band three-arm fold,
         optional pll_weight / dll_weight (data, pilot), dll_scale_spacing, arm_mult (per-arm ramp multipliers),
         code_freq_from_channel  True: channel.codeFreq (GPS_L5C tracking.m:165), False: codeFreqBasis)
    Epoch count NumToProcess = round(msToProcess/1000/intTime) (GAL_E1C tracking.m:51)."""
    n_ep = int(matlab_round(settings.msToProcess / 1000 / settings.intTime))
    d = settings.dllCorrelatorSpacing
    pdi = settings.intTime
    tau1code, tau2code = calc_loop_coef(settings.dllNoiseBandwidth, settings.dllDampingRatio, 1.0)
    if spec.pll == "2nd":
        tau1carr, tau2carr = calc_loop_coef(settings.pllNoiseBandwidth, settings.pllDampingRatio, 0.25)
    else:
        pf3, pf2, pf1 = calc_loop_coef_carr(settings, spec.coef_variant)
    n_total = if_bytes.shape[0] // 2
    fields = TRACK_FIELDS + ("Pilot_I_E", "Pilot_Q_E", "Pilot_I_P", "Pilot_Q_P", "Pilot_I_L", "Pilot_Q_L")
    results = []
    for _ in channel:
        tr = SimpleNamespace(status="-", PRN=0)
        for f in fields:
            setattr(tr, f, np.zeros(n_ep))
        results.append(tr)
    for tr, ch in zip(results, channel):
        if ch.PRN == 0:
            continue
        tr.PRN = ch.PRN
        pos = first_sample(settings, ch.codePhase, int16_branch=getattr(spec, "int16_branch", False))
        tables = spec.tables(ch.PRN)  # GLONASS: one code for every channel, PRN carries K (GLO tracking.m:88-89,136)
        basis = ch.codeFreq if spec.code_freq_from_channel else settings.codeFreqBasis
        code_freq = basis
        rem_code = 0.0
        carr_freq = carr_basis = ch.acquiredFreq
        rem_carr = 0.0
        old_code_nco = old_code_err = 0.0
        old_carr_nco = old_carr_err = 0.0
        d2_carr = d_carr = 0.0
        for e in range(n_ep):
            tr.absoluteSample[e] = pos
            step = code_freq / settings.samplingFreq
            n = blksize_for(settings.codeLength, rem_code, step)
            if pos + n > n_total:
                return results
            tr.remCodePhase[e] = rem_code
            tr.remCarrPhase[e] = rem_carr
            if correlate is None:
                sums, rem_code_new, rem_carr_new = correlate_block(
                    raw_from_if(if_bytes, pos, n, swap_iq=getattr(spec, "swap_iq", False)), tables, rem_code, step, d,
                    carr_freq, rem_carr, settings.samplingFreq, settings.codeLength, r=spec.r,
                    arm_mult=getattr(spec, "arm_mult", None))
            else:
                sums, rem_code_new, rem_carr_new = correlate(ch, pos, n, rem_code, step, d, carr_freq, rem_carr)
            pos += n
            rem_code, rem_carr = rem_code_new, rem_carr_new
            i_e, q_e, i_p, q_p, i_l, q_l = (float(v) for v in sums[0])
            with np.errstate(divide="ignore", invalid="ignore"):
                carr_err = float(np.arctan(np.float64(q_p) / np.float64(i_p)) / (2.0 * math.pi))
            code_err = float((math.sqrt(i_e * i_e + q_e * q_e) - math.sqrt(i_l * i_l + q_l * q_l)) /
                             (math.sqrt(i_e * i_e + q_e * q_e) + math.sqrt(i_l * i_l + q_l * q_l)))
            if spec.pilot_combine:
                if spec.pilot_combine == 4:
                    # BDS/B1C/include/WB_tracking.m:364-369: arms {data, pilot BOC(1,1), pilot BOC(6,1)}
                    a61, a11 = -math.sqrt(4 / 33), math.sqrt(29 / 33)
                    p11, p61 = [float(v) for v in sums[1]], [float(v) for v in sums[2]]
                    pil = []
                    for x in range(3):
                        pil += [a61 * p61[2 * x] + a11 * p11[2 * x + 1], a61 * p61[2 * x + 1] - a11 * p11[2 * x]]
                    pi_e, pq_e, pi_p, pq_p, pi_l, pq_l = pil
                elif spec.pilot_combine == 5:
                    # Galileo E1-C CBOC(6,1,1/11) pilot (OS SIS ICD 2.3.3): sqrt(10/11) BOC(1,1) - sqrt(1/11) BOC(6,1), in phase;
                    # not in the reference (GAL_E1C tracks with BOC(1,1) only): BASELINE config 3, parity against this oracle
                    a11, a61 = math.sqrt(10 / 11), -math.sqrt(1 / 11)
                    p11, p61 = [float(v) for v in sums[1]], [float(v) for v in sums[2]]
                    pi_e, pq_e, pi_p, pq_p, pi_l, pq_l = [a11 * p11[v] + a61 * p61[v] for v in range(6)]
                else:
                    pi_e, pq_e, pi_p, pq_p, pi_l, pq_l = (float(v) for v in sums[1])
                with np.errstate(divide="ignore", invalid="ignore"):
                    if spec.pilot_combine == 1:
                        qi = (pi_p + 1j * pq_p) * np.exp(-1j * math.pi / 2)
                        carr_err_q = float(np.arctan(np.float64(qi.imag) / np.float64(qi.real)) / (2.0 * math.pi))
                    elif spec.pilot_combine == 3:
                        carr_err_q = float(np.arctan(np.float64(-pi_p) / np.float64(pq_p)) / (2.0 * math.pi))   # NB_tracking.m:341
                    else:
                        carr_err_q = float(np.arctan(np.float64(pq_p) / np.float64(pi_p)) / (2.0 * math.pi))
                code_err_q = (math.sqrt(pi_e ** 2 + pq_e ** 2) - math.sqrt(pi_l ** 2 + pq_l ** 2)) / \
                             (math.sqrt(pi_e ** 2 + pq_e ** 2) + math.sqrt(pi_l ** 2 + pq_l ** 2))
                if getattr(spec, "dll_scale_spacing", False):      # * (1-earlyLateSpc), NB_tracking.m:346-348
                    code_err = code_err * (1 - d)
                    code_err_q = code_err_q * (1 - d)
                pw = getattr(spec, "pll_weight", None)
                dw = getattr(spec, "dll_weight", None)
                if pw:    # (carrError*11 + p11_carrError*29)/40 (NB :342); (carrError*1 + p_carrError*3)/4 (WB :382)
                    carr_err = (carr_err * pw[0] + carr_err_q * pw[1]) / (pw[0] + pw[1])
                else:
                    carr_err = (carr_err + carr_err_q) / 2
                if dw and spec.pilot_combine == 4:    # codeError*factor + p_codeError*(1-factor), WB_tracking.m:403
                    code_err = code_err * dw[0] + code_err_q * dw[1]
                elif dw:                               # (codeError*11 + p11_codeError*29)/40, NB_tracking.m:349
                    code_err = (code_err * dw[0] + code_err_q * dw[1]) / (dw[0] + dw[1])
                else:
                    code_err = (code_err + code_err_q) / 2
                tr.Pilot_I_E[e], tr.Pilot_Q_E[e], tr.Pilot_I_P[e] = pi_e, pq_e, pi_p
                tr.Pilot_Q_P[e], tr.Pilot_I_L[e], tr.Pilot_Q_L[e] = pq_p, pi_l, pq_l
            if spec.pll == "2nd":
                carr_nco = old_carr_nco + (tau2carr / tau1carr) * (carr_err - old_carr_err) + carr_err * (pdi / tau1carr)
                old_carr_nco, old_carr_err = carr_nco, carr_err
            else:
                d2_carr = d2_carr + carr_err * pf3
                d_carr = d2_carr + carr_err * pf2 + d_carr
                carr_nco = d_carr + carr_err * pf1
            tr.carrFreq[e] = carr_freq
            carr_freq = carr_basis + carr_nco
            code_nco = old_code_nco + (tau2code / tau1code) * (code_err - old_code_err) + code_err * (pdi / tau1code)
            old_code_nco, old_code_err = code_nco, code_err
            tr.codeFreq[e] = code_freq
            code_freq = basis - code_nco
            tr.dllDiscr[e], tr.dllDiscrFilt[e] = code_err, code_nco
            tr.pllDiscr[e], tr.pllDiscrFilt[e] = carr_err, carr_nco
            tr.I_E[e], tr.I_P[e], tr.I_L[e] = i_e, i_p, i_l
            tr.Q_E[e], tr.Q_P[e], tr.Q_L[e] = q_e, q_p, q_l
        tr.status = ch.status
    return results


# --------------------------------------------------------------------------------------
# GPS L5 I5 / Q5 codes (GPS/GPS_L5C/include/generateL5Icode.m:44-133, generateL5Qcode.m)
# --------------------------------------------------------------------------------------
# XB code advances, IS-GPS-705 Table 3-I, PRN 1..37 (the reference carries more SBAS/extended PRNs)
_L5I_ADV = [266, 365, 804, 1138, 1509, 1559, 1756, 2084, 2170, 2303, 2527, 2687, 2930, 3471, 3940, 4132,
            4332, 4924, 5343, 5443, 5641, 5816, 5898, 5918, 5955, 6243, 6345, 6477, 6518, 6875, 7168, 7187,
            7329, 7577, 7720, 7777, 8057]
_L5Q_ADV = [1701, 323, 5292, 2020, 5429, 7136, 1041, 5947, 4315, 148, 535, 1939, 5206, 5910, 3595, 5135,
            6082, 6990, 3546, 1523, 4548, 4484, 1893, 3961, 7106, 5299, 4660, 276, 4389, 3783, 1591, 1601,
            749, 1387, 1661, 3210, 708]


def generate_l5_code(prn: int, component: str, code_length: int = 10230) -> np.ndarray:
    """Product-form LFSRs exactly as the reference: XA taps {9,10,12,13}, short-cycled when the register
    equals [-1 x11, +1, -1] (:54-67); XB taps {1,3,4,6,7,8,12,13} pre-advanced by the PRN's table entry
    (:103-121); code = XB .* XA (:123)."""
    xa_reg = -np.ones(13)
    reset_state = np.array([-1.0] * 11 + [1.0, -1.0])
    xa = np.empty(code_length)
    for i in range(code_length):
        xa[i] = xa_reg[-1]
        if np.array_equal(xa_reg, reset_state):
            xa_reg = -np.ones(13)
        else:
            fb = xa_reg[8] * xa_reg[9] * xa_reg[11] * xa_reg[12]
            xa_reg = np.concatenate([[fb], xa_reg[:-1]])
    adv = (_L5I_ADV if component == "I" else _L5Q_ADV)[prn - 1]
    xb_reg = -np.ones(13)
    taps = [0, 2, 3, 5, 6, 7, 11, 12]
    for _ in range(adv):
        fb = np.prod(xb_reg[taps])
        xb_reg = np.concatenate([[fb], xb_reg[:-1]])
    xb = np.empty(code_length)
    for i in range(code_length):
        xb[i] = xb_reg[-1]
        fb = np.prod(xb_reg[taps])
        xb_reg = np.concatenate([[fb], xb_reg[:-1]])
    return xb * xa


# --------------------------------------------------------------------------------------
# GLONASS L1OF ranging code (GLO/GLO_GL1/include/generateCAcode.m:93-104) and BDS B1I
# (BDS/B1I/include/generateCAcode53.m:42-103)
# --------------------------------------------------------------------------------------


def generate_glo_code() -> np.ndarray:
    """511-chip m-sequence: 9-stage register of -1s, feedback reg(5)*reg(9), output stage 7."""
    reg = -np.ones(9)
    code = np.empty(511)
    for i in range(511):
        code[i] = reg[6]
        save1 = reg[4] * reg[8]
        reg[1:9] = reg[0:8].copy()
        reg[0] = save1
    return code


# G2 phase-selector taps for PRN 1..58 (BDS-SIS-ICD-B1I-3.0 Table 4-1; generateCAcode53.m:58-76): two stages up to PRN 37, a
# third one from PRN 38 on (:78-86)
_B1I_S1 = [1, 1, 1, 1, 1, 1, 1, 1, 2, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 6, 6, 6, 6, 8, 8, 8, 9, 9, 10] + \
          [1] * 16 + [2] * 3 + [3] * 2
_B1I_S2 = [3, 4, 5, 6, 8, 9, 10, 11, 7, 4, 5, 6, 8, 9, 10, 11, 5, 6, 8, 9, 10, 11, 6, 8, 9, 10, 11, 8, 9, 10, 11, 9, 10, 11, 10, 11, 11] + \
          [2] + [3] * 5 + [4] * 2 + [5] * 4 + [6, 8, 9, 9, 3, 5, 7, 4, 4]
_B1I_S3 = [7, 4, 6, 8, 10, 11, 5, 9, 6, 8, 10, 11, 9, 9, 10, 11, 7, 7, 9, 5, 9]


def generate_b1i_code(prn: int) -> np.ndarray:
    """2046 chips; G1 taps {1,7,8,9,10,11}, G2 taps {1,2,3,4,5,8,9,11}, both registers start at
    -1*[-1 1 -1 1 ...] (:43,:55), G2 output = product of the PRN's two selector stages (:94),
    CAcode = -(g1.*g2) (:103).  PRN 1..58."""
    init = -1.0 * np.array([-1, 1, -1, 1, -1, 1, -1, 1, -1, 1, -1], dtype=np.float64)
    reg = init.copy()
    g1 = np.empty(2046)
    for i in range(2046):
        g1[i] = reg[10]
        save = reg[0] * reg[6] * reg[7] * reg[8] * reg[9] * reg[10]
        reg[1:11] = reg[0:10].copy()
        reg[0] = save
    reg = init.copy()
    g2 = np.empty(2046)
    s1, s2 = _B1I_S1[prn - 1] - 1, _B1I_S2[prn - 1] - 1
    s3 = _B1I_S3[prn - 38] - 1 if prn > 37 else None
    for i in range(2046):
        g2[i] = reg[s1] * reg[s2] * (reg[s3] if s3 is not None else 1.0)
        save = reg[0] * reg[1] * reg[2] * reg[3] * reg[4] * reg[7] * reg[8] * reg[10]
        reg[1:11] = reg[0:10].copy()
        reg[0] = save
    return -(g1 * g2)


# --------------------------------------------------------------------------------------
# Code generators of the 10.23-Mcps family, L2C and B1C.  Per-chip restatements in the reference's own
# +-1 ("multiply = XOR") form; the ICD constant tables come from the data file the product ships
# (cu-sdr-collection_amd/data/icd_tables.npz, made by tests/golden/make_icd_tables.py).
# --------------------------------------------------------------------------------------
_ICD_TABLES = None


def _icd_table(name: str) -> np.ndarray:
    global _ICD_TABLES
    if _ICD_TABLES is None:
        here = os.path.dirname(os.path.abspath(__file__))
        _ICD_TABLES = dict(np.load(os.path.join(here, "..", "cu-sdr-collection_amd", "data", "icd_tables.npz")))
    return _ICD_TABLES[name]


def _shift_in(reg: np.ndarray, fb: float) -> np.ndarray:
    """circshift(reg',1)' ; reg(1) = fb"""
    return np.concatenate([[fb], reg[:-1]])


def generate_b2a_code(prn: int, component: str, code_length: int = 10230) -> np.ndarray:
    """BDS/B2a/include/generateB2aDataCode.m:109-138 (data: register-1 taps [1 5 11 13], register-2 taps
    [3 5 9 11 12 13]) and generateB2aPilotCode.m:104-138 (pilot: [3 6 7 13], [1 5 7 8 12 13]).  Register 1 starts
    at -1 x13 and is set back to that after chip 8190 (:124,136-138); register 2 starts at 1 - 2*ini(PRN,:)."""
    if component == "data":
        t1, t2, ini = [0, 4, 10, 12], [2, 4, 8, 10, 11, 12], _icd_table("b2a_data_g2")
    else:
        t1, t2, ini = [2, 5, 6, 12], [0, 4, 6, 7, 11, 12], _icd_table("b2a_pilot_g2")
    r1 = -np.ones(13)
    r2 = 1.0 - 2.0 * ini[prn - 1].astype(np.float64)
    code = np.empty(code_length)
    for i in range(code_length):
        code[i] = r1[-1] * r2[-1]
        r1 = _shift_in(r1, np.prod(r1[t1]))
        r2 = _shift_in(r2, np.prod(r2[t2]))
        if i + 1 == 8190:
            r1 = -np.ones(13)
    return code


def generate_b3i_code(prn: int) -> np.ndarray:
    """BDS/B3I/include/generateB3Icode.m:39-110.  CA: taps [1 3 4 13], output = last stage, and when the register
    equals [-1 x11, 1, 1] it is set to -1 x13 INSTEAD of shifting (:52-62).  CB: taps [1 5 6 7 9 10 12 13], shifted
    B3I_init(PRN) times before the first output (:77-86).  Code = CB .* CA."""
    reset_state = np.array([-1.0] * 11 + [1.0, 1.0])
    reg = -np.ones(13)
    ca = np.empty(10230)
    for i in range(10230):
        ca[i] = reg[-1]
        if np.array_equal(reg, reset_state):
            reg = -np.ones(13)
        else:
            reg = _shift_in(reg, reg[0] * reg[2] * reg[3] * reg[12])
    taps = [0, 4, 5, 6, 8, 9, 11, 12]
    reg = -np.ones(13)
    for _ in range(int(_icd_table("b3i_advance")[prn - 1])):
        reg = _shift_in(reg, np.prod(reg[taps]))
    cb = np.empty(10230)
    for i in range(10230):
        cb[i] = reg[-1]
        reg = _shift_in(reg, np.prod(reg[taps]))
    return cb * ca


def _dec2bin(v: int) -> np.ndarray:
    return np.array([int(c) for c in bin(int(v))[2:]], dtype=np.int64)


def generate_e5_primary(sig: str, prn: int) -> np.ndarray:
    """GAL/GAL_E5a/include/generateE5aIcode.m:36-103 (same structure in generateE5aQcode.m and the E5b twins).
    sig in {"e5ai", "e5aq", "e5bi", "e5bq"}.  Registers are 0/1 row vectors; taps = first 14 binary digits of the
    octal polynomial (:61-64); Register1 = ones, Register2 = start value right-aligned (:67-71); per chip: output
    (1-2*R1(1)*t1(1)) * (1-2*R2(1)*t2(1)), feedback = XOR of the tapped elements, shift LEFT, feedback into the
    last element (:76-103)."""
    polys = _icd_table(sig + "_poly_octal")
    t1 = _dec2bin(polys[0])[:14]
    t2 = _dec2bin(polys[1])[:14]
    r1 = np.ones(14, dtype=np.int64)
    sv = _dec2bin(_icd_table(sig + "_start_octal")[prn - 1])
    r2 = np.zeros(14, dtype=np.int64)
    r2[14 - sv.size:] = sv
    code = np.empty(10230)
    for i in range(10230):
        o1, o2 = r1 * t1, r2 * t2
        code[i] = (1 - 2 * o1[0]) * (1 - 2 * o2[0])
        f1 = 0 if np.prod(1 - 2 * o1) == 1 else 1
        f2 = 0 if np.prod(1 - 2 * o2) == 1 else 1
        r1 = np.concatenate([r1[1:], [f1]])
        r2 = np.concatenate([r2[1:], [f2]])
    return code


def generate_e5_secondary100(sig: str, prn: int) -> np.ndarray:
    """generateE5aQ_secondary.m:39-87: 25 hex digits; the first 13 give 52 bits, the last 12 give 48 bits, each
    right-aligned in its segment; 1 - 2*bit."""
    h = str(_icd_table(sig + "_secondary_hex")[prn - 1])
    first = np.zeros(52, dtype=np.int64)
    b = _dec2bin(int(h[:13], 16))
    first[52 - b.size:] = b
    second = np.zeros(48, dtype=np.int64)
    b = _dec2bin(int(h[13:], 16))
    second[48 - b.size:] = b
    return 1.0 - 2.0 * np.concatenate([first, second])


def generate_e5_code(sig: str, prn: int, flag: int) -> np.ndarray:
    """flag 1: primary.  flag 2: tiered — E5a-I with [-1 1 1 1 1 -1 1 1 1 1 -1 1 -1 -1 -1 1 -1 1 1 -1] (842E9,
    generateE5aIcode.m:110-121), E5b-I with [-1 -1 -1 1] (E), the Q components with their CS100."""
    prim = generate_e5_primary(sig, prn)
    if flag == 1:
        return prim
    if sig == "e5ai":
        sec = np.array([-1, 1, 1, 1, 1, -1, 1, 1, 1, 1, -1, 1, -1, -1, -1, 1, -1, 1, 1, -1], dtype=np.float64)
    elif sig == "e5bi":
        sec = np.array([-1, -1, -1, 1], dtype=np.float64)
    else:
        sec = generate_e5_secondary100(sig, prn)
    return np.concatenate([prim * s for s in sec])


def _l2c_register(octal_as_int: int) -> np.ndarray:
    """dec2bin(oct2dec(init)) left-padded to 27, then 1 -> -1, 0 -> +1 (generateCMcode.m:97-101)."""
    b = _dec2bin(octal_as_int)
    reg = np.concatenate([np.zeros(27 - b.size, dtype=np.int64), b]).astype(np.float64)
    return np.where(reg == 1, -1.0, 1.0)


def generate_l2c_code(prn: int, which: str, n_chips: int) -> np.ndarray:
    """GPS/GPS_L2C/include/generateCMcode.m:39-111 / generateCLcode.m: out = reg(end); reg = circshift(reg,1);
    reg([4 7 9 12 15 17 19 22 23 24 25]) *= out.  Returned interleaved with zeros: CM = [chip 0 chip 0 ...],
    CL = [0 chip 0 chip ...] (:110-111).  PRN 1..63 -> entry PRN, 159..210 -> entry PRN - 95 (:84-91)."""
    idx = prn - 1 if 1 <= prn <= 63 else prn - 96
    reg = _l2c_register(int(_icd_table("l2cm_init_octal" if which == "CM" else "l2cl_init_octal")[idx]))
    pos = np.array([4, 7, 9, 12, 15, 17, 19, 22, 23, 24, 25]) - 1
    chips = np.empty(n_chips)
    for i in range(n_chips):
        chips[i] = reg[-1]
        reg = np.roll(reg, 1)
        reg[pos] *= chips[i]
    out = np.zeros(2 * n_chips)
    out[(0 if which == "CM" else 1)::2] = chips
    return out


def jacobi_symbol(a: int, n: int) -> int:
    """Jacobi symbol (a/n), n odd positive — the standard binary algorithm (BDS/B1C/include/JacobiSymbol.m computes
    the same function through factorisation and reciprocity)."""
    a %= n
    result = 1
    while a:
        while a % 2 == 0:
            a //= 2
            if n % 8 in (3, 5):
                result = -result
        a, n = n, a
        if a % 4 == 3 and n % 4 == 3:
            result = -result
        a %= n
    return result if n == 1 else 0


def generate_weil(n_mod: int, w: int, p: int, n: int) -> np.ndarray:
    """generateDataBOC11.m:60-78: legendre(1) = 0, legendre(i+1) = Jacobi(i, N) with -1 -> 0;
    chip(ind) = xor(legendre(k+1), legendre(mod(k+w, N)+1)), k = mod(ind + p - 1, N); 1 - 2*chip."""
    leg = np.zeros(n_mod, dtype=np.int64)
    for i in range(1, n_mod):
        leg[i] = 1 if jacobi_symbol(i, n_mod) == 1 else 0
    out = np.empty(n)
    for ind in range(n):
        k = (ind + p - 1) % n_mod
        out[ind] = 1.0 - 2.0 * (leg[k] ^ leg[(k + w) % n_mod])
    return out


def generate_b1c_code(prn: int, which: str) -> np.ndarray:
    """which = "data" (generateDataBOC11.m:80-91, chip -> [-chip, chip]), "pilot11" (generatePilotBOC11.m, same
    sub-carrier), "pilot61" (generatePilotBOC61.m:89-96, chip -> (-1)^ii * chip, ii = 1..12), "secondary"
    (generate2ndCode.m: N = 3607, 1800 chips)."""
    if which == "secondary":
        w, p = _icd_table("b1c_secondary_wp")[prn - 1]
        return generate_weil(3607, int(w), int(p), 1800)
    w, p = _icd_table("b1c_data_wp" if which == "data" else "b1c_pilot_wp")[prn - 1]
    prim = generate_weil(10243, int(w), int(p), 10230)
    if which == "pilot61":
        sub = np.array([(-1.0) ** ii for ii in range(1, 13)])
    else:
        sub = np.array([-1.0, 1.0])
    return (prim[:, None] * sub[None, :]).reshape(-1)


def tracking_l2c(if_bytes: np.ndarray, channel, settings):
    """GPS/GPS_L2C/include/tracking.m:40-420 restated: the loop runs on the RZ-doubled code (earlyLateSpc*2,
    codeLength*2, codeFreqBasis*2, :107-109,171), seeks to skipNumberOfBytes + codePhase without -1 (:153), reads the
    CL arm at index + codeLength*(CLCodePhase-1) with CLCodePhase cycling 1..75 (:261,357-360), averages data and
    pilot discriminators when pilotTRKflag (:336-339,352-356), records remCodePhase, codeFreq, dllDiscr and
    dllDiscrFilt halved and absoluteSample = position + 1 - remCodePhase/codePhaseStep (:226,250,376,382-383)."""
    n_ep = int(matlab_round(settings.msToProcess / 1000 / settings.intTime))
    d = settings.dllCorrelatorSpacing * 2
    code_length = settings.codeLength * 2
    pdi = settings.intTime
    pilot = bool(getattr(settings, "pilotTRKflag", 0))
    tau1code, tau2code = calc_loop_coef(settings.dllNoiseBandwidth, settings.dllDampingRatio, 1.0)
    pf3, pf2, pf1 = calc_loop_coef_carr(settings, "a")
    n_total = if_bytes.shape[0] // 2
    fields = TRACK_FIELDS + ("Pilot_I_E", "Pilot_Q_E", "Pilot_I_P", "Pilot_Q_P", "Pilot_I_L", "Pilot_Q_L")
    results = []
    for _ in channel:
        tr = SimpleNamespace(status="-", PRN=0)
        for f in fields:
            setattr(tr, f, np.zeros(n_ep))
        results.append(tr)
    for tr, ch in zip(results, channel):
        if ch.PRN == 0:
            continue
        tr.PRN = ch.PRN
        pos = int(settings.skipNumberOfBytes + ch.codePhase)
        tables = [pad_code(generate_l2c_code(ch.PRN, "CM", int(settings.codeLength)))]
        cl_phase = 0
        if pilot:
            cl_phase = int(ch.CLCodePhase)
            tables.append(pad_code(generate_l2c_code(ch.PRN, "CL", int(settings.CLCodeLength))))
        code_freq = settings.codeFreqBasis * 2
        rem_code = 0.0
        carr_freq = carr_basis = ch.acquiredFreq
        rem_carr = 0.0
        old_code_nco = old_code_err = 0.0
        d2_carr = d_carr = 0.0
        for e in range(n_ep):
            step = code_freq / settings.samplingFreq
            tr.absoluteSample[e] = pos + 1 - rem_code / step
            n = blksize_for(code_length, rem_code, step)
            if pos + n > n_total:
                return results
            tr.remCodePhase[e] = rem_code / 2
            tr.remCarrPhase[e] = rem_carr
            off = [0, int(code_length) * (cl_phase - 1)] if pilot else None
            sums, rem_code, rem_carr = correlate_block(raw_from_if(if_bytes, pos, n), tables, rem_code, step, d, carr_freq,
                                                       rem_carr, settings.samplingFreq, code_length, table_offset=off)
            pos += n
            i_e, q_e, i_p, q_p, i_l, q_l = (float(v) for v in sums[0])
            with np.errstate(divide="ignore", invalid="ignore"):
                carr_err = float(np.arctan(np.float64(q_p) / np.float64(i_p)) / (2.0 * math.pi))
            code_err = (math.sqrt(i_e ** 2 + q_e ** 2) - math.sqrt(i_l ** 2 + q_l ** 2)) / \
                       (math.sqrt(i_e ** 2 + q_e ** 2) + math.sqrt(i_l ** 2 + q_l ** 2))
            if pilot:
                pi_e, pq_e, pi_p, pq_p, pi_l, pq_l = (float(v) for v in sums[1])
                with np.errstate(divide="ignore", invalid="ignore"):
                    carr_err = (carr_err + float(np.arctan(np.float64(pq_p) / np.float64(pi_p)) / (2.0 * math.pi))) / 2
                code_err_cl = (math.sqrt(pi_e ** 2 + pq_e ** 2) - math.sqrt(pi_l ** 2 + pq_l ** 2)) / \
                              (math.sqrt(pi_e ** 2 + pq_e ** 2) + math.sqrt(pi_l ** 2 + pq_l ** 2))
                code_err = (code_err + code_err_cl) / 2
                cl_phase += 1
                if cl_phase >= 76:
                    cl_phase = 1
                tr.Pilot_I_E[e], tr.Pilot_Q_E[e], tr.Pilot_I_P[e] = pi_e, pq_e, pi_p
                tr.Pilot_Q_P[e], tr.Pilot_I_L[e], tr.Pilot_Q_L[e] = pq_p, pi_l, pq_l
            d2_carr = d2_carr + carr_err * pf3
            d_carr = d2_carr + carr_err * pf2 + d_carr
            carr_nco = d_carr + carr_err * pf1
            tr.carrFreq[e] = carr_freq
            carr_freq = carr_basis + carr_nco
            code_nco = old_code_nco + (tau2code / tau1code) * (code_err - old_code_err) + code_err * (pdi / tau1code)
            old_code_nco, old_code_err = code_nco, code_err
            tr.codeFreq[e] = code_freq / 2
            code_freq = settings.codeFreqBasis * 2 - code_nco
            tr.dllDiscr[e], tr.dllDiscrFilt[e] = code_err / 2, code_nco / 2
            tr.pllDiscr[e], tr.pllDiscrFilt[e] = carr_err, carr_nco
            tr.I_E[e], tr.I_P[e], tr.I_L[e] = i_e, i_p, i_l
            tr.Q_E[e], tr.Q_P[e], tr.Q_L[e] = q_e, q_p, q_l
        tr.status = ch.status
    return results


# --------------------------------------------------------------------------------------
# Acquisition, circshift family — float64 restatements
# --------------------------------------------------------------------------------------
def _if_complex(if_bytes: np.ndarray, first: int, n: int) -> np.ndarray:
    seg = if_bytes[2 * first:2 * (first + n)].astype(np.float64)
    return seg[0::2] + 1j * seg[1::2]


def _second_peak_ratio(corr: np.ndarray, code_phase: int, exclude: int, period: int) -> float:
    e1, e2 = code_phase - exclude, code_phase + exclude
    if e1 < 2:
        rng = np.arange(e2, period + e1 + 1)
    elif e2 >= period:
        rng = np.arange(e2 - period + 1, e1 + 1)
    else:
        rng = np.concatenate([np.arange(1, e1 + 1), np.arange(e2, period + 1)])
    return float(np.max(corr[rng - 1]))


def acquisition_b1i(if_bytes: np.ndarray, settings, first_sample: int = 0):
    """BDS/B1I/include/acquisition.m:34-176 (resampling off, stepSize = 125 -> Nshifts = 2): two consecutive 4-ms
    blocks, per carrier shift one FFT each; per PRN and Doppler bin circshift(spectrum, bin-1) .* conj(fft(local
    code)), ifft, abs; the (shift, block, bin) with the largest peak wins by the sequential rule of :98-119; metric =
    peak / second peak outside +-1 chip within one code period (:139-160)."""
    ncodes, nblocks = 2, 4
    fs = settings.samplingFreq
    spb = int(matlab_round(fs / (settings.codeFreqBasis / (nblocks * settings.codeLength))))
    sig = [_if_complex(if_bytes, first_sample, spb), _if_complex(if_bytes, first_sample + spb, spb)]
    ts = 1.0 / fs
    phase_points = np.arange(spb) * 2 * math.pi * ts
    freq_res = fs / spb
    nbins = int(matlab_round(settings.acqSearchBand * 1e3 / freq_res)) + 1
    steps = np.arange(1, freq_res / 2 + 1e-9, 0.25)
    steps = steps[np.remainder(freq_res, steps) == 0]
    diff = steps - settings.stepSize
    m = int(np.argmin(np.abs(diff)))
    step = settings.stepSize if settings.stepSize == freq_res else (steps[m - 1] if diff[m] > 0 else steps[m])
    nshifts = int(freq_res / step)
    spc2 = int(matlab_round(fs / (settings.codeFreqBasis / (ncodes * settings.codeLength))))
    tc = 1.0 / settings.codeFreqBasis
    init_freq = settings.IF + (settings.acqSearchBand / 2) * 1000
    acq = SimpleNamespace(carrFreq=np.zeros(58), codePhase=np.zeros(58), peakMetric=np.zeros(58))
    chip = int(matlab_round(fs / settings.codeFreqBasis))
    spectra = []
    for it in range(nshifts):
        f = init_freq + it * (freq_res / nshifts)
        carr = np.exp(-1j * f * phase_points)
        spectra.append([np.fft.fft(carr * s) for s in sig])
    for prn in settings.acqSatelliteList:
        ca = generate_b1i_code(prn)
        ca2 = np.concatenate([ca, ca])
        idx = np.ceil(ts * np.arange(1, spc2 + 1) / tc).astype(np.int64)
        idx[-1] = ncodes * 2046
        local = np.concatenate([ca2[idx - 1], np.zeros(spb // ncodes)])
        code_fd = np.conj(np.fft.fft(local))
        prevmax, corr_vec, freq_shift, bin_idx = 0.0, np.zeros(spb), 0, 0
        for it in range(nshifts):
            for b in range(1, nbins + 1):
                if b == nbins and it > 0:
                    continue
                r1 = np.abs(np.fft.ifft(np.roll(spectra[it][0], b - 1) * code_fd))
                r2 = np.abs(np.fft.ifft(np.roll(spectra[it][1], b - 1) * code_fd))
                p1, p2 = r1.max(), r2.max()
                if p1 > prevmax or p2 > prevmax:
                    if p1 > p2:
                        prevmax, corr_vec = p1, r1
                    else:
                        prevmax, corr_vec = p2, r2
                    freq_shift, bin_idx = it + 1, b
        code_phase = int(np.argmax(corr_vec)) + 1
        max_peak = float(corr_vec[code_phase - 1])
        second = _second_peak_ratio(corr_vec, code_phase, chip, spb // nblocks)
        acq.peakMetric[prn - 1] = max_peak / second
        if max_peak / second > settings.acqThreshold:
            acq.codePhase[prn - 1] = code_phase
            acq.carrFreq[prn - 1] = init_freq - freq_res * (bin_idx - 1) + (freq_res / nshifts) * (freq_shift - 1)
    return acq


def acquisition_l2c(if_bytes: np.ndarray, settings, first_sample: int = 0):
    """GPS/GPS_L2C/include/acquisition.m:13-170: CM code over a 2-period block, carriers initFreq - (binIter-1)*
    freqResolution/Nshifts, bins by circshift, first/second peak metric, CL segment search with pilotTRKflag."""
    nblocks = 2
    fs = settings.samplingFreq
    spc = int(matlab_round(fs / (settings.codeFreqBasis / settings.codeLength)))
    chip = int(matlab_round(fs / settings.codeFreqBasis))
    spb = spc * nblocks
    long_signal = _if_complex(if_bytes, first_sample, if_bytes.shape[0] // 2 - first_sample)
    signal = long_signal[:spb]
    ts = 1.0 / fs
    phase_points = np.arange(spb) * 2 * math.pi * ts
    freq_res = fs / spb
    nbins = int(matlab_round(settings.acqSearchBand * 1e3 / freq_res)) + 1
    nshifts = int(freq_res / settings.acqStep)
    init_freq = settings.IF + (settings.acqSearchBand / 2) * 1000
    acq = SimpleNamespace(carrFreq=np.zeros(32), codePhase=np.zeros(32), peakMetric=np.zeros(32), CLCodePhase=np.zeros(0))
    tc = 1.0 / (settings.codeFreqBasis * 2)
    spectra = [np.fft.fft(np.exp(-1j * (init_freq - it * (freq_res / nshifts)) * phase_points) * signal) for it in range(nshifts)]
    for prn in settings.acqSatelliteList:
        cm = generate_l2c_code(prn, "CM", int(settings.codeLength))
        idx = np.ceil(ts * np.arange(spc) / tc).astype(np.int64)     # makeCMTable.m
        idx[-1] = int(settings.codeLength) * 2
        idx[0] = 1
        local = np.concatenate([cm[idx - 1], np.zeros(spc)])
        code_fd = np.conj(np.fft.fft(local))
        prevmax, corr_vec, freq_shift, bin_idx = 0.0, np.zeros(spb), 0, 0
        for it in range(nshifts):
            for b in range(1, nbins + 1):
                if b == nbins and it > 0:
                    continue
                r = np.abs(np.fft.ifft(np.roll(spectra[it], b - 1) * code_fd))
                if r.max() > prevmax:
                    prevmax, corr_vec, freq_shift, bin_idx = r.max(), r, it + 1, b
        code_phase = int(np.argmax(corr_vec)) + 1
        max_peak = float(corr_vec[code_phase - 1])
        second = _second_peak_ratio(corr_vec, code_phase, chip, spb // nblocks)
        acq.peakMetric[prn - 1] = max_peak / second
        if max_peak / second > settings.acqThreshold:
            f = init_freq - freq_res * (bin_idx - 1) - (freq_res / nshifts) * (freq_shift - 1)
            acq.carrFreq[prn - 1] = f
            acq.codePhase[prn - 1] = code_phase
            if getattr(settings, "pilotTRKflag", 0) == 1:
                s0 = long_signal[code_phase - 1:code_phase - 1 + spc]
                s0 = s0 - np.mean(s0)
                carr = np.exp(-1j * f * (np.arange(spc) * 2 * math.pi * ts))
                cl = generate_l2c_code(prn, "CL", int(settings.CLCodeLength))
                cidx = np.ceil(ts * np.arange(spc) / tc).astype(np.int64)
                cidx[0] = 1
                cidx[-1] = int(settings.codeLength) * (1 if settings.acqCohT <= 10 else 2)
                power = [abs(np.sum(s0 * cl[cidx - 1 + int(settings.codeLength) * 2 * ind] * carr)) for ind in range(75)]
                if acq.CLCodePhase.shape[0] < prn:       # the field is created by this assignment and grows with it (GPS_L2C acquisition.m:165):
                    acq.CLCodePhase = np.concatenate([acq.CLCodePhase, np.zeros(prn - acq.CLCodePhase.shape[0])])   # numel = highest PRN found
                acq.CLCodePhase[prn - 1] = int(np.argmax(power)) + 1
    return acq


def acquisition_b1c(if_bytes: np.ndarray, settings, first_sample: int = 0):
    """BDS/B1C/include/acquisition.m:108-260 (resampling off)."""
    fs = settings.samplingFreq
    long_signal = _if_complex(if_bytes, first_sample, if_bytes.shape[0] // 2 - first_sample)
    spc = int(matlab_round(fs / (settings.codeFreqBasis / settings.codeLength)))
    xlen = int(matlab_round(spc / 10 * settings.acqCohT))
    n = int(matlab_round(spc / 10 * (10 + settings.acqCohT)))
    sig = long_signal[:n]
    ts = 1.0 / fs
    phase_points = np.arange(n) * 2 * math.pi * ts
    nbins = int(matlab_round(settings.acqSearchBand * 2 / settings.acqStep)) + 1
    nmax = max(settings.acqSatelliteList)
    acq = SimpleNamespace(carrFreq=np.zeros(nmax), codePhase=np.zeros(nmax), peakMetric=np.zeros(nmax))
    nfine = int(matlab_round(settings.acqStep / 25)) * 2 + 1
    fine_phase = np.arange(spc) * 2 * math.pi * ts
    sig_power = math.sqrt(np.var(sig[:xlen], ddof=1) * xlen)
    init_freq = settings.IF + settings.acqSearchBand
    iq_fd = np.fft.fft(np.exp(-1j * init_freq * phase_points) * sig)
    pilot = getattr(settings, "pilotACQflag", 0) == 1
    tc = 1.0 / settings.codeFreqBasis / 2

    def table(code):
        idx = np.ceil(ts * np.arange(1, spc + 1) / tc).astype(np.int64)
        idx[-1] = int(settings.codeLength) * 2
        idx[0] = 1
        return code[idx - 1]
    for prn in settings.acqSatelliteList:
        dtab = table(generate_b1c_code(prn, "data"))
        dfd = np.conj(np.fft.fft(np.concatenate([dtab[:xlen], np.zeros(n - xlen)])))
        if pilot:
            ptab = table(generate_b1c_code(prn, "pilot11"))
            pfd = np.conj(np.fft.fft(np.concatenate([ptab[:xlen], np.zeros(n - xlen)])))
        results = np.empty((nbins, n))
        for b in range(1, nbins + 1):
            sh = np.roll(iq_fd, b - 1)
            results[b - 1] = np.abs(np.fft.ifft(sh * dfd))
            if pilot:
                results[b - 1] = (results[b - 1] * math.sqrt(11) + np.abs(np.fft.ifft(sh * pfd)) * math.sqrt(29)) / math.sqrt(40)
        bin_idx = int(np.argmax(results.max(axis=1))) + 1
        sel_freq = init_freq - (bin_idx - 1) * settings.acqStep
        colmax = results.max(axis=0)
        code_phase = int(np.argmax(colmax)) + 1
        acq.peakMetric[prn - 1] = float(colmax.max()) / sig_power
        if code_phase + spc - 1 > long_signal.shape[0]:
            code_phase -= spc
        if acq.peakMetric[prn - 1] > settings.acqThreshold:
            s0 = long_signal[code_phase - 1:code_phase - 1 + spc]
            xc = s0 * dtab
            fine, freqs = np.empty(nfine), np.empty(nfine)
            for k in range(nfine):
                freqs[k] = sel_freq + settings.acqStep - 25 * k
                c = np.exp(-1j * freqs[k] * fine_phase)
                fine[k] = abs(np.sum(xc * c))
                if pilot:
                    fine[k] = (fine[k] * 11 + abs(np.sum(s0 * ptab * c)) * 29) / 40
            f = float(freqs[int(np.argmax(fine))])
            acq.carrFreq[prn - 1] = f if f != 0 else 1
            acq.codePhase[prn - 1] = code_phase
    return acq


def acquisition_family_a(if_bytes: np.ndarray, settings, first_sample: int, coarse_codes, fine_codes, ncodes: int,
                         fine_step: float, combine, secondary=None, n_results: int = 32, boc: bool = False,
                         index_offset: int = 1):
    """GPS_L5C / GAL_E5a / BDS B2a acquisition.m (resampling off): the L1CA coarse scheme with the per-PRN replica
    list `coarse_codes(prn)` summed (|ifft| of each), metric peak/sigPower/acqNonCohTime, then the package's fine stage
    on `fine_codes(prn)`: codeValueIndex = floor(ts*(1:K*spc)/tc), per-code sums, and
      combine "circular": max over circular shifts of secondary(prn) of |sum(sumPerCode .* shifted)| (L5 NH20, E5a CS100)
      combine "noncoh":   sum(abs(sumPerCode1)) + sum(abs(sumPerCode2))                               (B2a :207)
      combine "split":    secondary aligned, then shifts k with |sum(1:k)| + |sum(k+1:end)|  (E1C :237-245, B3I MEO)
      combine callable(prn, per_code) for anything else (B3I GEO pairing :252-256); fine_step 0: no fine stage (E5b).
    boc: half-chip tables and fine code at 2*codeFreqBasis over 2*codeLength entries (GAL_E1C); index_offset: the fine
    codeValueIndex runs over (1:K*spc) (1) or (0:K*spc-1) (0)."""
    long_signal = _if_complex(if_bytes, first_sample, if_bytes.shape[0] // 2 - first_sample)
    fs = settings.samplingFreq
    spc = samples_per_code(settings)
    ts, tc = 1.0 / fs, 1.0 / settings.codeFreqBasis / (2 if boc else 1)
    clen = int(settings.codeLength) * (2 if boc else 1)
    acq = SimpleNamespace(carrFreq=np.zeros(n_results), codePhase=np.zeros(n_results), peakMetric=np.zeros(n_results))
    sig_power = math.sqrt(np.var(long_signal[:spc], ddof=1) * spc)
    nfine = int(matlab_round(settings.acqSearchStep / fine_step)) + 1 if fine_step else 0
    fine_phase = np.arange(ncodes * spc) * 2 * math.pi * ts
    idx_t = np.ceil(ts * np.arange(1, spc + 1) / tc).astype(np.int64)
    idx_t[-1] = clen
    if boc:
        idx_t[0] = 1
    for prn in settings.acqSatelliteList:
        tables = [c[idx_t - 1] for c in coarse_codes(prn)]
        results = acquisition_coarse_results(long_signal, prn, settings, tables=tables)
        coarse_bin = int(np.argmax(results.max(axis=1))) + 1
        colmax = results.max(axis=0)
        code_phase = int(np.argmax(colmax)) + 1
        acq.peakMetric[prn - 1] = float(colmax.max()) / sig_power / settings.acqNonCohTime
        if acq.peakMetric[prn - 1] > settings.acqThreshold:
            coarse_freq = settings.IF + settings.acqSearchBand - settings.acqSearchStep * (coarse_bin - 1)
            if not fine_step:
                acq.carrFreq[prn - 1] = coarse_freq
                acq.codePhase[prn - 1] = code_phase
                continue
            cvi = np.floor(ts * (np.arange(ncodes * spc) + index_offset) / tc).astype(np.int64)
            longs = [c[np.remainder(cvi, clen)] for c in fine_codes(prn)]
            sig = long_signal[code_phase - 1:code_phase - 1 + ncodes * spc]
            fine, freqs = np.empty(nfine), np.empty(nfine)
            for k in range(nfine):
                freqs[k] = coarse_freq + settings.acqSearchStep / 2 - fine_step * k
                carr = np.exp(-1j * freqs[k] * fine_phase)
                per_code = [(lc * carr * sig).reshape(ncodes, spc).sum(axis=1) for lc in longs]
                if callable(combine):
                    fine[k] = combine(prn, per_code)
                elif combine == "split":
                    sec = np.asarray(secondary(prn), dtype=np.float64)
                    best = abs(np.sum(per_code[0] * sec))
                    for kk in range(1, sec.shape[0]):
                        t = per_code[0] * np.roll(sec, kk)
                        best = max(best, abs(np.sum(t[:kk])) + abs(np.sum(t[kk:])))
                    fine[k] = best
                elif combine == "circular":
                    sec = np.asarray(secondary(prn), dtype=np.float64).copy()
                    best = 0.0
                    for _ in range(sec.shape[0]):
                        best = max(best, abs(np.sum(per_code[0] * sec)))
                        sec = np.roll(sec, 1)
                    fine[k] = best
                else:
                    fine[k] = sum(float(np.sum(np.abs(pc))) for pc in per_code)
            f = float(freqs[int(np.argmax(fine))])
            acq.carrFreq[prn - 1] = f if f != 0 else 1
            acq.codePhase[prn - 1] = code_phase
    return acq


def acquisition_glo(if_bytes: np.ndarray, settings, first_sample: int = 0):
    """GLO/GLO_GL1/include/acquisition.m:120-200 (resampling off)."""
    long_signal = _if_complex(if_bytes, first_sample, if_bytes.shape[0] // 2 - first_sample)
    fs = settings.samplingFreq
    spc = samples_per_code(settings)
    ts = 1.0 / fs
    code = generate_glo_code()

    def sampled(n):
        step = 511e3 / fs
        s = np.floor(colon(0.0, step, n * step - step)).astype(np.int64)
        return code[np.remainder(s, 511)]
    table = sampled(spc)
    code40 = sampled(spc * 40)
    nfine = int(matlab_round(settings.acqSearchStep / 25)) + 1
    fine_phase = np.arange(40 * spc) * 2 * math.pi * ts
    sig_power = math.sqrt(np.var(long_signal[:spc], ddof=1) * spc)
    acq = SimpleNamespace(carrFreq=np.zeros(21), codePhase=np.zeros(21), peakMetric=np.zeros(21))
    s2 = SimpleNamespace(**vars(settings))
    for K in settings.acqSatelliteList:
        s2.IF = settings.IF - settings.freqSpacing * K
        results = acquisition_coarse_results(long_signal, 0, s2, tables=[table])
        coarse_bin = int(np.argmax(results.max(axis=1))) + 1
        colmax = results.max(axis=0)
        code_phase = int(np.argmax(colmax)) + 1
        acq.peakMetric[K + 7] = float(colmax.max()) / sig_power / settings.acqNonCohTime
        if acq.peakMetric[K + 7] > settings.acqThreshold:
            coarse_freq = s2.IF + settings.acqSearchBand - settings.acqSearchStep * (coarse_bin - 1)
            x = long_signal[code_phase - 1:code_phase - 1 + 40 * spc] * code40
            fine, freqs = np.empty(nfine), np.empty(nfine)
            for k in range(nfine):
                freqs[k] = coarse_freq + settings.acqSearchStep / 2 - 25 * k
                per_code = (x * np.exp(-1j * freqs[k] * fine_phase)).reshape(40, spc).sum(axis=1)
                fine[k] = max(abs(np.sum(per_code[c:c + 10]) - np.sum(per_code[c + 10:c + 20])) for c in range(20))
            acq.carrFreq[K + 7] = float(freqs[int(np.argmax(fine))])
            acq.codePhase[K + 7] = code_phase
    return acq


def acquisition_front_end(long_signal: np.ndarray, settings):
    """acquisition.m:46-111 (SURVEY.md §8a row A0 — restated on the CPU only; `resamplingflag` is 0 in every package's
    initSettings and the hot path never takes this branch): FIR(700) band-pass around IF with zero-phase filtering,
    then band-pass-sampling decimation by index selection and IF remapping.
    Returns (long_signal', settings') — settings' carries the new samplingFreq / IF plus oldFreq / oldIF — or the
    inputs unchanged when the branch is not taken.
    fir1(700, wp) = Hamming-window design scaled to unit gain at the pass-band centre = scipy.signal.firwin(701, wp,
    pass_zero=False); filtfilt(b, 1, x) pads with 3*(length(b)-1) odd-reflected samples at both ends."""
    if not (settings.samplingFreq > settings.resamplingThreshold and getattr(settings, "resamplingflag", 0) == 1):
        return long_signal, settings
    from scipy.signal import filtfilt, firwin
    fs = settings.samplingFreq
    bw = settings.codeFreqBasis * 2 + 0.5e6                      # :50
    w1, w2 = settings.IF - bw / 2, settings.IF + bw / 2           # :52-53
    b = firwin(701, [w1 * 2 / fs, w2 * 2 / fs], pass_zero=False)  # :56-58
    x = filtfilt(b, [1.0], long_signal, padtype="odd", padlen=3 * (b.size - 1))   # :60
    fu = settings.IF + bw / 2                                     # :63
    n = max(1, int(math.floor(fu / bw)))                          # :66-69
    lower = 2 * fu / n                                            # :70
    fl = settings.IF - bw / 2
    upper = 2 * fl / (n - 1) if n > 1 else lower                  # :73-77
    s2 = SimpleNamespace(**vars(settings))
    s2.oldFreq = fs
    s2.samplingFreq = float(math.ceil((lower + upper) / 2))       # :81
    sig_len = int(math.floor((x.shape[0] - 1) / fs * s2.samplingFreq))   # :84
    index = np.ceil(np.arange(sig_len) / s2.samplingFreq * fs).astype(np.int64)   # :87
    index[0] = 1
    s2.oldIF = settings.IF
    s2.IF = math.fmod(settings.IF, s2.samplingFreq)               # :95 rem()
    return x[index - 1], s2


def calc_cno_pld(i_p, q_p, pilot_i_p, pilot_q_p, loop_cnt: int, interval: int, int_time: float, pilot_flag: int):
    """BDS/B2a/include/Calc_CNo_PLD.m:35-97 and BDS/B1C/include/Calc_CNo_PLD.m (same file plus the pilotTRKflag == 2
    branch): returns (CNo[3], PllDetector[2]).  Per arm over epochs loop_cnt-interval+1 .. loop_cnt (1-based):
    Z = I^2+Q^2, Pav = sqrt(mean(Z)^2 - var(Z)), Nv = (mean(Z) - Pav)/2, C/N0 = |Pav / (2 Nv T)|;
    NBP/NBD from sum|I| (data wipe-off by sign) and sum Q.  The pilot arm reads (I, Q) = (Pilot_Q_P, Pilot_I_P) when
    pilot_flag == 1 and (Pilot_I_P, Pilot_Q_P) when pilot_flag == 2."""
    lo = loop_cnt - interval

    def one(I, Q):
        zs = [float(a) * float(a) + float(b) * float(b) for a, b in zip(I, Q)]
        zm = math.fsum(zs) / len(zs)
        zv = math.fsum((z - zm) ** 2 for z in zs) / (len(zs) - 1)
        pav = cmath.sqrt(zm * zm - zv)
        nv = 0.5 * (zm - pav)
        lin = abs((1.0 / int_time) * pav / (2.0 * nv))
        wiped = math.fsum(abs(float(a)) for a in I)
        sq = math.fsum(float(b) for b in Q)
        return lin, (wiped * wiped - sq * sq) / (wiped * wiped + sq * sq)

    cno = [0.0, 0.0, 0.0]
    pld = [0.0, 0.0]
    data, pld[0] = one(i_p[lo:loop_cnt], q_p[lo:loop_cnt])
    cno[0] = 10.0 * math.log10(data)
    pilot = 0.0
    if pilot_flag in (1, 2):
        a, b = pilot_i_p[lo:loop_cnt], pilot_q_p[lo:loop_cnt]
        pilot, pld[1] = one(a, b) if pilot_flag == 2 else one(b, a)
        cno[1] = 10.0 * math.log10(pilot)
    cno[2] = 10.0 * math.log10(data + pilot)
    return cno, pld


def unpack_cplx(packed: np.ndarray) -> np.ndarray:
    """GPS/GPS_L5C/include/unpack_cplx.m:32-49 as arithmetic instead of four 256-entry tables: each byte holds two
    complex samples of 2-bit sign-magnitude components — I1: sign bit 0, magnitude bit 2; Q1: bits 1, 3; I2: bits 4, 6;
    Q2: bits 5, 7; value = (1 + 2*magnitude) * (1 - 2*sign).  Returns the schar stream I1, Q1, I2, Q2, ..."""
    v = np.asarray(packed, dtype=np.uint8).astype(np.int64)

    def dec(sbit, mbit):
        return (1 + 2 * ((v >> mbit) & 1)) * (1 - 2 * ((v >> sbit) & 1))
    out = np.empty(4 * v.shape[0], dtype=np.int8)
    out[0::4], out[1::4], out[2::4], out[3::4] = dec(0, 2), dec(1, 3), dec(4, 6), dec(5, 7)
    return out


def nav_parity_check(ndat) -> int:
    """Common/navPartyChk.m in its own +-1 product form."""
    n = [int(v) for v in ndat]
    if n[1] != 1:
        for k in range(2, 26):
            n[k] = -n[k]
    d = [None] + n   # 1-based like the .m

    def prod(ix):
        r = 1
        for k in ix:
            r *= d[k]
        return r
    parity = [prod([1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22, 25]),
              prod([2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23, 26]),
              prod([1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24]),
              prod([2, 4, 6, 7, 8, 10, 11, 15, 16, 17, 18, 19, 22, 23, 25]),
              prod([2, 3, 5, 7, 8, 9, 11, 12, 16, 17, 18, 19, 20, 23, 24, 26]),
              prod([1, 5, 7, 8, 10, 11, 12, 13, 15, 17, 21, 24, 25, 26])]
    return -n[1] if parity == n[26:32] else 0


def find_subframe_start(i_p: np.ndarray, ms_to_process: int, search_start_offset: int = 0):
    """GPS/GPS_L1CA/include/NAVdecoding.m:55-100: hard-limited prompt stream, xcorr with the 8-bit preamble stretched to
    160 samples, |.| > 153, candidates 6000 ms apart, parity of the first two words."""
    i_p = np.asarray(i_p, dtype=np.float64)
    pre = np.kron(np.array([1, -1, -1, -1, 1, -1, 1, 1], dtype=np.float64), np.ones(20))
    bits = np.where(i_p[search_start_offset:] > 0, 1.0, -1.0)
    n = bits.shape[0]
    full = np.correlate(np.concatenate([bits, np.zeros(pre.shape[0])]), pre, mode="valid")[:n]   # xcorr's lags 0..n-1
    index = np.flatnonzero(np.abs(full) > 153) + 1 + search_start_offset
    index = index[(index > 40) & (index < ms_to_process - (20 * 60 - 1))]
    for i in index:
        if np.any(index - i == 6000):
            b = i_p[i - 40 - 1:i + 20 * 60 - 1].reshape(-1, 20).sum(axis=1)
            b = np.where(b > 0, 1, -1)
            if nav_parity_check(b[0:32]) != 0 and nav_parity_check(b[30:62]) != 0:
                return int(i), full
    return None, full


# --------------------------------------------------------------------------------------
# Bit / frame synchronisation of every package (the block NAVdecoding.m starts with)
# --------------------------------------------------------------------------------------
_NH20 = np.array([-1, -1, -1, -1, -1, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, 1, 1, 1, -1], dtype=np.float64)
_BDS_PRE = np.array([1, 1, 1, -1, -1, -1, 1, -1, -1, 1, -1], dtype=np.float64)


def xcorr_nonneg(bits: np.ndarray, pattern: np.ndarray) -> np.ndarray:
    """tlmXcorrResult(xcorrLength : 2*xcorrLength - 1) of xcorr(bits, pattern): lags 0 .. n-1, the shorter input zero-padded."""
    n = bits.shape[0]
    return np.correlate(np.concatenate([bits, np.zeros(pattern.shape[0])]), pattern, mode="valid")[:n]


def bch_15_11_errors(bits15) -> int:
    """cnumerr of bchdec(gf(bits, 1), 15, 11) for hard bits, first = highest power of x: 0 when g(x) = x^4 + x + 1 divides the
    word, else 1 (the (15,11) Hamming code corrects every single error, so a non-zero syndrome always decodes)."""
    r = 0
    for b in bits15:
        r = (r << 1) | int(b)
        if r & 0x10:
            r ^= 0x13
    return 0 if r == 0 else 1


def nav_sync(package: str, i_p: np.ndarray, ms_to_process: int, prn: int = 0):
    """The synchronisation block of <package>/include/NAVdecoding.m on one channel's prompt stream.  Returns
    (xcorr over the non-negative lags, index as it stands when the candidate loop starts, candidates passing the spacing rule,
    first verified start or None).  Lines restated:
      GPS_L1CA  GPS/GPS_L1CA/include/NAVdecoding.m:66-145      GAL_E1C  GAL/GAL_E1C/include/NAVdecoding.m:59,79-110
      GAL_E5a   GAL/GAL_E5a/include/NAVdecoding.m:54,69-108    GAL_E5b  GAL/GAL_E5b/include/NAVdecoding.m:59,80-119
      BDS_B1I   BDS/B1I/include/NAVdecoding.m:68-170           BDS_B3I  BDS/B3I/include/NAVdecoding.m:69-164
      GLO_GL1   GLO/GLO_GL1/include/NAVdecoding.m:66-105 (GLO_GL2: the same file)"""
    x = np.asarray(i_p, dtype=np.float64).reshape(-1)
    n = x.shape[0]
    off = 1000 if package in ("BDS_B1I", "BDS_B3I") else 0
    seg = x[off:]
    if package == "GAL_E1C":
        bits = 1.0 - 2.0 * (seg < 0)                                     # :84-85
    else:
        bits = np.where(seg > 0, 1.0, -1.0)                              # bits(bits > 0) = 1; bits(bits <= 0) = -1
    c = 1
    if package == "GPS_L1CA":
        pat = np.kron(np.array([1, -1, -1, -1, 1, -1, 1, 1.0]), np.ones(20))
    elif package == "GAL_E1C":
        pat = np.array([1, -1, 1, -1, -1, 1, 1, 1, 1, 1.0])
    elif package == "GAL_E5a":
        cs = 1.0 - 2.0 * np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 1, 1, 0, 1, 0, 0, 1])
        pat = np.kron(np.array([-1, 1, -1, -1, 1, -1, -1, -1, 1, 1, 1, 1.0]), cs)
    elif package == "GAL_E5b":
        pat = np.kron(np.array([1, -1, 1, -1, -1, 1, 1, 1, 1, 1.0]), np.array([-1, -1, -1, 1.0]))
    elif package in ("BDS_B1I", "BDS_B3I"):
        geo = prn <= 5 if package == "BDS_B1I" else (1 <= prn <= 5 or 59 <= prn <= 63)
        c = 2 if geo else 20
        pat = np.kron(_BDS_PRE, np.ones(2)) if geo else np.kron(_BDS_PRE, -_NH20)
    elif package in ("GLO_GL1", "GLO_GL2"):
        pat = np.kron(np.array([1, 1, 1, 1, 1, -1, -1, -1, 1, 1, -1, 1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, 1, -1, -1, 1, -1, 1, 1, -1.0]), np.ones(10))
    else:
        raise ValueError(package)
    r = xcorr_nonneg(bits, pat)
    a = np.abs(r)
    rnd = np.floor(a + 0.5)
    hit = {"GPS_L1CA": a > 153, "GAL_E1C": rnd >= 9.99, "GAL_E5a": rnd >= 239.99, "GAL_E5b": a > 39.99,
           "BDS_B1I": a >= c * 10, "BDS_B3I": a >= c * 10, "GLO_GL1": a > 271, "GLO_GL2": a > 271}[package]
    index = np.flatnonzero(hit) + 1 + off + (300 if package.startswith("GLO") else 0)
    first = None
    if package == "GPS_L1CA":
        index = index[(index > 40) & (index < ms_to_process - (20 * 60 - 1))]
        cand = [int(i) for i in index if np.any(index - i == 6000)]
        for i in cand:
            b = x[i - 40 - 1:i + 20 * 60 - 1].reshape(-1, 20).sum(axis=1)
            b = np.where(b > 0, 1, -1)
            if nav_parity_check(b[0:32]) != 0 and nav_parity_check(b[30:62]) != 0:
                first = i
                break
    elif package == "GAL_E1C":
        cand = [int(i) for i in index if np.any(index - i == 250) and np.any(index - i == 500) and i < ms_to_process / 4 - 7500]
    elif package == "GAL_E5a":
        cand = [int(i) for i in index if np.any(np.abs(index - i) == 10e3)]      # :102-108: index = newIndex
        index = np.array(cand, dtype=np.int64)
    elif package == "GAL_E5b":
        cand = [int(i) for i in index if np.any(index - i == 250 * 4) and bits[i - 1:].shape[0] > 7500 * 4]
    elif package in ("BDS_B1I", "BDS_B3I"):
        if package == "BDS_B3I":
            index = index[index < ms_to_process - 1500 * 20 + 300 * c]
        cand = [int(i) for i in index if np.any(index - i == 300 * c)]
        for i in cand:
            if i + 30 * c - 1 > n:
                break
            b = x[i - 1:i + 30 * c - 1].reshape(-1, c).sum(axis=1)
            b = (b > 0).astype(np.int64)
            if bch_15_11_errors(b[15:30]) == 0:
                first = i
                break
    else:
        cand = [int(i) for i in index if np.any(index - i == 2000)]
    return r, index.astype(np.int64), np.array(cand, dtype=np.int64), first
