"""Per-signal parameters of the tracking hot path (SURVEY.md §8a signal-variant matrix): which replica
tables a channel carries, the index scale R of the ramps, the carrier loop filter and how the pilot
arm enters the discriminators.  One entry per reference package that is wired up so far."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable

import numpy as np

from . import _lib as L
from . import codes


@dataclass(frozen=True)
class SignalSpec:
    name: str
    tables: Callable            # (PRN, settings) -> list of padded int8 tables [c(end) c c(1)], arm 0 = data
    index_scale: float          # R: 1, or 2 for BOC(1,1) half-chip tables (GAL_E1C/include/tracking.m:236)
    pll_kind: int               # L.GC_PLL_2ND_ORDER | L.GC_PLL_3_STATE
    coef_variant: str           # calcLoopCoefCarr.m variant "a" (a3=b3=2, Wn=1.2*LBW) | "b" (1.1, 2.4, LBW/0.7845)
    pilot_combine: int          # 0 none | 1 rotate pilot by -pi/2 then average | 2 plain average
    code_freq_from_channel: bool  # initial codeFreq = channel.codeFreq (GPS_L5C tracking.m:165) or codeFreqBasis
    arm_mult: tuple | None = None          # per-arm ramp multipliers (B1C wide-band: (1, 1, 6), WB_tracking.m:293)
    pll_weight: tuple | Callable | None = None   # (data, pilot) discriminator weights; None = plain average
    dll_weight: tuple | Callable | None = None   # tuple or settings -> tuple (B1C WB: CalcWeighingFactor)
    dll_scale_spacing: bool = False        # DLL discriminators times (1 - earlyLateSpc), NB_tracking.m:346-348
    windows: tuple | None = None           # per-arm LDS staging window in entries (L2C CL: one block's worth)
    doubled_code: bool = False             # GPS L2C: the loop runs on the RZ-doubled code (tracking.m:107-109,171)
    int16_branch: bool = False             # tracking.m has the int16 seek / ftell branch (GPS_L1CA tracking.m:145-148,212-213;
                                           # also GAL_E5a, GAL_E5b, BDS/B3I); the other packages assume one byte per component
    recorded_pilot: str = "prompt"         # which Pilot_* fields the package's trackResults holds when the pilot is tracked: "prompt"
                                           # (Pilot_I_P, Pilot_Q_P: GPS_L5C tracking.m:71-74, B2a, E5a, E5b, B1C NB), "all" six (BDS/B1C
                                           # WB_tracking.m:79-86, GPS_L2C tracking.m:397-402), "none" (GAL_E1C records the data arm only)
    id_field: str = "PRN"                  # channel field naming the satellite: 'K' for GLONASS (GLO_GL1 preRun.m:66), whose
                                           # channels are active when status ~= '-' (K = 0 is a valid frequency number, tracking.m:138)


def calcLoopCoefCarr(settings, variant: str = "a"):
    """(pf3, pf2, pf1) of Common/calcLoopCoefCarr.m (two variants in the tree, see SignalSpec)."""
    lbw, t = settings.pllNoiseBandwidth, settings.intTime
    a3, b3, wn = (2, 2, 1.2 * lbw) if variant == "a" else (1.1, 2.4, lbw / 0.7845)
    return wn ** 3 * t ** 2, a3 * wn ** 2 * t, b3 * wn


def _e1_cboc_tables(prn, settings):
    # arms {E1-B BOC(1,1), E1-C BOC(1,1), E1-C BOC(6,1)}: the pilot's two subcarrier components are correlated apart
    # (int8 tables, ramp multipliers 1 and 6) and folded with sqrt(10/11), -sqrt(1/11) on the host (pilot_combine 5)
    return [codes.padded_table(codes.generateE1Bcode(prn)), codes.padded_table(codes.generateE1Ccode(prn)),
            codes.padded_table(codes.generateE1C_BOC61(prn))]


def _l1ca_tables(prn, settings):
    return [codes.padded_table(codes.generateCAcode(prn))]                   # GPS_L1CA tracking.m:156-158


def _e1_tables(prn, settings):
    t = [codes.padded_table(codes.generateE1Bcode(prn))]                      # GAL_E1C tracking.m:139-143
    if getattr(settings, "pilotTRKflag", 0) == 1:
        t.append(codes.padded_table(codes.generateE1Ccode(prn)))              # :145-150
    return t


def _l5_tables(prn, settings):
    t = [codes.padded_table(codes.generateL5Icode(prn))]                      # GPS_L5C tracking.m:150-152
    if getattr(settings, "pilotTRKflag", 0) == 1:
        t.append(codes.padded_table(codes.generateL5Qcode(prn)))              # :154-159
    return t


def _glo_tables(prn, settings):
    return [codes.padded_table(codes.generateGLOcode())]                      # GLO_GL1 tracking.m:88-89 (one code for all)


def _b1i_tables(prn, settings):
    return [codes.padded_table(codes.generateCAcode53(prn))]                  # BDS/B1I tracking.m:144-145


def _b2a_tables(prn, settings):
    t = [codes.padded_table(codes.generateB2aDataCode(prn))]                 # BDS/B2a tracking.m:156-158
    if getattr(settings, "pilotTRKflag", 0) == 1:
        t.append(codes.padded_table(codes.generateB2aPilotCode(prn)))         # :160-165
    return t


def _b3i_tables(prn, settings):
    return [codes.padded_table(codes.generateB3Icode(prn))]                  # BDS/B3I tracking.m:150-152


def _e5_tables(i_fn, q_fn):
    def tables(prn, settings):
        L = int(settings.codeLength)
        tiered = i_fn(prn, 2)
        # GAL_E5a tracking.m:148-150: [E5aICode(codeLength) E5aICode E5aICode(1)] of the TIERED code, of which the
        # ramps only ever reach the first codeLength + 2 entries (entry L+1 is tiered chip L+1 = primary chip 1 times
        # the SECOND secondary chip: not the periodic wrap of the first period)
        data = np.concatenate([tiered[L - 1:L], tiered[:L + 1]]).astype(np.int8)
        t = [data]
        if getattr(settings, "pilotTRKflag", 0) == 1:
            t.append(codes.padded_table(q_fn(prn, 1)))                        # :152-156, primary code only
        return t
    return tables


def _l2c_tables(prn, settings):
    cm = codes.generateCMcode(prn, int(settings.codeLength))                  # GPS_L2C tracking.m:155-156
    t = [codes.padded_table(cm)]
    if getattr(settings, "pilotTRKflag", 0):
        t.append(codes.padded_table(codes.generateCLcode(prn, int(settings.CLCodeLength))))   # :158-162
    return t


def _b1c_nb_tables(prn, settings):
    return [codes.padded_table(codes.generateDataBOC11(prn)),                # BDS/B1C NB_tracking.m:161-164
            codes.padded_table(codes.generatePilotBOC11(prn))]


def _b1c_wb_tables(prn, settings):
    return _b1c_nb_tables(prn, settings) + [codes.padded_table(codes.generatePilotBOC61(prn))]   # WB_tracking.m:176-188


def CalcWeighingFactor(settings) -> float:
    """BDS/B1C/include/CalcWeighingFactor.m: weight of the data-component DLL discriminator in the wide-band
    tracker, from the RMS bandwidths of BOC(1,1) and of the 29/33 BOC(1,1) + 4/33 BOC(6,1) pilot inside the
    front-end bandwidth settings.FEBW.  MATLAB's integral() -> scipy.integrate.quad (host-side, once per run)."""
    from scipy.integrate import quad
    fc = settings.codeFreqBasis
    tc = 1.0 / fc
    br = settings.FEBW

    def g11(f):
        f = f if f != 0.0 else 1e-9
        return tc * (math.sin(math.pi / 2 * f / fc) * math.sin(math.pi * f / fc) / math.cos(math.pi / 2 * f / fc) * fc / f / math.pi) ** 2

    def g61(f):
        f = f if f != 0.0 else 1e-9
        return tc * (math.sin(math.pi / 12 * f / fc) * math.sin(math.pi * f / fc) / math.cos(math.pi / 12 * f / fc) * fc / f / math.pi) ** 2

    def integ(fn):
        return quad(fn, -br / 2, br / 2, limit=400, epsabs=0, epsrel=1e-10)[0]

    p11_2, p11 = integ(lambda f: g11(f) * f * f), integ(g11)
    pil_2 = integ(lambda f: (29 / 33 * g11(f) + 4 / 33 * g61(f)) * f * f)
    pil = integ(lambda f: 29 / 33 * g11(f) + 4 / 33 * g61(f))
    t1 = 11 * p11 * (p11_2 / p11)      # 11 * Power * RMS_BW^2
    t2 = 33 * pil * (pil_2 / pil)
    return t1 / (t1 + t2)


def _b1c_wb_dll_weight(settings):
    f = getattr(settings, "dllWeighingFactor", None)
    if f is None:
        f = CalcWeighingFactor(settings)
    return (f, 1.0 - f)


SIGNALS = {
    "GPS_L1CA": SignalSpec("GPS_L1CA", _l1ca_tables, 1.0, L.GC_PLL_2ND_ORDER, "a", 0, False, int16_branch=True),
    # pilot_combine is applied only when settings.pilotTRKflag == 1 (see receiver.tracking)
    "GAL_E1C": SignalSpec("GAL_E1C", _e1_tables, 2.0, L.GC_PLL_3_STATE, "a", 2, False, recorded_pilot="none"),
    # BASELINE config 3: the E1-C pilot tracked with its CBOC(6,1,1/11) subcarrier (needs pilotTRKflag = 1 and a narrow
    # correlator: dllCorrelatorSpacing * 12 < 1 table entry); an extension, the reference's package stops at BOC(1,1)
    "GAL_E1C_CBOC": SignalSpec("GAL_E1C_CBOC", _e1_cboc_tables, 2.0, L.GC_PLL_3_STATE, "a", 5, False, arm_mult=(1.0, 1.0, 6.0), recorded_pilot="none"),
    # GLONASS: the record must be loaded with layout GC_QI (GLO_GL1 tracking.m:227 swaps the components);
    # channel.PRN carries the frequency number K (GLO_GL1 preRun.m:66), the FDMA offset lives in acquiredFreq
    "GLO_GL1": SignalSpec("GLO_GL1", _glo_tables, 1.0, L.GC_PLL_3_STATE, "a", 0, False, id_field="K"),
    # GLONASS L2OF: the same package with freqSpacing 437.5 kHz (GLO_GL2/initSettings.m:73; tracking.m identical to GLO_GL1's)
    "GLO_GL2": SignalSpec("GLO_GL2", _glo_tables, 1.0, L.GC_PLL_3_STATE, "a", 0, False, id_field="K"),
    "BDS_B1I": SignalSpec("BDS_B1I", _b1i_tables, 1.0, L.GC_PLL_3_STATE, "a", 0, False),
    "GPS_L5C": SignalSpec("GPS_L5C", _l5_tables, 1.0, L.GC_PLL_3_STATE, "a", 1, True),
    # the 10.23-Mcps family shares GPS L5's loop closure (pilot rotated by -pi/2, discriminators averaged) and
    # differs in codes, loop-coefficient variant (Common/calcLoopCoefCarr.m of each package) and bandwidths
    "BDS_B2a": SignalSpec("BDS_B2a", _b2a_tables, 1.0, L.GC_PLL_3_STATE, "a", 1, True),
    "BDS_B3I": SignalSpec("BDS_B3I", _b3i_tables, 1.0, L.GC_PLL_3_STATE, "b", 0, True, int16_branch=True),
    "GAL_E5a": SignalSpec("GAL_E5a", _e5_tables(codes.generateE5aIcode, codes.generateE5aQcode), 1.0, L.GC_PLL_3_STATE, "a", 1, True, int16_branch=True),
    # BDS B1C: 10-ms blocks, BOC(1,1) half-chip tables (R = 2).  Narrow-band: data + pilot BOC(1,1), pilot in
    # quadrature, 11:29 weights; wide-band: + the pilot's BOC(6,1) arm read through ceil(6*t), folded 1:3 / factor
    "BDS_B1C_NB": SignalSpec("BDS_B1C_NB", _b1c_nb_tables, 2.0, L.GC_PLL_3_STATE, "b", 3, True,
                             pll_weight=(11.0, 29.0), dll_weight=(11.0, 29.0), dll_scale_spacing=True),
    "BDS_B1C_WB": SignalSpec("BDS_B1C_WB", _b1c_wb_tables, 2.0, L.GC_PLL_3_STATE, "b", 4, True, arm_mult=(1.0, 1.0, 6.0),
                             pll_weight=(1.0, 3.0), dll_weight=_b1c_wb_dll_weight, dll_scale_spacing=True, recorded_pilot="all"),
    # GPS L2C: RZ-interleaved CM (+ CL through a moving window of its 1.5-s table), everything in doubled-code units
    "GPS_L2C": SignalSpec("GPS_L2C", _l2c_tables, 1.0, L.GC_PLL_3_STATE, "a", 2, False, windows=(0, 20464), doubled_code=True, recorded_pilot="all"),
    "GAL_E5b": SignalSpec("GAL_E5b", _e5_tables(codes.generateE5bIcode, codes.generateE5bQcode), 1.0, L.GC_PLL_3_STATE, "b", 1, True, int16_branch=True),
}


def epochs_to_process(settings) -> int:
    """NumToProcess = round(msToProcess/1000/intTime) (GAL_E1C tracking.m:51); = msToProcess for 1-ms codes."""
    x = settings.msToProcess / 1000 / settings.intTime
    return int(np.floor(x + 0.5))
