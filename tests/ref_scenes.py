"""Synthetic scenes shared by tests/golden/make_ref_vectors.py (which runs the REFERENCE's own .m files on them through
oracle/mlab, in the build container) and by the tests that compare the oracle and the HIP path with the stored results.

A scene is fully determined by seeds: the tests regenerate the same record bytes (the fixture holds their CRC-32) and the same
channel / settings values, so nothing but numbers travels to the GPU box."""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from types import SimpleNamespace

import numpy as np

GC_REAL, GC_IQ, GC_QI = 0, 1, 2
TRACK_FIELDS = ("absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L", "dllDiscr", "dllDiscrFilt",
                "pllDiscr", "pllDiscrFilt", "remCodePhase", "remCarrPhase")
PILOT_FIELDS = ("Pilot_I_E", "Pilot_Q_E", "Pilot_I_P", "Pilot_Q_P", "Pilot_I_L", "Pilot_Q_L")


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@dataclass
class TrackScene:
    name: str                     # fixture: tests/golden/ref_track_<name>.npz
    pkg: str                      # reference package directory, relative to the reference root
    signal: str                   # cu_sdr_collection_amd.signals.SIGNALS key
    settings_fn: str              # cu_sdr_collection_amd.settings function mirroring the package's initSettings.m
    overrides: dict               # settings fields changed for the scene (applied to the mirror AND to the reference's struct)
    build: object                 # (P, S) -> (record ndarray, layout, channels list)
    fn: str = "tracking"          # function file of the package (BDS/B1C: NB_tracking / WB_tracking)
    pilot: bool = False
    oracle: object = None         # (O, record, channels, S) -> list of trackResults from the oracle
    extra_fields: tuple = ()
    notes: str = ""


def _sats(P, prns, seed, period, cn0, dmax=3e3):
    rng = np.random.default_rng(seed)
    return [P.synth.SatSpec(prn=p, doppler=float(rng.uniform(-dmax, dmax)), code_phase_samples=float(rng.uniform(0, period)),
                            carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=cn0) for p in prns]


def _channels(S, sats, df, code_freq=False, pad=1, extra=None):
    ch = []
    for s in sats:
        f = S.IF + s.doppler + df
        c = SimpleNamespace(PRN=s.prn, acquiredFreq=f, codePhase=int(np.ceil(s.code_phase_samples)) + 1, status="T")
        if code_freq:
            c.codeFreq = S.codeFreqBasis + (f - S.IF) / S.carrFreqBasis * S.codeFreqBasis      # preRun.m of the package
        if extra:
            extra(c, s)
        ch.append(c)
    for _ in range(pad):
        c = SimpleNamespace(PRN=0, acquiredFreq=0.0, codePhase=0, status="-")
        if code_freq:
            c.codeFreq = 0.0
        if extra:
            extra(c, None)
        ch.append(c)
    return ch


# ---- record / channel builders -------------------------------------------------------------------------------------
def _l1ca(P, S):
    sats = _sats(P, (7, 19), 1001, 18000, 47.0, 5e3)
    iq = P.synth.generate_if(sats, int((S.msToProcess + 6) * 1e-3 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode,
                             S.codeFreqBasis, 1023, seed=1002)
    return iq, GC_IQ, _channels(S, sats, 3.0)


def _l1ca_int16_skip(P, S):
    iq, _, ch = _l1ca(P, S)
    return (iq.astype(np.int16) * 5), GC_IQ, ch


def _l1ca_real(P, S):
    sats = _sats(P, (5, 23), 1011, 18000, 50.0, 5e3)
    iq = P.synth.generate_if(sats, int((S.msToProcess + 6) * 1e-3 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode,
                             S.codeFreqBasis, 1023, seed=1012)
    return np.ascontiguousarray(iq[0::2]), GC_REAL, _channels(S, sats, 3.0)


def _e1(P, S):
    sats = _sats(P, (4, 19), 1021, 72000, 48.0)
    iq = P.synth.generate_if(sats, int((S.msToProcess + 10) * 1e-3 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateE1Bcode,
                             2 * S.codeFreqBasis, 8184, seed=1022, bit_periods=1, pilot_fn=P.codes.generateE1Ccode)
    return iq, GC_IQ, _channels(S, sats, 2.0)


def _ten23(data, pilot, ratio, prns, seed, pilot_phase=np.pi / 2):
    def build(P, S):
        sats = _sats(P, prns, seed, 18000, 50.0)
        use_pilot = pilot is not None and getattr(S, "pilotTRKflag", 0) == 1
        iq = P.synth.generate_if(sats, int((S.msToProcess + 4) * 1e-3 * S.samplingFreq), S.samplingFreq, S.IF, getattr(P.codes, data) if isinstance(data, str) else data(P),
                                 S.codeFreqBasis, 10230, seed=seed + 1, carrier_ratio=ratio, bit_periods=10,
                                 pilot_fn=(getattr(P.codes, pilot) if isinstance(pilot, str) else pilot(P)) if use_pilot else None, pilot_phase=pilot_phase)
        return iq, GC_IQ, _channels(S, sats, 2.0, code_freq=True)
    return build


def _single(code, rate_attr, code_len, ratio, prns, seed, layout=GC_IQ, glonass=False):
    def build(P, S):
        fs = S.samplingFreq
        rng = np.random.default_rng(seed)
        acc = np.zeros(2 * int((S.msToProcess + 4) * 1e-3 * fs))
        sats, ifs = [], []
        for p in prns:
            s = P.synth.SatSpec(prn=p, doppler=float(rng.uniform(-2e3, 2e3)), code_phase_samples=float(rng.uniform(0, fs * 1e-3)),
                                carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=50.0)
            f_if = S.IF + (p * S.freqSpacing if glonass else 0.0)
            sats.append(s)
            ifs.append(f_if)
            acc += P.synth.generate_if([s], acc.shape[0] // 2, fs, f_if, code(P), S.codeFreqBasis, code_len, seed=seed + 100 + p, carrier_ratio=ratio,
                                       noise=False)
        iq = np.clip(np.rint(acc + 20.0 * rng.standard_normal(acc.shape[0])), -127, 127).astype(np.int8)
        if layout == GC_QI:
            rec = np.empty_like(iq)
            rec[0::2], rec[1::2] = iq[1::2], iq[0::2]
        else:
            rec = iq
        ch = []
        for s, f_if in zip(sats, ifs):
            c = SimpleNamespace(acquiredFreq=f_if + s.doppler + 2.0, codePhase=int(np.ceil(s.code_phase_samples)) + 1, status="T")
            setattr(c, "K" if glonass else "PRN", s.prn)
            ch.append(c)
        pad = SimpleNamespace(acquiredFreq=0.0, codePhase=0, status="-")
        setattr(pad, "K" if glonass else "PRN", 0)
        ch.append(pad)
        return rec, layout, ch
    return build


def _b1c(P, S):
    sats = _sats(P, (8, 41), 1061, 180000, 47.0)
    iq = P.synth.generate_if(sats, int((S.msToProcess + 12) * 1e-3 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateDataBOC11,
                             2 * S.codeFreqBasis, 20460, seed=1062, bit_periods=1, pilot_fn=P.codes.generatePilotBOC11, pilot_phase=np.pi / 2)
    return iq, GC_IQ, _channels(S, sats, 1.0, code_freq=True)


_L2C_START = {5: 74, 17: 3}


def _l2c(P, S):
    def combined(prn):
        cm, cl = P.codes.generateCMcode(prn).astype(np.float64), P.codes.generateCLcode(prn).astype(np.float64)
        return np.roll(np.tile(cm, 75) + cl, -20460 * (_L2C_START[prn] - 1))
    sats = _sats(P, (5, 17), 1071, 160000, 45.0, 2e3)
    iq = P.synth.generate_if(sats, int((S.msToProcess + 24) * 1e-3 * S.samplingFreq), S.samplingFreq, S.IF, combined, 2 * S.codeFreqBasis,
                             20460 * 75, seed=1072, carrier_ratio=1200.0, bit_periods=1)
    ch = []
    for s in sats:
        ch.append(SimpleNamespace(PRN=s.prn, acquiredFreq=S.IF + s.doppler + 0.5, status="T", codePhase=int(np.ceil(s.code_phase_samples)),
                                  CLCodePhase=_L2C_START[s.prn]))
    ch.append(SimpleNamespace(PRN=0, acquiredFreq=0.0, codePhase=0, status="-", CLCodePhase=0))
    return iq, GC_IQ, ch


# ---- oracle runners ------------------------------------------------------------------------------------------------------
def _o_l1ca(O, rec, ch, S):
    return O.tracking_l1ca(rec, ch, S)


def _o_generic(**kw):
    def run(O, rec, ch, S):
        k = dict(kw)
        tables = k.pop("tables")
        spec = SimpleNamespace(tables=lambda prn: tables(O, prn), **k)
        och = [SimpleNamespace(**{**vars(c), "PRN": (1 if c.status != "-" else 0) if hasattr(c, "K") else c.PRN}) for c in ch]
        if not hasattr(S, "skipNumberOfBytes"):
            S = SimpleNamespace(**vars(S), skipNumberOfBytes=getattr(S, "skipNumberOfSamples", 0))
        return O.tracking_generic(rec, och, S, spec)
    return run


def _pad(O, c):
    return O.pad_code(c)


def _e5_tables(i_sig, q_sig):
    def t(O, prn):
        tiered = O.generate_e5_code(i_sig, prn, 2)
        data = np.concatenate([[tiered[10229]], tiered, [tiered[0]]])[:10232]
        return [data, O.pad_code(O.generate_e5_primary(q_sig, prn))]
    return t


def _o_b1c_wb(O, rec, ch, S):
    from cu_sdr_collection_amd import signals
    run = _o_generic(tables=lambda O_, prn: [O_.pad_code(O_.generate_b1c_code(prn, "data")), O_.pad_code(O_.generate_b1c_code(prn, "pilot11")),
                                             O_.pad_code(O_.generate_b1c_code(prn, "pilot61"))],
                     arm_mult=[1.0, 1.0, 6.0], r=2.0, pll="3state", coef_variant="b", pilot_combine=4, code_freq_from_channel=True,
                     dll_scale_spacing=True, pll_weight=(1.0, 3.0), dll_weight=signals._b1c_wb_dll_weight(S))
    return run(O, rec, ch, S)


TRACK_SCENES = [
    TrackScene("GPS_L1CA", "GPS/GPS_L1CA", "GPS_L1CA", "initSettings", dict(msToProcess=80, numberOfChannels=3), _l1ca, oracle=_o_l1ca),
    TrackScene("GPS_L1CA_int16_skip", "GPS/GPS_L1CA", "GPS_L1CA", "initSettings",
               dict(msToProcess=40, numberOfChannels=3, dataType="int16", skipNumberOfBytes=7200), _l1ca_int16_skip, oracle=_o_l1ca,
               notes="tracking.m:145-148,212-213: the int16 seek / ftell arithmetic with a non-zero skipNumberOfBytes"),
    TrackScene("GPS_L1CA_real", "GPS/GPS_L1CA", "GPS_L1CA", "initSettings", dict(msToProcess=40, numberOfChannels=3, fileType=1, IF=4.5e6),
               _l1ca_real, oracle=_o_l1ca, notes="tracking.m:126-130,232-236: fileType 1, real samples"),
    TrackScene("GAL_E1C", "GAL/GAL_E1C", "GAL_E1C", "initSettings_GAL_E1C", dict(msToProcess=100, numberOfChannels=3), _e1, pilot=True,
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_e1_code(prn, "B")), O.pad_code(O.generate_e1_code(prn, "C"))],
                                 r=2.0, pll="3state", coef_variant="a", pilot_combine=2, code_freq_from_channel=False)),
    TrackScene("GPS_L5C", "GPS/GPS_L5C", "GPS_L5C", "initSettings_GPS_L5C", dict(msToProcess=50, numberOfChannels=3, pilotTRKflag=1),
               _ten23("generateL5Icode", "generateL5Qcode", 1150.0, (6, 30), 1031), pilot=True,
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_l5_code(prn, "I")), O.pad_code(O.generate_l5_code(prn, "Q"))],
                                 r=1.0, pll="3state", coef_variant="a", pilot_combine=1, code_freq_from_channel=True)),
    TrackScene("GPS_L5C_data_only", "GPS/GPS_L5C", "GPS_L5C", "initSettings_GPS_L5C", dict(msToProcess=30, numberOfChannels=3, pilotTRKflag=0),
               _ten23("generateL5Icode", "generateL5Qcode", 1150.0, (6, 30), 1035),
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_l5_code(prn, "I"))],
                                 r=1.0, pll="3state", coef_variant="a", pilot_combine=0, code_freq_from_channel=True)),
    TrackScene("BDS_B2a", "BDS/B2a", "BDS_B2a", "initSettings_BDS_B2a", dict(msToProcess=50, numberOfChannels=3, pilotTRKflag=1, CNoInterval=25),
               _ten23("generateB2aDataCode", "generateB2aPilotCode", 1150.0, (20, 44), 1041), pilot=True,
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_b2a_code(prn, "data")), O.pad_code(O.generate_b2a_code(prn, "pilot"))],
                                 r=1.0, pll="3state", coef_variant="a", pilot_combine=1, code_freq_from_channel=True),
               extra_fields=("DataCNo", "PilotCNo", "B2a_CNo", "DataPLD", "PilotPLD")),
    TrackScene("BDS_B3I", "BDS/B3I", "BDS_B3I", "initSettings_BDS_B3I", dict(msToProcess=50, numberOfChannels=3),
               _ten23("generateB3Icode", None, 1240.0, (3, 37), 1043),
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_b3i_code(prn))], r=1.0, pll="3state", coef_variant="b",
                                 pilot_combine=0, code_freq_from_channel=True, int16_branch=True)),
    TrackScene("GAL_E5a", "GAL/GAL_E5a", "GAL_E5a", "initSettings_GAL_E5a", dict(msToProcess=50, numberOfChannels=3),
               _ten23(lambda P: (lambda prn: P.codes.generateE5aIcode(prn, 1)), lambda P: (lambda prn: P.codes.generateE5aQcode(prn, 1)), 1150.0, (2, 33), 1047),
               pilot=True, oracle=_o_generic(tables=_e5_tables("e5ai", "e5aq"), r=1.0, pll="3state", coef_variant="a", pilot_combine=1,
                                             code_freq_from_channel=True, int16_branch=True)),
    TrackScene("GAL_E5b", "GAL/GAL_E5b", "GAL_E5b", "initSettings_GAL_E5b", dict(msToProcess=50, numberOfChannels=3),
               _ten23(lambda P: (lambda prn: P.codes.generateE5bIcode(prn, 1)), lambda P: (lambda prn: P.codes.generateE5bQcode(prn, 1)), 1180.0, (11, 36), 1053),
               pilot=True, oracle=_o_generic(tables=_e5_tables("e5bi", "e5bq"), r=1.0, pll="3state", coef_variant="b", pilot_combine=1,
                                             code_freq_from_channel=True, int16_branch=True)),
    TrackScene("BDS_B1I", "BDS/B1I", "BDS_B1I", "initSettings_BDS_B1I", dict(msToProcess=60, numberOfChannels=3),
               _single(lambda P: P.codes.generateCAcode53, "codeFreqBasis", 2046, 1526.0, (7, 23), 1017),
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_b1i_code(prn))], r=1.0, pll="3state", coef_variant="a",
                                 pilot_combine=0, code_freq_from_channel=False)),
    TrackScene("GLO_GL1", "GLO/GLO_GL1", "GLO_GL1", "initSettings_GLO_GL1", dict(msToProcess=60, numberOfChannels=3),
               _single(lambda P: (lambda prn: P.codes.generateGLOcode()), "codeFreqBasis", 511, 3135.0, (-3, 5), 1019, layout=GC_QI, glonass=True),
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_glo_code())], r=1.0, pll="3state", coef_variant="a",
                                 pilot_combine=0, code_freq_from_channel=False, swap_iq=True)),
    TrackScene("GLO_GL2", "GLO/GLO_GL2", "GLO_GL2", "initSettings_GLO_GL2", dict(msToProcess=40, numberOfChannels=3),
               _single(lambda P: (lambda prn: P.codes.generateGLOcode()), "codeFreqBasis", 511, 2438.0, (0, -7), 1023, layout=GC_QI, glonass=True),
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_glo_code())], r=1.0, pll="3state", coef_variant="a",
                                 pilot_combine=0, code_freq_from_channel=False, swap_iq=True)),
    TrackScene("BDS_B1C_NB", "BDS/B1C", "BDS_B1C_NB", "initSettings_BDS_B1C", dict(msToProcess=80, numberOfChannels=3, CNoInterval=4), _b1c,
               fn="NB_tracking", pilot=True,
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_b1c_code(prn, "data")), O.pad_code(O.generate_b1c_code(prn, "pilot11"))],
                                 r=2.0, pll="3state", coef_variant="b", pilot_combine=3, code_freq_from_channel=True, dll_scale_spacing=True,
                                 pll_weight=(11.0, 29.0), dll_weight=(11.0, 29.0))),
    TrackScene("BDS_B1C_WB", "BDS/B1C", "BDS_B1C_WB", "initSettings_BDS_B1C", dict(msToProcess=50, numberOfChannels=3, CNoInterval=5, pilotTRKflag=2), _b1c,
               fn="WB_tracking", pilot=True, oracle=_o_b1c_wb,
               notes="the DLL weighting factor comes from CalcWeighingFactor.m's four integral() calls"),
    TrackScene("GPS_L2C", "GPS/GPS_L2C", "GPS_L2C", "initSettings_GPS_L2C", dict(msToProcess=100, numberOfChannels=3, pilotTRKflag=1), _l2c,
               pilot=True, oracle=lambda O, rec, ch, S: O.tracking_l2c(rec, ch, S)),
]


# a long closed loop of the reference's own tracking.m (1200 epochs: 30 PLL time constants, every 40-epoch C/N0 interval)
LONG_TRACK_SCENES = [
    TrackScene("GPS_L1CA_long", "GPS/GPS_L1CA", "GPS_L1CA", "initSettings", dict(msToProcess=1200, numberOfChannels=3), _l1ca, oracle=_o_l1ca,
               notes="tracking.m:133-368 over 1.2 s: long-run equivalence of the loops, not only of single epochs"),
    # BOC(1,1) tables, 4-ms blocks, pilot averaged in phase (GAL_E1C/include/tracking.m:236-348): 300 epochs = 1.2 s
    TrackScene("GAL_E1C_long", "GAL/GAL_E1C", "GAL_E1C", "initSettings_GAL_E1C", dict(msToProcess=1200, numberOfChannels=3), _e1, pilot=True,
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_e1_code(prn, "B")), O.pad_code(O.generate_e1_code(prn, "C"))],
                                 r=2.0, pll="3state", coef_variant="a", pilot_combine=2, code_freq_from_channel=False)),
    # the 3-state PLL of the other packages with a data + pilot pair, 800 epochs (GPS_L5C/include/tracking.m:255-382)
    TrackScene("GPS_L5C_long", "GPS/GPS_L5C", "GPS_L5C", "initSettings_GPS_L5C", dict(msToProcess=800, numberOfChannels=3, pilotTRKflag=1),
               _ten23("generateL5Icode", "generateL5Qcode", 1150.0, (6, 30), 1037), pilot=True,
               oracle=_o_generic(tables=lambda O, prn: [O.pad_code(O.generate_l5_code(prn, "I")), O.pad_code(O.generate_l5_code(prn, "Q"))],
                                 r=1.0, pll="3state", coef_variant="a", pilot_combine=1, code_freq_from_channel=True)),
]


def scene_inputs(P, sc: TrackScene):
    """(settings mirror with the overrides applied, record, layout, channels)."""
    from cu_sdr_collection_amd import settings as SET
    S = getattr(SET, sc.settings_fn)()
    for k, v in sc.overrides.items():
        setattr(S, k, v)
    rec, layout, ch = sc.build(P, S)
    return S, rec, layout, ch


# =====================================================================================================================
# acquisition scenes: acqResults = acquisition(longSignal, settings) of every package
# =====================================================================================================================
@dataclass
class AcqScene:
    name: str                     # fixture: tests/golden/ref_acq_<name>.npz
    pkg: str
    settings_fn: str
    overrides: dict
    build: object                 # (P, S) -> int8 I/Q record (longSignal = the whole record as data1 + 1i*data2)
    product: object               # (P, engine, S) -> acqResults of the HIP path (record already loaded, first sample 0)
    oracle: object = None         # (O, P, rec, S) -> acqResults of the oracle
    fields: tuple = ("carrFreq", "codePhase", "peakMetric")
    metric_rtol: float = 1e-9     # the winning cells are re-evaluated in float64 (csrc/acq_guard.h); *_resampled scenes search a float32 conditioned signal: 2e-7


def _acq_family_record(data, pilot, ratio, present, seed, ms, code_len=10230, rate_mult=1.0, bit_periods=1000, pilot_phase=np.pi / 2, cn0=50.0, dmax=4e3):
    def build(P, S):
        fs = S.samplingFreq
        sats = _sats(P, present, seed, fs * 1e-3, cn0, dmax)
        dfn = getattr(P.codes, data) if isinstance(data, str) else data(P)
        pfn = None if pilot is None else (getattr(P.codes, pilot) if isinstance(pilot, str) else pilot(P))
        return P.synth.generate_if(sats, int(ms * 1e-3 * fs), fs, S.IF, dfn, rate_mult * S.codeFreqBasis, code_len, seed=seed + 1,
                                   carrier_ratio=ratio, bit_periods=bit_periods, pilot_fn=pfn, pilot_phase=pilot_phase if pfn else 0.0)
    return build


def _acq_l1ca_record(P, S):
    rng = np.random.default_rng(5)
    sats = [P.synth.SatSpec(prn=p, doppler=float(rng.uniform(-5e3, 5e3)), code_phase_samples=float(rng.uniform(0, 18000)),
                            carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=cn0) for p, cn0 in ((7, 50.0), (14, 47.0), (22, 44.0), (31, 52.0))]
    return P.synth.generate_if(sats, 44 * 18000, S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=12)


def _acq_l2c_record(P, S):
    seg = 31

    def combined(prn):
        cm, cl = P.codes.generateCMcode(prn).astype(np.float64), P.codes.generateCLcode(prn).astype(np.float64)
        return np.roll(np.tile(cm, 75) + cl, -20460 * (seg - 1))
    sats = [P.synth.SatSpec(prn=5, doppler=212.0, code_phase_samples=70003.4, carrier_phase=1.0, cn0_dbhz=45.0)]
    return P.synth.generate_if(sats, int(0.25 * S.samplingFreq), S.samplingFreq, S.IF, combined, 2 * S.codeFreqBasis, 20460 * 75, seed=91,
                               carrier_ratio=1200.0, bit_periods=1)


def _acq_b1i_record(P, S):
    rng = np.random.default_rng(81)
    sats = [P.synth.SatSpec(prn=p, doppler=d, code_phase_samples=float(rng.uniform(0, 18000)), carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=c)
            for p, d, c in ((7, 2310.0, 50.0), (23, -3890.0, 47.0), (30, 40.0, 52.0))]
    return P.synth.generate_if(sats, int(0.012 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode53, S.codeFreqBasis, 2046, seed=82,
                               carrier_ratio=1526.0, bit_periods=20)


def _acq_b1c_record(P, S):
    sats = [P.synth.SatSpec(prn=8, doppler=-430.0, code_phase_samples=123456.7, carrier_phase=2.0, cn0_dbhz=47.0)]
    return P.synth.generate_if(sats, int(0.045 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateDataBOC11, 2 * S.codeFreqBasis, 20460, seed=93,
                               bit_periods=1, pilot_fn=P.codes.generatePilotBOC11, pilot_phase=np.pi / 2)


def _acq_e1_record(P, S):
    sats = [P.synth.SatSpec(prn=4, doppler=820.0, code_phase_samples=33333.3, carrier_phase=0.5, cn0_dbhz=50.0)]
    return P.synth.generate_if(sats, int(0.112 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateE1Bcode, 2 * S.codeFreqBasis, 8184, seed=111,
                               bit_periods=1, pilot_fn=P.codes.generateE1Ccode)


def _acq_glo_record(spacing_sign_ks, seed):
    def build(P, S):
        fs = S.samplingFreq
        rng = np.random.default_rng(seed)
        acc = np.zeros(2 * int(0.050 * fs))
        for K in spacing_sign_ks:
            s = P.synth.SatSpec(prn=1, doppler=float(rng.uniform(-3e3, 3e3)), code_phase_samples=float(rng.uniform(0, 12000)),
                                carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=50.0)
            acc += P.synth.generate_if([s], acc.shape[0] // 2, fs, S.IF - S.freqSpacing * K, lambda prn: P.codes.generateGLOcode(), S.codeFreqBasis, 511,
                                       seed=seed + 100 + K, carrier_ratio=3135.0, noise=False, bit_periods=10)
        return np.clip(np.rint(acc + 20.0 * rng.standard_normal(acc.shape[0])), -127, 127).astype(np.int8)
    return build


_NH20 = [1, 1, 1, 1, 1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, -1, -1, -1, 1]


def _b3i_combine(prn, per_code):
    x = per_code[0]
    if 1 <= prn <= 5 or 59 <= prn <= 63:
        return max(float(np.sum(np.abs(x.reshape(10, 2).sum(axis=1)))),
                   float(np.sum(np.abs(x[[0, 19]])) + np.sum(np.abs(x[1:19].reshape(9, 2).sum(axis=1)))))
    sec = np.array(_NH20, dtype=np.float64)
    best = abs(np.sum(x * sec))
    for k in range(1, 20):
        t = x * np.roll(sec, k)
        best = max(best, abs(np.sum(t[:k])) + abs(np.sum(t[k:])))
    return best


def _fam(**kw):
    def run(O, P, rec, S):
        k = {a: (b(O, P) if callable(b) and a in ("coarse_codes", "fine_codes", "secondary") else b) for a, b in kw.items()}
        return O.acquisition_family_a(rec, S, 0, **k)
    return run


ACQ_SCENES = [
    AcqScene("GPS_L1CA", "GPS/GPS_L1CA", "initSettings", dict(acqNonCohTime=4, acqSatelliteList=[3, 7, 11, 14, 19, 22, 28, 31]), _acq_l1ca_record,
             product=lambda P, eng, S: P.acquisition(eng, S, first_sample=0),
             oracle=lambda O, P, rec, S: O.acquisition_l1ca(rec[0::2].astype(np.float64) + 1j * rec[1::2].astype(np.float64), S)),
    # acquisition.m:46-111 (row A0): zero-phase FIR(700) band-pass + band-pass-sampling decimation, 18 Msps / IF 4.5 MHz ->
    # 6 113 500 Hz (12 228 = 2^2*3*1019 points per search), results mapped back (:264-276)
    AcqScene("GPS_L1CA_resampled", "GPS/GPS_L1CA", "initSettings", dict(acqNonCohTime=3, acqSatelliteList=[3, 7, 14, 22, 28, 31], IF=4.5e6, resamplingflag=1),
             _acq_l1ca_record, product=lambda P, eng, S: P.acquisition(eng, S, first_sample=0, n_long=44 * 18000),
             oracle=lambda O, P, rec, S: O.acquisition_l1ca(rec[0::2].astype(np.float64) + 1j * rec[1::2].astype(np.float64), S), metric_rtol=2e-7),
    AcqScene("GPS_L5C", "GPS/GPS_L5C", "initSettings_GPS_L5C", dict(acqNonCohTime=4, acqSearchBand=4500, acqSatelliteList=[3, 22, 9]),
             _acq_family_record("generateL5Icode", "generateL5Qcode", 1150.0, (3, 22), 101, 30),
             product=lambda P, eng, S: P.acq_family.acquisition_L5(eng, S, first_sample=0),
             oracle=_fam(coarse_codes=lambda O, P: (lambda prn: [O.generate_l5_code(prn, "I"), O.generate_l5_code(prn, "Q")]),
                         fine_codes=lambda O, P: (lambda prn: [O.generate_l5_code(prn, "Q")]), ncodes=20, fine_step=25.0, combine="circular",
                         secondary=lambda O, P: (lambda prn: _NH20))),
    # the same conditioning block in a data + pilot package (GPS_L5C acquisition.m:56-118, BW = 2*10.23 MHz + 0.5 MHz): 60 Msps /
    # IF 15 MHz -> 50 960 000 Hz, 101 920-point searches, NH20 fine stage on the conditioned signal
    AcqScene("GPS_L5C_resampled", "GPS/GPS_L5C", "initSettings_GPS_L5C",
             dict(acqNonCohTime=2, acqSearchBand=2000, acqSatelliteList=[3, 22], samplingFreq=60e6, IF=15e6, resamplingflag=1),
             _acq_family_record("generateL5Icode", "generateL5Qcode", 1150.0, (3, 22), 105, 30, dmax=1.5e3),
             product=lambda P, eng, S: P.acq_family.acquisition_L5(eng, S, first_sample=0), oracle=None, metric_rtol=2e-7),
    AcqScene("GAL_E5a", "GAL/GAL_E5a", "initSettings_GAL_E5a", dict(acqNonCohTime=3, acqSearchBand=4500, acqSatelliteList=[11, 30]),
             _acq_family_record(lambda P: (lambda prn: P.codes.generateE5aIcode(prn, 1)), lambda P: (lambda prn: P.codes.generateE5aQcode(prn, 1)), 1150.0, (11,), 101, 110),
             product=lambda P, eng, S: P.acq_family.acquisition_E5a(eng, S, first_sample=0),
             oracle=_fam(coarse_codes=lambda O, P: (lambda prn: [O.generate_e5_primary("e5ai", prn), O.generate_e5_primary("e5aq", prn)]),
                         fine_codes=lambda O, P: (lambda prn: [O.generate_e5_primary("e5aq", prn)]), ncodes=100, fine_step=5.0, combine="circular",
                         secondary=lambda O, P: (lambda prn: O.generate_e5_secondary100("e5aq", prn)), n_results=50)),
    AcqScene("BDS_B2a", "BDS/B2a", "initSettings_BDS_B2a", dict(acqNonCohTime=4, acqSearchBand=4500, acqSatelliteList=[21, 45, 33]),
             _acq_family_record("generateB2aDataCode", "generateB2aPilotCode", 1150.0, (21, 45), 101, 20),
             product=lambda P, eng, S: P.acq_family.acquisition_B2a(eng, S, first_sample=0),
             oracle=_fam(coarse_codes=lambda O, P: (lambda prn: [O.generate_b2a_code(prn, "data"), O.generate_b2a_code(prn, "pilot")]),
                         fine_codes=lambda O, P: (lambda prn: [O.generate_b2a_code(prn, "data"), O.generate_b2a_code(prn, "pilot")]),
                         ncodes=10, fine_step=25.0, combine="noncoh", n_results=45)),
    AcqScene("GAL_E5b", "GAL/GAL_E5b", "initSettings_GAL_E5b", dict(acqNonCohTime=3, acqSearchBand=4500, acqSatelliteList=[4, 19]),
             _acq_family_record(lambda P: (lambda prn: P.codes.generateE5bIcode(prn, 1)), lambda P: (lambda prn: P.codes.generateE5bQcode(prn, 1)), 1180.0, (4,), 101, 12),
             product=lambda P, eng, S: P.acq_family.acquisition_E5b(eng, S, first_sample=0),
             oracle=_fam(coarse_codes=lambda O, P: (lambda prn: [O.generate_e5_primary("e5bi", prn), O.generate_e5_primary("e5bq", prn)]),
                         fine_codes=None, ncodes=0, fine_step=0.0, combine=None, n_results=50)),
    # a package whose map back to the record's IF has no mirror branch (GAL_E5b acquisition.m:234-243), with the IF in the upper half of
    # the new Nyquist band: 100 Msps / IF 32 MHz -> 42 885 000 Hz (BW = 20.46 MHz, n = 2), 85 770-point searches, coarse bin = answer
    AcqScene("GAL_E5b_resampled", "GAL/GAL_E5b", "initSettings_GAL_E5b",
             dict(acqNonCohTime=2, acqSearchBand=1500, acqSatelliteList=[4, 19], samplingFreq=100e6, IF=32e6, resamplingflag=1),
             _acq_family_record(lambda P: (lambda prn: P.codes.generateE5bIcode(prn, 1)), lambda P: (lambda prn: P.codes.generateE5bQcode(prn, 1)), 1180.0, (4,), 107, 5, dmax=1.2e3),
             product=lambda P, eng, S: P.acq_family.acquisition_E5b(eng, S, first_sample=0), oracle=None, metric_rtol=2e-7),
    AcqScene("BDS_B3I", "BDS/B3I", "initSettings_BDS_B3I", dict(acqNonCohTime=3, acqSearchBand=4500, acqSatelliteList=[3, 30, 44]),
             _acq_family_record("generateB3Icode", None, 1240.0, (3, 30), 101, 30),
             product=lambda P, eng, S: P.acq_family.acquisition_B3I(eng, S, first_sample=0),
             oracle=_fam(coarse_codes=lambda O, P: (lambda prn: [O.generate_b3i_code(prn)]), fine_codes=lambda O, P: (lambda prn: [O.generate_b3i_code(prn)]),
                         ncodes=20, fine_step=25.0, combine=_b3i_combine, n_results=63, index_offset=0)),
    # BDS B3I spells the switch resamplingFlag (BDS/B3I/include/acquisition.m:47) and maps back without the mirror branch (:289-298)
    AcqScene("BDS_B3I_resampled", "BDS/B3I", "initSettings_BDS_B3I",
             dict(acqNonCohTime=2, acqSearchBand=2000, acqSatelliteList=[3, 30, 44], samplingFreq=60e6, IF=15e6, resamplingFlag=1),
             _acq_family_record("generateB3Icode", None, 1240.0, (3, 30), 109, 30, dmax=1.5e3),
             product=lambda P, eng, S: P.acq_family.acquisition_B3I(eng, S, first_sample=0), oracle=None, metric_rtol=2e-7),
    AcqScene("GAL_E1C", "GAL/GAL_E1C", "initSettings_GAL_E1C", dict(acqSearchBand=1500, acqSearchStep=150, acqNonCohTime=1, acqThreshold=10, acqSatelliteList=[4, 27]),
             _acq_e1_record, product=lambda P, eng, S: P.acq_family.acquisition_E1C(eng, S, first_sample=0),
             oracle=_fam(coarse_codes=lambda O, P: (lambda prn: [O.generate_e1_code(prn, "B"), O.generate_e1_code(prn, "C")]),
                         fine_codes=lambda O, P: (lambda prn: [O.generate_e1_code(prn, "C")]), ncodes=25, fine_step=10.0, combine="split",
                         secondary=lambda O, P: (lambda prn: P.acq_family.E1C_SECONDARY), n_results=50, boc=True, index_offset=0)),
    # Galileo E1 with the conditioning block (BW = 20.552 MHz, GAL_E1C acquisition.m:57): 60 Msps / IF 15 MHz -> 50 552 000 Hz; the BOC(1,1)
    # tables are sampled at the new rate (makeE1BTable.m with the overwritten settings.samplingFreq), 404 416-point searches
    AcqScene("GAL_E1C_resampled", "GAL/GAL_E1C", "initSettings_GAL_E1C",
             dict(acqSearchBand=900, acqSearchStep=150, acqNonCohTime=1, acqThreshold=10, acqSatelliteList=[4, 27], samplingFreq=60e6, IF=15e6, resamplingflag=1),
             _acq_e1_record, product=lambda P, eng, S: P.acq_family.acquisition_E1C(eng, S, first_sample=0), oracle=None, metric_rtol=2e-7),
    AcqScene("GLO_GL1", "GLO/GLO_GL1", "initSettings_GLO_GL1", dict(acqNonCohTime=4, acqSatelliteList=[-3, 0, 5]), _acq_glo_record((-3, 5), 121),
             product=lambda P, eng, S: P.acq_family.acquisition_GLO(eng, S, first_sample=0), oracle=lambda O, P, rec, S: O.acquisition_glo(rec, S, 0)),
    AcqScene("GLO_GL2", "GLO/GLO_GL2", "initSettings_GLO_GL2", dict(acqNonCohTime=3, acqSatelliteList=[-7, 2, 6]), _acq_glo_record((-7, 6), 131),
             product=lambda P, eng, S: P.acq_family.acquisition_GLO(eng, S, first_sample=0), oracle=lambda O, P, rec, S: O.acquisition_glo(rec, S, 0)),
    # GLONASS with the input-conditioning block (GLO_GL1 acquisition.m:50-119, BW = 9 MHz): 30 Msps / IF 7 MHz -> 23 Msps; the code phase is
    # mapped back, the carrier frequency lands in the field the reference spells carrFreqcarrFreq (:284) and carrFreq stays in the new band
    AcqScene("GLO_GL1_resampled", "GLO/GLO_GL1", "initSettings_GLO_GL1",
             dict(acqNonCohTime=2, acqSearchBand=2000, acqSatelliteList=[-3, 0, 5], samplingFreq=30e6, IF=7e6, resamplingflag=1), _acq_glo_record((-3, 5), 141),
             product=lambda P, eng, S: P.acq_family.acquisition_GLO(eng, S, first_sample=0), oracle=None, metric_rtol=2e-7,
             fields=("carrFreq", "codePhase", "peakMetric", "carrFreqcarrFreq")),
    AcqScene("BDS_B1I", "BDS/B1I", "initSettings_BDS_B1I", dict(acqSatelliteList=[7, 12, 23, 30]), _acq_b1i_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_B1I(eng, S, first_sample=0), oracle=lambda O, P, rec, S: O.acquisition_b1i(rec, S, 0)),
    AcqScene("GPS_L2C", "GPS/GPS_L2C", "initSettings_GPS_L2C", dict(pilotTRKflag=1, acqSearchBand=1, acqSatelliteList=[5, 9]), _acq_l2c_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_L2C(eng, S, first_sample=0), oracle=lambda O, P, rec, S: O.acquisition_l2c(rec, S, 0),
             fields=("carrFreq", "codePhase", "peakMetric", "CLCodePhase")),
    AcqScene("BDS_B1C", "BDS/B1C", "initSettings_BDS_B1C", dict(acqSearchBand=1000, acqSatelliteList=[8, 20]), _acq_b1c_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_B1C(eng, S, first_sample=0), oracle=lambda O, P, rec, S: O.acquisition_b1c(rec, S, 0)),
    # BDS B1C with its conditioning block (BW = 9 MHz, band edges widened by 0.002): 30 Msps / IF 6.5 MHz -> 22 Msps, where the
    # 20-ms transform is 440 000 = 2^6*5^4*11 points - no size for the radix plan: the bins run carrier by carrier on the padded transform
    AcqScene("BDS_B1C_resampled", "BDS/B1C", "initSettings_BDS_B1C",
             dict(acqSearchBand=500, acqSatelliteList=[8, 20], samplingFreq=30e6, IF=6.5e6, resamplingflag=1), _acq_b1c_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_B1C(eng, S, first_sample=0), oracle=None, metric_rtol=2e-7),
]


# ---- constructed near-ties and near-threshold metrics (the float64 guard of csrc/acq_guard.h) ----------------------------------------
# The HIP searches transform in float32; the reference picks `[peakSize, codePhase] = max(max(results))` and gates on `peakMetric >
# acqThreshold` in float64 (GPS_L1CA/include/acquisition.m:196-206).  These records put two cells of `results`, or the metric and the
# threshold, a few 1e-7 apart (relative): far above anything float64 transforms could confuse (~1e-13), below what float32 ones resolve.
_TIE_PRN, _TIE_TAU0 = 7, 4321


def _sampled_ca(P, S, prn):
    return np.asarray(P.codes.makeCaTable(prn, S), dtype=np.float64)          # one code period at the sampling rate (makeCaTable.m)


def _acq_tie_cols_record(P, S):
    """IF = 0, no noise, Q = 0: PRN 7 twice, one sample apart, at exactly 0 Hz (bin 15 of 29) - columns tau0 and tau0 + 1 of that bin hold
    the same integer sum per hop; ONE sample raised by 1 where the replica's neighbours differ makes the LATER column larger by 2 in
    one hop (3.6e-7 of the sum over the hops): first-occurrence on float32-equal values would answer tau0."""
    t = _sampled_ca(P, S, _TIE_PRN)
    spc, n = t.shape[0], 44 * t.shape[0]
    idx = np.arange(n)
    xi = 40.0 * (t[(idx - _TIE_TAU0) % spc] + t[(idx - _TIE_TAU0 - 1) % spc])
    k = next(k for k in range(100, spc - 1) if t[k] == 1.0 and t[k + 1] == -1.0)    # column tau0 + 1 sees t[k] = +1 there, column tau0 t[k + 1] = -1
    xi[_TIE_TAU0 + 1 + k] += 1.0
    rec = np.zeros(2 * n, dtype=np.int8)
    rec[0::2] = xi.astype(np.int8)
    return rec


def _acq_tie_bins_record(P, S):
    """IF = 0, no noise: PRN 7 on a REAL carrier cos(2 pi 500 t), Q = 0 - the bins at +500 and -500 Hz (14 and 16 of 29) are complex
    conjugates of each other, equal in magnitude whatever the int8 rounding did; ONE sample of Q set to 1 tilts them by ~5e-7
    towards the LATER bin."""
    t = _sampled_ca(P, S, _TIE_PRN)
    spc, n = t.shape[0], 44 * t.shape[0]
    idx = np.arange(n)
    xi = np.rint(100.0 * t[(idx - _TIE_TAU0) % spc] * np.cos(2.0 * np.pi * 500.0 * idx / S.samplingFreq))
    rec = np.zeros(2 * n, dtype=np.int8)
    rec[0::2] = xi.astype(np.int8)
    rec[2 * (_TIE_TAU0 + 777) + 1] = _TIE_BINS_Q
    return rec


def _acq_b1i_tie_record(P, S):
    """BDS B1I (circshift search: 2 carriers x 2 blocks of 4 ms x 41 bins, a 2-ms replica), IF = 0, no noise, Q = 0: PRN 7 on a REAL
    carrier cos(2 pi 1000 t).  The rows at +1000 and -1000 Hz are conjugates, the two 4-ms blocks are identical, and the four code periods
    of a block give four equal columns: sixteen cells of `results` tie exactly.  Three single samples raised by 1 make ONE of them the
    largest by a few 1e-7 (relative) - the row at -1000 Hz (scanned AFTER the one at +1000 Hz: the reference's `>` keeps the first of
    equal values, BDS/B1I/include/acquisition.m:98-119), first block, first period - and leave the others 1e-7 .. 1e-6 apart."""
    fs, tau0 = S.samplingFreq, 5000
    t = np.asarray(P.codes.generateCAcode53(_TIE_PRN), dtype=np.float64)
    spc = 18000
    idx = np.ceil(np.arange(1, spc + 1) / fs * S.codeFreqBasis).astype(np.int64)
    idx[-1] = 2046
    t = t[idx - 1]
    n = int(0.012 * fs)
    k = np.arange(n)
    xi = np.rint(100.0 * t[(k - tau0) % spc] * np.cos(2.0 * np.pi * 1000.0 * k / fs))
    rec = np.zeros(2 * n, dtype=np.int8)
    rec[0::2] = xi.astype(np.int8)
    # I + sign(replica): one sample in the first period after tau0 (seen by the windows at tau0 and tau0 + 3 periods), one in the second
    # (windows at tau0 and tau0 + 1 period): the column at tau0 gains 2, its periodic copies 1, 1 and 0
    for off in (1234, spc + 4321):
        rec[2 * (tau0 + off)] += np.int8(1 if xi[tau0 + off] >= 0 else -1)
    rec[2 * (tau0 + 777) + 1] = 1          # Q: tilts the conjugate rows towards -1000 Hz
    return rec


_TIE_BINS_Q = 1          # the sign that makes bin 16 (-500 Hz) the larger one (checked by test_ref_vectors against the fixture's carrFreq)
_THR_METRIC_B1I_PRN23 = 3.1654201090103307  # max_peak / second of PRN 23 on the BDS_B1I scene's record, likewise
_THR_METRIC_PRN22 = 4.690999243563728       # peakMetric(22) of the GPS_L1CA scene's record: oracle and reference fixture agree to the last bit

GUARD_ACQ_SCENES = [
    AcqScene("GPS_L1CA_tie_cols", "GPS/GPS_L1CA", "initSettings", dict(IF=0.0, acqNonCohTime=4, acqSatelliteList=[_TIE_PRN]), _acq_tie_cols_record,
             product=lambda P, eng, S: P.acquisition(eng, S, first_sample=0),
             oracle=lambda O, P, rec, S: O.acquisition_l1ca(rec[0::2].astype(np.float64) + 1j * rec[1::2].astype(np.float64), S), metric_rtol=1e-9),
    AcqScene("GPS_L1CA_tie_bins", "GPS/GPS_L1CA", "initSettings", dict(IF=0.0, acqNonCohTime=4, acqSatelliteList=[_TIE_PRN]), _acq_tie_bins_record,
             product=lambda P, eng, S: P.acquisition(eng, S, first_sample=0),
             oracle=lambda O, P, rec, S: O.acquisition_l1ca(rec[0::2].astype(np.float64) + 1j * rec[1::2].astype(np.float64), S), metric_rtol=1e-9),
    # the GPS_L1CA scene's record with the threshold 2.5e-7 (relative) below / above PRN 22's metric: detected / not detected
    AcqScene("GPS_L1CA_thr_below", "GPS/GPS_L1CA", "initSettings", dict(acqNonCohTime=4, acqSatelliteList=[22, 3], acqThreshold=_THR_METRIC_PRN22 * (1.0 - 2.5e-7)),
             _acq_l1ca_record, product=lambda P, eng, S: P.acquisition(eng, S, first_sample=0),
             oracle=lambda O, P, rec, S: O.acquisition_l1ca(rec[0::2].astype(np.float64) + 1j * rec[1::2].astype(np.float64), S), metric_rtol=1e-9),
    AcqScene("GPS_L1CA_thr_above", "GPS/GPS_L1CA", "initSettings", dict(acqNonCohTime=4, acqSatelliteList=[22, 3], acqThreshold=_THR_METRIC_PRN22 * (1.0 + 2.5e-7)),
             _acq_l1ca_record, product=lambda P, eng, S: P.acquisition(eng, S, first_sample=0),
             oracle=lambda O, P, rec, S: O.acquisition_l1ca(rec[0::2].astype(np.float64) + 1j * rec[1::2].astype(np.float64), S), metric_rtol=1e-9),
    # the circshift family: sixteen near-tied cells over four rows (which row the sequential rule keeps decides carrFreq), a second peak on a
    # plateau of equal side lobes
    AcqScene("BDS_B1I_tie_rows", "BDS/B1I", "initSettings_BDS_B1I", dict(IF=0.0, acqSatelliteList=[_TIE_PRN]), _acq_b1i_tie_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_B1I(eng, S, first_sample=0), oracle=lambda O, P, rec, S: O.acquisition_b1i(rec, S, 0), metric_rtol=1e-9),
    # the BDS_B1I scene's record with the threshold 2.5e-7 (relative) below / above max_peak / second of PRN 23 (:157-166)
    AcqScene("BDS_B1I_thr_below", "BDS/B1I", "initSettings_BDS_B1I", dict(acqSatelliteList=[23, 12], acqThreshold=_THR_METRIC_B1I_PRN23 * (1.0 - 2.5e-7)), _acq_b1i_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_B1I(eng, S, first_sample=0), oracle=lambda O, P, rec, S: O.acquisition_b1i(rec, S, 0), metric_rtol=1e-9),
    AcqScene("BDS_B1I_thr_above", "BDS/B1I", "initSettings_BDS_B1I", dict(acqSatelliteList=[23, 12], acqThreshold=_THR_METRIC_B1I_PRN23 * (1.0 + 2.5e-7)), _acq_b1i_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_B1I(eng, S, first_sample=0), oracle=lambda O, P, rec, S: O.acquisition_b1i(rec, S, 0), metric_rtol=1e-9),
]


# ---- the reference's DEFAULT searches: settings = initSettings() unmodified (only the record is ours) -----------------------------
# GPS L1 C/A 32 PRNs x 29 bins x 20 hops (GPS_L1CA/initSettings.m:80-89), GPS L5 25 hops (GPS_L5C/initSettings.m:74-83), Galileo E5b
# 168 bins of 60 Hz x 15 hops x 36 PRNs (GAL_E5b/initSettings.m:83-92), Galileo E1 94 bins x 144 000 points (GAL_E1C/initSettings.m:81-89),
# BDS B3I 63 PRNs (BDS/B3I/initSettings.m:76-84), BDS B1C 62 PRNs x 201 bins x 360 000 points (BDS/B1C/initSettings.m:67,92-99), GPS L2C
# 401 bins x 320 000 points (GPS_L2C/initSettings.m:84-94), BDS B1I 53 PRNs (BDS/B1I/initSettings.m:83-94), GLONASS K = -7..6 (:89-97).
# The interpreter's FFTs are NumPy: a fixture takes minutes to make (make_ref_vectors.py acq_default), once; the CPU suite does not
# re-run them through the oracle (oracle=None), the GPU suite and bench.py compare the HIP path with them (array_equal).
def _acq_l1ca_default_record(P, S):
    rng = np.random.default_rng(2405)
    spec = ((2, 51.0), (5, 47.0), (9, 42.0), (12, 45.0), (17, 40.0), (21, 49.0), (25, 43.5), (29, 38.5), (32, 46.0))
    sats = [P.synth.SatSpec(prn=p, doppler=float(rng.uniform(-6.5e3, 6.5e3)), code_phase_samples=float(rng.uniform(0, 18000)),
                            carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=cn0) for p, cn0 in spec]
    return P.synth.generate_if(sats, 44 * 18000, S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=2406)


def _acq_glo_default_record(ks, seed):
    def build(P, S):
        fs = S.samplingFreq
        rng = np.random.default_rng(seed)
        acc = np.zeros(2 * int(0.046 * fs))
        for K, cn0 in ks:
            s = P.synth.SatSpec(prn=1, doppler=float(rng.uniform(-4.5e3, 4.5e3)), code_phase_samples=float(rng.uniform(0, 12000)),
                                carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=cn0)
            acc += P.synth.generate_if([s], acc.shape[0] // 2, fs, S.IF - S.freqSpacing * K, lambda prn: P.codes.generateGLOcode(), S.codeFreqBasis, 511,
                                       seed=seed + 100 + K, carrier_ratio=3135.0, noise=False, bit_periods=10)
        return np.clip(np.rint(acc + 20.0 * rng.standard_normal(acc.shape[0])), -127, 127).astype(np.int8)
    return build


def _acq_b1i_default_record(P, S):
    rng = np.random.default_rng(2481)
    sats = [P.synth.SatSpec(prn=p, doppler=d, code_phase_samples=float(rng.uniform(0, 18000)), carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=c)
            for p, d, c in ((8, 4310.0, 49.0), (19, -7890.0, 46.0), (33, 640.0, 51.0), (46, 9120.0, 47.0), (58, -2225.0, 45.0))]
    return P.synth.generate_if(sats, int(0.012 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode53, S.codeFreqBasis, 2046, seed=2482,
                               carrier_ratio=1526.0, bit_periods=20)


def _acq_l2c_default_record(P, S):
    segs = {5: 31, 17: 7}

    def combined(prn):
        cm, cl = P.codes.generateCMcode(prn).astype(np.float64), P.codes.generateCLcode(prn).astype(np.float64)
        return np.roll(np.tile(cm, 75) + cl, -20460 * (segs[prn] - 1))
    sats = [P.synth.SatSpec(prn=5, doppler=3212.0, code_phase_samples=70003.4, carrier_phase=1.0, cn0_dbhz=45.0),
            P.synth.SatSpec(prn=17, doppler=-4431.0, code_phase_samples=141234.2, carrier_phase=4.0, cn0_dbhz=44.0)]
    return P.synth.generate_if(sats, int(0.25 * S.samplingFreq), S.samplingFreq, S.IF, combined, 2 * S.codeFreqBasis, 20460 * 75, seed=2491,
                               carrier_ratio=1200.0, bit_periods=1)


def _acq_b1c_default_record(P, S):
    sats = [P.synth.SatSpec(prn=8, doppler=-3430.0, code_phase_samples=123456.7, carrier_phase=2.0, cn0_dbhz=47.0),
            P.synth.SatSpec(prn=27, doppler=1275.0, code_phase_samples=40404.1, carrier_phase=0.3, cn0_dbhz=44.0),
            P.synth.SatSpec(prn=61, doppler=4760.0, code_phase_samples=171717.9, carrier_phase=5.1, cn0_dbhz=45.0)]
    return P.synth.generate_if(sats, int(0.045 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateDataBOC11, 2 * S.codeFreqBasis, 20460, seed=2493,
                               bit_periods=1, pilot_fn=P.codes.generatePilotBOC11, pilot_phase=np.pi / 2)


def _acq_e1_default_record(P, S):
    sats = [P.synth.SatSpec(prn=4, doppler=5820.0, code_phase_samples=33333.3, carrier_phase=0.5, cn0_dbhz=50.0),
            P.synth.SatSpec(prn=19, doppler=-2410.0, code_phase_samples=60606.6, carrier_phase=2.5, cn0_dbhz=46.0),
            P.synth.SatSpec(prn=33, doppler=-6905.0, code_phase_samples=7007.7, carrier_phase=4.5, cn0_dbhz=48.0)]
    return P.synth.generate_if(sats, int(0.112 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateE1Bcode, 2 * S.codeFreqBasis, 8184, seed=2411,
                               bit_periods=1, pilot_fn=P.codes.generateE1Ccode)


DEFAULT_ACQ_SCENES = [
    AcqScene("GPS_L1CA_default", "GPS/GPS_L1CA", "initSettings", {}, _acq_l1ca_default_record,
             product=lambda P, eng, S: P.acquisition(eng, S, first_sample=0)),
    AcqScene("GPS_L5C_default", "GPS/GPS_L5C", "initSettings_GPS_L5C", {},
             _acq_family_record("generateL5Icode", "generateL5Qcode", 1150.0, (3, 10, 22, 27), 2501, 42, dmax=4.8e3, cn0=47.0),
             product=lambda P, eng, S: P.acq_family.acquisition_L5(eng, S, first_sample=0)),
    AcqScene("GAL_E5a_default", "GAL/GAL_E5a", "initSettings_GAL_E5a", {},
             _acq_family_record(lambda P: (lambda prn: P.codes.generateE5aIcode(prn, 1)), lambda P: (lambda prn: P.codes.generateE5aQcode(prn, 1)), 1150.0, (11, 24, 36), 2511, 104,
                                dmax=4.8e3, cn0=47.0),
             product=lambda P, eng, S: P.acq_family.acquisition_E5a(eng, S, first_sample=0)),
    AcqScene("GAL_E5b_default", "GAL/GAL_E5b", "initSettings_GAL_E5b", {},
             _acq_family_record(lambda P: (lambda prn: P.codes.generateE5bIcode(prn, 1)), lambda P: (lambda prn: P.codes.generateE5bQcode(prn, 1)), 1180.0, (4, 19, 31), 2521, 24,
                                dmax=4.8e3, cn0=47.0),
             product=lambda P, eng, S: P.acq_family.acquisition_E5b(eng, S, first_sample=0)),
    AcqScene("BDS_B2a_default", "BDS/B2a", "initSettings_BDS_B2a", {},
             _acq_family_record("generateB2aDataCode", "generateB2aPilotCode", 1150.0, (21, 34, 45, 60), 2531, 20, dmax=4.8e3, cn0=47.0),
             product=lambda P, eng, S: P.acq_family.acquisition_B2a(eng, S, first_sample=0)),
    AcqScene("BDS_B3I_default", "BDS/B3I", "initSettings_BDS_B3I", {},
             _acq_family_record("generateB3Icode", None, 1240.0, (3, 30, 44, 61), 2541, 30, dmax=4.8e3, cn0=47.0),
             product=lambda P, eng, S: P.acq_family.acquisition_B3I(eng, S, first_sample=0)),
    AcqScene("GAL_E1C_default", "GAL/GAL_E1C", "initSettings_GAL_E1C", {}, _acq_e1_default_record,
             product=lambda P, eng, S: P.acq_family.acquisition_E1C(eng, S, first_sample=0)),
    AcqScene("GLO_GL1_default", "GLO/GLO_GL1", "initSettings_GLO_GL1", {}, _acq_glo_default_record(((-7, 49.0), (-2, 45.0), (3, 47.0), (6, 50.0)), 2551),
             product=lambda P, eng, S: P.acq_family.acquisition_GLO(eng, S, first_sample=0)),
    AcqScene("GLO_GL2_default", "GLO/GLO_GL2", "initSettings_GLO_GL2", {}, _acq_glo_default_record(((-5, 48.0), (0, 46.0), (4, 50.0)), 2561),
             product=lambda P, eng, S: P.acq_family.acquisition_GLO(eng, S, first_sample=0)),
    AcqScene("BDS_B1I_default", "BDS/B1I", "initSettings_BDS_B1I", {}, _acq_b1i_default_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_B1I(eng, S, first_sample=0)),
    AcqScene("GPS_L2C_default", "GPS/GPS_L2C", "initSettings_GPS_L2C", {}, _acq_l2c_default_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_L2C(eng, S, first_sample=0)),   # pilotTRKflag = 0 by default: no CL search (:125)
    AcqScene("BDS_B1C_default", "BDS/B1C", "initSettings_BDS_B1C", {}, _acq_b1c_default_record,
             product=lambda P, eng, S: P.acq_shift.acquisition_B1C(eng, S, first_sample=0)),
]


def acq_inputs(P, sc: AcqScene):
    from cu_sdr_collection_amd import settings as SET
    S = getattr(SET, sc.settings_fn)()
    for k, v in sc.overrides.items():
        setattr(S, k, v)
    return S, sc.build(P, S)


# =========================================================================================================================
# Bit / frame synchronisation: prompt streams for the sync block of every package's NAVdecoding.m
# =========================================================================================================================
@dataclass
class NavSyncScene:
    name: str                     # fixture: tests/golden/ref_navsync_<name>.npz
    pkg: str                      # reference package directory
    package: str                  # cu_sdr_collection_amd.nav_sync.SYNC key
    ranges: tuple                 # 1-based inclusive line ranges of <pkg>/include/NAVdecoding.m that are executed (the sync block)
    stream_var: str               # name of the prompt stream in the file (function argument)
    prn: int
    n: int                        # values in the stream
    ms_to_process: int
    frame: int                    # samples between two sync patterns
    first: int                    # 0-based position of the first pattern
    sps: int                      # samples per data symbol
    spread: tuple = ()            # what a data symbol is multiplied by, sample by sample (secondary / NH code); () = ones
    sigma: float = 0.4            # noise sigma in units of the symbol amplitude
    seed: int = 0
    verified: str = ""            # "gps": TLM / HOW words with valid parity after the preamble; "bds": valid BCH(15,11) second half
    loop_var: str = ""            # workspace variable the executed loop leaves the verified start in ('' = the range ends before the loop)


def _gps_words(rng, parity_check, polarity, prev):
    """One 300-bit sub-frame as +-1 values whose every word passes navPartyChk: word 1 starts with the TLM preamble (x polarity),
    the 24 information bits of a word are random, its six parity bits are the one combination (of 64) that checks."""
    out = []
    for w in range(10):
        t = rng.integers(0, 2, 24) * 2 - 1
        if w == 0:
            t[:8] = polarity * np.array([1, -1, -1, -1, 1, -1, 1, 1])
        for c in range(64):
            par = np.array([1 if (c >> k) & 1 else -1 for k in range(6)])
            word = np.concatenate([prev, t, par])
            if parity_check(word) != 0:
                break
        else:
            raise AssertionError("no parity combination checks")
        out.append(word[2:])
        prev = word[30:32]
    return np.concatenate(out), prev


def navsync_stream(sc: NavSyncScene, pattern: np.ndarray, parity_check=None) -> np.ndarray:
    """The scene's prompt in-phase stream (float64[n]): random data symbols (each spread by sc.spread), the sync pattern at
    sc.first + k * sc.frame with alternating-ish polarity, Gaussian noise, a handful of exact zeros.  `pattern`: the package's sync
    pattern at the stream's rate; `parity_check`: navPartyChk (needed for verified == "gps")."""
    rng = np.random.default_rng(sc.seed)
    spread = np.asarray(sc.spread if sc.spread else np.ones(sc.sps), dtype=np.float64)
    assert spread.shape[0] == sc.sps
    nsym = sc.n // sc.sps + 2
    x = np.kron(rng.integers(0, 2, nsym) * 2.0 - 1.0, spread)
    shift = sc.first % sc.sps                       # data symbol boundaries line up with the frames
    x = np.concatenate([x[sc.sps - shift:], x[:sc.sps - shift]])[:sc.n] if shift else x[:sc.n]
    prev = np.array([1, 1])
    k = 0
    while sc.first + k * sc.frame < sc.n:
        s0 = sc.first + k * sc.frame
        pol = 1 if (k % 3) != 1 else -1
        if sc.verified == "gps":
            words, prev = _gps_words(rng, parity_check, pol, prev)
            seg = np.kron(words.astype(np.float64), spread)
        else:
            seg = pol * pattern.astype(np.float64)
            if sc.verified == "bds":                # 11 preamble bits, 4 free bits, then a BCH(15,11) code word (first bit = highest power)
                msg = rng.integers(0, 2, 11)
                r = 0
                for b in list(msg) + [0, 0, 0, 0]:
                    r = (r << 1) | int(b)
                    if r & 0x10:
                        r ^= 0x13
                cw = np.concatenate([msg, [(r >> 3) & 1, (r >> 2) & 1, (r >> 1) & 1, r & 1]])
                if k % 4 == 2:
                    cw[5] ^= 1                      # every fourth frame carries a corrupted word: found by the correlation, refused by the check
                bits = np.concatenate([rng.integers(0, 2, 4), cw]) * 2.0 - 1.0
                # the reference sums a bit's samples WITHOUT wiping the NH code off (BDS/B1I NAVdecoding.m:146-147): bit x spread
                seg = np.concatenate([seg, pol * np.kron(bits, spread)])
        m = min(seg.shape[0], sc.n - s0)
        x[s0:s0 + m] = seg[:m]
        k += 1
    x = 1000.0 * (x + sc.sigma * rng.standard_normal(sc.n))
    x[rng.integers(0, sc.n, 12)] = 0.0              # exact zeros: `bits <= 0` vs Galileo E1's `I_P < 0`
    for j in range(0, sc.n - sc.first, sc.frame):   # ... some of them inside a sync pattern, where the rule decides a detection
        x[sc.first + j + (3 * (j // sc.frame)) % max(1, pattern.shape[0])] = 0.0
    return x


_NH20 = (-1, -1, -1, -1, -1, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, 1, 1, 1, -1)
NAVSYNC_SCENES = [
    NavSyncScene("GPS_L1CA", "GPS/GPS_L1CA", "GPS_L1CA", ((66, 145),), "I_P_InputBits", 7, 20000, 20000, 6000, 777, 20, sigma=0.42, seed=8101,
                 verified="gps", loop_var="subFrameStart"),
    NavSyncScene("GAL_E1C", "GAL/GAL_E1C", "GAL_E1C", ((59, 59), (79, 97)), "I_P", 11, 9500, 38000, 250, 123, 1, sigma=0.55, seed=8102),
    NavSyncScene("GAL_E5a", "GAL/GAL_E5a", "GAL_E5a", ((54, 54), (69, 71), (84, 108)), "I_P", 12, 32000, 32000, 10000, 1501, 20,
                 spread=tuple(1 - 2 * np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 1, 1, 0, 1, 0, 0, 1])), sigma=0.22, seed=8103),
    NavSyncScene("GAL_E5b", "GAL/GAL_E5b", "GAL_E5b", ((59, 59), (80, 85), (88, 107)), "I_P", 19, 36000, 36000, 1000, 333, 4, spread=(-1, -1, -1, 1),
                 sigma=0.3, seed=8104),
    NavSyncScene("BDS_B1I_MEO", "BDS/B1I", "BDS_B1I", ((68, 170),), "I_P_InputBits", 12, 20000, 20000, 6000, 1404, 20, spread=tuple(-v for v in _NH20),
                 sigma=0.35, seed=8105, verified="bds", loop_var="subFrameStart"),
    NavSyncScene("BDS_B1I_GEO", "BDS/B1I", "BDS_B1I", ((68, 170),), "I_P_InputBits", 3, 6000, 6000, 600, 1111, 2, sigma=0.5, seed=8106,
                 verified="bds", loop_var="subFrameStart"),
    NavSyncScene("BDS_B3I_MEO", "BDS/B3I", "BDS_B3I", ((69, 164),), "I_P_InputBits", 30, 44000, 44000, 6000, 2222, 20, spread=tuple(-v for v in _NH20),
                 sigma=0.35, seed=8107, verified="bds", loop_var="subFrameStart"),
    NavSyncScene("BDS_B3I_GEO", "BDS/B3I", "BDS_B3I", ((69, 164),), "I_P_InputBits", 60, 34000, 34000, 600, 1051, 2, sigma=0.5, seed=8108,
                 verified="bds", loop_var="subFrameStart"),
    NavSyncScene("GLO_GL1", "GLO/GLO_GL1", "GLO_GL1", ((66, 97),), "I_P_InputBits", 0, 12000, 12000, 2000, 455, 10, sigma=0.5, seed=8109),
]
