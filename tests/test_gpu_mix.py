"""BASELINE config 5 on one GPU: channels of several reference packages tracked AT THE SAME TIME on shared records
(gc_track_multi / receiver.tracking_multi), each package checked against the oracle's restatement of ITS tracking.m.

The reference runs its packages one after the other (one `settings`, one tracking() per package: tracking.m:133); the
channels are independent, so running them side by side must give exactly what the separate calls give."""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import gnss_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _all_six_pilot_sums(monkeypatch):
    """These tests compare the pilot arm's early / late sums with the oracle too; the reference's trackResults keeps the prompt
    pair only for most packages (receiver.DEFAULT_PILOT_FIELDS; the reference's exact field sets: tests/test_gpu_ref_vectors.py)."""
    from cu_sdr_collection_amd import receiver
    monkeypatch.setattr(receiver, "DEFAULT_PILOT_FIELDS", "all")

_SUMS = ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")
_PILOT = tuple("Pilot_" + f for f in _SUMS)


def _l1_band_scene():
    """One 18-Msps L1-band record (GPS/GPS_L1CA/initSettings.m:60-69 front end) carrying GPS L1 C/A, Galileo E1-B/C
    (BOC(1,1), data + pilot) and BDS B1C (data BOC(1,1) + pilot BOC(1,1) in quadrature)."""
    import cu_sdr_collection_amd as P
    from cu_sdr_collection_amd.settings import initSettings_BDS_B1C, initSettings_GAL_E1C
    from cu_sdr_collection_amd.synth import SatSpec, SignalGroup, generate_if_mix
    fs = 18e6
    rng = np.random.default_rng(505)

    def sats(prns, period_samples, cn0):
        return [SatSpec(prn=p, doppler=float(rng.uniform(-3e3, 3e3)), code_phase_samples=float(rng.uniform(0, period_samples)),
                        carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=cn0) for p in prns]

    l1, e1, b1c = sats((3, 17, 28), 18000, 46.0), sats((4, 19, 31), 72000, 47.0), sats((8, 41), 180000, 47.0)
    groups = [SignalGroup(l1, P.codes.generateCAcode, 1.023e6, 1023),
              SignalGroup(e1, P.codes.generateE1Bcode, 2.046e6, 8184, bit_periods=1, pilot_fn=P.codes.generateE1Ccode),
              SignalGroup(b1c, P.codes.generateDataBOC11, 2.046e6, 20460, bit_periods=1, pilot_fn=P.codes.generatePilotBOC11,
                          pilot_phase=np.pi / 2)]
    n = int(0.096 * fs)
    iq = generate_if_mix(groups, n, fs, 20e3, seed=506)
    S1 = P.initSettings()
    S1.msToProcess, S1.numberOfChannels = 80, 4
    S2 = initSettings_GAL_E1C()
    S2.msToProcess, S2.numberOfChannels = 80, 3          # 20 epochs of 4 ms
    S3 = initSettings_BDS_B1C()
    S3.msToProcess, S3.numberOfChannels = 80, 2          # 8 epochs of 10 ms

    def chans(S, sv, nch, code_freq=False):
        out = []
        for s in sv:
            f = S.IF + s.doppler + 2.0
            c = SimpleNamespace(PRN=s.prn, acquiredFreq=f, status="T", codePhase=int(np.ceil(s.code_phase_samples)) + 1)
            if code_freq:
                c.codeFreq = S.codeFreqBasis + (f - S.IF) / S.carrFreqBasis * S.codeFreqBasis
            out.append(c)
        while len(out) < nch:
            out.append(SimpleNamespace(PRN=0, acquiredFreq=0.0, codePhase=0, status="-"))
        return out

    return iq, (S1, chans(S1, l1, 4)), (S2, chans(S2, e1, 3)), (S3, chans(S3, b1c, 2, code_freq=True))


def _check_generic(tr, ref, nact, scale, pilot=True):
    for k in range(nact):
        assert tr[k].status == "T" and ref[k].status == "T"
        assert np.array_equal(tr[k].absoluteSample, ref[k].absoluteSample)
        for f in _SUMS + (_PILOT if pilot else ()):
            assert np.max(np.abs(getattr(tr[k], f) - getattr(ref[k], f))) < 1e-5 * scale, (k, f)
        assert np.max(np.abs(tr[k].carrFreq - ref[k].carrFreq)) < 1e-3
        assert np.max(np.abs(tr[k].remCodePhase - ref[k].remCodePhase)) < 1e-7


@pytest.mark.parametrize("device_loop", [False, True])
def test_three_packages_on_one_record_run_concurrently_and_match_their_oracles(engine, device_loop):
    import cu_sdr_collection_amd as P
    iq, (S1, ch1), (S2, ch2), (S3, ch3) = _l1_band_scene()
    engine.load_if(iq, fs=18e6)
    with P.Engine(0) as e2, P.Engine(0) as e3:
        e2.share_if(engine)
        e3.share_if(engine)
        assert e2.if_buffer() == engine.if_buffer()          # one record in HBM, three readers
        (tr1, _), (tr2, _), (tr3, _) = P.receiver.tracking_multi(
            [(engine, ch1, S1, "GPS_L1CA"), (e2, ch2, S2, "GAL_E1C"), (e3, ch3, S3, "BDS_B1C_NB")], device_loop=device_loop)
        # the same three calls one after the other, as the reference would run them
        seq1, _ = P.tracking(engine, ch1, S1, device_loop=device_loop)
        seq2, _ = P.tracking(e2, ch2, S2, signal="GAL_E1C", device_loop=device_loop)
        # (B1C's 2 x 20 462-entry tables are f16 in LDS: no device-loop instantiation, gc_track_multi fell back to gc_track)
        seq3, _ = P.tracking(e3, ch3, S3, signal="BDS_B1C_NB")
    for a, b in ((tr1, seq1), (tr2, seq2), (tr3, seq3)):
        for x, y in zip(a, b):
            assert x.status == y.status and x.PRN == y.PRN
            assert np.array_equal(x.absoluteSample, y.absoluteSample)
            # same kernels, no atomics; only the team size (partial-sum order) may differ when the device is shared
            for f in ("carrFreq", "codeFreq", "I_P", "Q_P", "I_E", "Q_L", "remCodePhase", "remCarrPhase"):
                assert np.allclose(getattr(x, f), getattr(y, f), rtol=1e-6, atol=1e-6 * np.abs(y.I_P).max() + 1e-12), f

    # GPS L1 C/A vs GPS/GPS_L1CA/include/tracking.m (C twin of the oracle)
    ref1, done, aborted = CO.track_l1ca(iq, ch1, S1)
    assert not aborted
    for k in range(3):
        assert tr1[k].status == "T"
        assert np.array_equal(tr1[k].absoluteSample, ref1["absoluteSample"][k])
        assert np.max(np.abs(tr1[k].carrFreq - ref1["carrFreq"][k])) < 1e-3
        scale = 2.0 * 18000 * 28.0
        for f in _SUMS:
            assert np.max(np.abs(getattr(tr1[k], f) - ref1[f][k])) < 1e-5 * scale, f
        assert np.mean(np.abs(tr1[k].I_P[40:])) > 3 * np.mean(np.abs(tr1[k].Q_P[40:]))
    assert tr1[3].status == "-" and not tr1[3].I_P.any()

    # Galileo E1 B+C vs GAL/GAL_E1C/include/tracking.m
    spec2 = SimpleNamespace(tables=lambda prn: [O.pad_code(O.generate_e1_code(prn, "B")), O.pad_code(O.generate_e1_code(prn, "C"))],
                            r=2.0, pll="3state", coef_variant="a", pilot_combine=2, code_freq_from_channel=False)
    _check_generic(tr2, O.tracking_generic(iq, ch2, S2, spec2), 3, 2.0 * 72000 * 28.0)

    # BDS B1C narrow-band vs BDS/B1C/include/NB_tracking.m
    spec3 = SimpleNamespace(tables=lambda prn: [O.pad_code(O.generate_b1c_code(prn, "data")), O.pad_code(O.generate_b1c_code(prn, "pilot11"))],
                            r=2.0, pll="3state", coef_variant="b", pilot_combine=3, code_freq_from_channel=True,
                            dll_scale_spacing=True, pll_weight=(11.0, 29.0), dll_weight=(11.0, 29.0))
    _check_generic(tr3, O.tracking_generic(iq, ch3, S3, spec3), 2, 2.0 * 180000 * 28.0)


def test_two_records_two_packages_and_a_short_read(engine, l1ca_scene, capsys):
    """Jobs may read DIFFERENT records (an L1-band and an L5-band file on one GPU), and a short read of one package
    (tracking.m:241-245) leaves the others untouched."""
    import cu_sdr_collection_amd as P
    from cu_sdr_collection_amd.settings import initSettings_GPS_L5C
    S, sats, iq = l1ca_scene
    S.msToProcess, S.numberOfChannels = 400, 2            # the record holds 300 ms: short read
    ch = [SimpleNamespace(PRN=s.prn, acquiredFreq=S.IF + s.doppler + 4.0, codePhase=int(np.ceil(s.code_phase_samples)) + 1, status="T")
          for s in sats[:2]]
    S5 = initSettings_GPS_L5C()
    S5.msToProcess, S5.numberOfChannels, S5.pilotTRKflag = 40, 2, 1
    rng = np.random.default_rng(515)
    sv = [P.synth.SatSpec(prn=p, doppler=float(rng.uniform(-2e3, 2e3)), code_phase_samples=float(rng.uniform(0, 18000)),
                          carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=47.0) for p in (6, 25)]
    iq5 = P.synth.generate_if(sv, int(0.045 * 18e6), 18e6, S5.IF, P.codes.generateL5Icode, 10.23e6, 10230, seed=516,
                              carrier_ratio=115.0, bit_periods=10, pilot_fn=P.codes.generateL5Qcode, pilot_phase=np.pi / 2)
    ch5 = []
    for s in sv:
        f = S5.IF + s.doppler + 1.0
        ch5.append(SimpleNamespace(PRN=s.prn, acquiredFreq=f, status="T", codePhase=int(np.ceil(s.code_phase_samples)) + 1,
                                   codeFreq=S5.codeFreqBasis + (f - S5.IF) / S5.carrFreqBasis * S5.codeFreqBasis))
    engine.load_if(iq, fs=S.samplingFreq)
    with P.Engine(0) as e5:
        e5.load_if(iq5, fs=18e6)
        (tr, _), (tr5, _) = P.receiver.tracking_multi([(engine, ch, S, "GPS_L1CA"), (e5, ch5, S5, "GPS_L5C")])
    assert "Not able to read the specified number of samples" in capsys.readouterr().out
    ref, done, aborted = CO.track_l1ca(iq, ch, S)
    assert aborted and tr[0].status == "-" and tr[1].status == "-"
    n0 = int(done[0])
    assert np.array_equal(tr[0].absoluteSample[:n0], ref["absoluteSample"][0][:n0]) and not tr[0].I_P[n0:].any()
    spec = SimpleNamespace(tables=lambda prn: [O.pad_code(O.generate_l5_code(prn, "I")), O.pad_code(O.generate_l5_code(prn, "Q"))],
                           r=1.0, pll="3state", coef_variant="a", pilot_combine=1, code_freq_from_channel=True)
    _check_generic(tr5, O.tracking_generic(iq5, ch5, S5, spec), 2, 2.0 * 18000 * 28.0)


def test_track_multi_argument_checks(engine, l1ca_scene):
    import cu_sdr_collection_amd as P
    from cu_sdr_collection_amd import _lib as L
    from cu_sdr_collection_amd.receiver import track_params
    S, sats, iq = l1ca_scene
    S.msToProcess = 5
    engine.load_if(iq, fs=S.samplingFreq)
    engine.set_channel(0, [P.codes.padded_table(P.codes.generateCAcode(sats[0].prn))])
    init = [L.gc_channel_init(channel=0, prn=sats[0].prn, acquired_freq=S.IF + sats[0].doppler, code_freq=S.codeFreqBasis,
                              code_phase=int(np.ceil(sats[0].code_phase_samples)) + 1)]
    p = track_params(S)
    with pytest.raises(L.GnssCorrError) as e:
        P.Engine.track_multi([(engine, p, init), (engine, p, init)])       # one context, two jobs
    assert e.value.status == L.GC_E_INVALID
    with P.Engine(0) as other:
        with pytest.raises(L.GnssCorrError) as e:
            other.share_if(other)
        assert e.value.status == L.GC_E_INVALID
        with pytest.raises(L.GnssCorrError) as e:
            P.Engine.track_multi([(engine, p, init), (other, p, init)])   # `other` has no record and no channel
        assert e.value.status == L.GC_E_STATE
    # a spacing that leaves the [c(end) c c(1)] padding is refused up front (gc_track validated nothing)
    wide = track_params(S)
    wide.el_spacing = 1.0
    with pytest.raises(L.GnssCorrError) as e:
        engine.track(wide, init)
    assert e.value.status == L.GC_E_INVALID
    longer = track_params(S)
    longer.code_length = 2046.0                                              # table holds 1025 entries
    with pytest.raises(L.GnssCorrError) as e:
        engine.track(longer, init)
    assert e.value.status == L.GC_E_INVALID
    with pytest.raises(L.GnssCorrError) as e:
        engine.track(longer, init, device_loop=True)
    assert e.value.status == L.GC_E_INVALID


def test_job_set_whose_persistent_grids_do_not_fit_together_shrinks_its_teams_or_launches_per_epoch(engine):
    """gc_track_multi launches the jobs' persistent kernels plainly (cooperative launches of different streams do not
    overlap), so nothing but the library's own admission check keeps partly resident grids from spinning on each other for good.
    Twelve GPS L1 C/A channels make 12 x 32 one-wave members = 48 workgroups per XCD; the Galileo E1 job's members need a CU
    each (82 KB of tables, eight waves of ~170 VGPRs): 51 workgroups for an XCD's 32 CUs.  Whichever kernel comes second is
    refused at that size: the GPS job halves its teams until the set fits (12 x 8 members = 12 + 3 workgroups per XCD), the
    Galileo job - which no team size brings under an XCD already holding 48 workgroups - runs with a launch per epoch.  Either
    way the same records as the packages tracked one after the other."""
    import cu_sdr_collection_amd as P
    from cu_sdr_collection_amd.settings import initSettings_GAL_E1C
    from cu_sdr_collection_amd.synth import SatSpec, SignalGroup, generate_if_mix_gpu
    fs = 18e6
    rng = np.random.default_rng(707)

    def sats(prns, period_samples, cn0):
        return [SatSpec(prn=int(p), doppler=float(rng.uniform(-3e3, 3e3)), code_phase_samples=float(rng.uniform(0, period_samples)),
                        carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=cn0) for p in prns]

    l1 = sats(rng.choice(np.arange(1, 33), size=12, replace=False), 18000, 47.0)
    e1 = sats((5, 11, 24), 72000, 47.0)
    groups = [SignalGroup(l1, P.codes.generateCAcode, 1.023e6, 1023),
              SignalGroup(e1, P.codes.generateE1Bcode, 2.046e6, 8184, bit_periods=1, pilot_fn=P.codes.generateE1Ccode)]
    # four seconds of record, 3.9 s tracked: the first job's persistent kernel is resident for tens of milliseconds (6 us per epoch),
    # so the second job - started at the same time on its own host thread - meets it on the device whichever of the two comes first
    # (with 40 epochs the first kernel had often left before the second asked: both admitted, nothing tested).  The sequential runs
    # come first: they load both persistent kernels' code, which would otherwise be part of the race.
    generate_if_mix_gpu(engine, groups, int(4.0 * fs), fs, 20e3, seed=708)
    engine.set_sampling_freq(fs)
    S1 = P.initSettings()
    S1.msToProcess, S1.numberOfChannels = 3900, 12
    S2 = initSettings_GAL_E1C()
    S2.msToProcess, S2.numberOfChannels = 3900, 3
    ch1 = [SimpleNamespace(PRN=s.prn, acquiredFreq=S1.IF + s.doppler + 2.0, status="T", codePhase=int(np.ceil(s.code_phase_samples)) + 1) for s in l1]
    ch2 = [SimpleNamespace(PRN=s.prn, acquiredFreq=S2.IF + s.doppler + 2.0, status="T", codePhase=int(np.ceil(s.code_phase_samples)) + 1) for s in e1]
    with P.Engine(0) as e2:
        e2.share_if(engine)
        seq1, _ = P.tracking(engine, ch1, S1)
        seq2, _ = P.tracking(e2, ch2, S2, signal="GAL_E1C")
        assert (engine.last_track_mode(), e2.last_track_mode()) == (1, 1)   # alone on the device both take their persistent kernel
        (tr1, _), (tr2, _) = P.receiver.tracking_multi([(engine, ch1, S1, "GPS_L1CA"), (e2, ch2, S2, "GAL_E1C")])
        modes = (engine.last_track_mode(), e2.last_track_mode())
        # GPS first: its 384 workgroups stay, Galileo is refused at every team size -> (1, 0); Galileo first: GPS shrinks -> (1, 1)
        assert modes in ((1, 0), (1, 1)), modes
    # the first 40 epochs of every channel (later on a launch-per-epoch loop and a persistent one may cut a block one sample apart:
    # their float32 partial sums are grouped differently)
    for a, b in ((tr1, seq1), (tr2, seq2)):
        for x, y in zip(a, b):
            assert x.status == y.status == "T" and np.array_equal(x.absoluteSample[:40], y.absoluteSample[:40])
            for f in ("carrFreq", "codeFreq", "I_P", "Q_P", "I_E", "Q_L", "remCodePhase", "remCarrPhase"):
                assert np.allclose(getattr(x, f)[:40], getattr(y, f)[:40], rtol=1e-6, atol=1e-6 * np.abs(y.I_P).max() + 1e-12), f
