"""The drop-in boundary exercised end to end on the GPU, without MATLAB:

  * matlab/gnsscorr_mex.c, the MEX gateway a maintainer builds with `mex`, is compiled here against a test-only mex.h
    (tests/mexstub) and its mexFunction is driven command by command;
  * the MATLAB drop-ins themselves - matlab/packages/<package>/tracking.m -> matlab/gnsscorr_tracking.m, called with the
    REFERENCE'S signature [trackResults, channel] = tracking(fid, channel, settings) - are executed by the mini-MATLAB
    interpreter of oracle/mlab with `gnsscorr_mex` bound to that compiled gateway, and must reproduce what the reference's own
    tracking.m computed on the same record (tests/golden/ref_track_*.npz), field for field."""
import os
import sys

import numpy as np
import pytest

import ref_scenes as RS

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "mexstub"))
pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gateway():
    import harness
    g = harness.Gateway()
    yield g
    g.lib.stub_run_atexit()          # mexAtExit: destroys whatever contexts are left


def test_mex_gateway_commands(gateway, l1ca_scene):
    import cu_sdr_collection_amd as P
    import harness
    from oracle import c_oracle as CO
    S, sats, iq = l1ca_scene
    h = gateway.call("create", 0)
    assert gateway.lib.stub_lock_count() >= 1                            # mexLock while a context lives
    name, cus = gateway.call("device_info", h, nargout=2)
    assert "MI355" in name or "gfx950" in name or cus.item() >= 200
    gateway.call("load_if", h, iq, 2, S.samplingFreq, nargout=0)
    back = gateway.call("read_if", h, 100, 50, "int8", 2)
    assert np.array_equal(back.reshape(-1), iq[200:300])
    n_ep = 40
    for k, s in enumerate(sats[:2]):
        code = P.codes.generateCAcode(s.prn).astype(np.float64)
        gateway.call("set_channel", h, k, [np.concatenate([code[-1:], code, code[:1]])], 1, nargout=0)      # double tables: cast in the gateway
    params = dict(samplingFreq=S.samplingFreq, codeFreqBasis=S.codeFreqBasis, codeLength=S.codeLength, dllCorrelatorSpacing=S.dllCorrelatorSpacing,
                  intTime=S.intTime, dllNoiseBandwidth=S.dllNoiseBandwidth, dllDampingRatio=S.dllDampingRatio, pllNoiseBandwidth=S.pllNoiseBandwidth,
                  pllDampingRatio=S.pllDampingRatio, pllKind=0, skipSamples=0, numEpochs=n_ep)
    chan = np.array([[k, s.prn, S.IF + s.doppler + 4.0, S.codeFreqBasis, int(np.ceil(s.code_phase_samples)) + 1, 0] for k, s in enumerate(sats[:2])], dtype=np.float64).T
    trk, epochs, status = gateway.call("track", h, params, chan, nargout=3)
    assert trk.shape == (n_ep, 21 * 2) and list(epochs.reshape(-1)) == [n_ep, n_ep] and status.item() == 0
    from types import SimpleNamespace
    S.msToProcess = n_ep
    ch = [SimpleNamespace(PRN=s.prn, acquiredFreq=S.IF + s.doppler + 4.0, codePhase=int(np.ceil(s.code_phase_samples)) + 1, status="T") for s in sats[:2]]
    ref, done, aborted = CO.track_l1ca(iq, ch, S)
    for k in range(2):
        assert np.array_equal(trk[:, 21 * k + 0], ref["absoluteSample"][k])
        assert np.max(np.abs(trk[:, 21 * k + 5] - ref["I_P"][k])) < 1e-5 * 2 * 18000 * 28
    # coarse acquisition through the gateway
    acq = dict(samplingFreq=S.samplingFreq, codeFreqBasis=S.codeFreqBasis, codeLength=S.codeLength, IF=S.IF, acqSearchBand=7000, acqSearchStep=500,
               acqNonCohTime=4, firstSample=0)
    prns = [sats[0].prn, 1 if sats[0].prn != 1 else 2]
    tabs = np.stack([P.codes.makeCaTable(p, S) for p in prns]).astype(np.int8).T          # spc x nprn, column per PRN
    res = gateway.call("acquire_coarse", h, acq, np.asfortranarray(tabs))
    assert res.shape == (5, 2) and res[3, 0] > 3.5
    assert abs((res[1, 0] - 1 - sats[0].code_phase_samples + 9000) % 18000 - 9000) < 3
    # errors come back as MATLAB errors (mexErrMsgIdAndTxt), with the library's message
    with pytest.raises(harness.MexError) as e:
        gateway.call("track", 7, params, chan, nargout=3)
    assert "invalid context handle" in str(e.value)
    bad = dict(params, dllCorrelatorSpacing=1.5)
    with pytest.raises(harness.MexError) as e:
        gateway.call("track", h, bad, chan, nargout=3)
    assert "gc_track" in str(e.value) and "table" in str(e.value)
    with pytest.raises(harness.MexError):
        gateway.call("no_such_command", h)
    # outputs are sized from the context, not from the caller's numbers - a class or a length that disagrees with
    # what the library will write is an error in MATLAB's terms, never a write past an mxArray
    assert gateway.call("read_if", h, 100, 50).dtype == np.int8                # class and values per sample default to the record's
    with pytest.raises(harness.MexError) as e:
        gateway.call("read_if", h, 100, 50, "int16", 2)
    assert "int8" in str(e.value)
    with pytest.raises(harness.MexError):
        gateway.call("read_if", h, 100, 50, "int8", 1)
    gateway.call("load_if", h, iq[:40000].astype(np.int16), 2, S.samplingFreq, nargout=0)
    back16 = gateway.call("read_if", h, 10, 20)
    assert back16.dtype == np.int16 and np.array_equal(back16.reshape(-1), iq[20:60].astype(np.int16))
    with pytest.raises(harness.MexError) as e:
        gateway.call("read_if", h, 10, 20, "int8", 2)                           # the r2 default: would have written 80 bytes into 40
    assert "int16" in str(e.value)
    with pytest.raises(harness.MexError) as e:
        gateway.call("acq_shift_row", h, 0, 123)                                # nothing prepared
    assert "gc_acq_shift_prepare" in str(e.value)
    # sync_xcorr: classes checked, the Galileo E1 zero rule as a flag
    x = np.array([3.0, 0.0, -2.0, 5.0, 0.0, 1.0])
    pat = np.array([1, -1, 1], dtype=np.int8)
    assert list(gateway.call("sync_xcorr", h, x, pat).reshape(-1)) == [1 + 1 - 1, -1 + 1 + 1, -1 - 1 - 1, 1 + 1 + 1, -1 - 1, 1]
    assert list(gateway.call("sync_xcorr", h, x, pat, 1).reshape(-1)) == [1 - 1 - 1, 1 + 1 + 1, -1 - 1 + 1, 1 - 1 + 1, 1 - 1, 1]
    with pytest.raises(harness.MexError):
        gateway.call("sync_xcorr", h, x, pat.astype(np.float64))
    gateway.call("destroy", h, nargout=0)
    with pytest.raises(harness.MexError):
        gateway.call("device_info", h, nargout=2)


_WRAPPER_DIR = {"GPS_L1CA": "GPS_L1CA", "GPS_L5C": "GPS_L5C", "GPS_L2C": "GPS_L2C", "GAL_E1C": "GAL_E1C", "GAL_E5a": "GAL_E5a", "GAL_E5b": "GAL_E5b",
                "BDS_B1I": "BDS_B1I", "BDS_B2a": "BDS_B2a", "BDS_B3I": "BDS_B3I", "GLO_GL1": "GLO_GL1", "GLO_GL2": "GLO_GL2", "BDS_B1C_NB": "BDS_B1C",
                "BDS_B1C_WB": "BDS_B1C"}


@pytest.mark.parametrize("sc", RS.TRACK_SCENES, ids=[s.name for s in RS.TRACK_SCENES])
def test_matlab_drop_in_reproduces_the_references_tracking_m(gateway, sc, tmp_path):
    import bridge
    import cu_sdr_collection_amd as P
    from oracle import mlab
    z = np.load(os.path.join(GOLD, f"ref_track_{sc.name}.npz"))
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    assert RS.crc(rec) == int(z["record_crc32"][0])
    path = str(tmp_path / "record.bin")
    rec.tofile(path)
    I = bridge.install(bridge.interpreter_for(_WRAPPER_DIR[sc.signal]), gateway, P, sc.signal)
    fid = mlab.register_file(I, rec.tobytes(), path)           # fopen(fid) answers with the file name, as in MATLAB
    from types import SimpleNamespace
    mch = mlab.to_matlab([SimpleNamespace(**{k: (v if isinstance(v, str) else float(v)) for k, v in vars(c).items()}) for c in ch])
    try:
        tr, chout = I.call(sc.fn, fid, mch, mlab.to_matlab(S), nargout=2)      # the reference's signature, the reference's file name
    finally:
        I.call("gnsscorr_context", "", "clear")
    tr = mlab.from_matlab(tr)
    ref_fields = [k[2:] for k in z.files if k.startswith("f_") and k != "f_PRN"]
    comp = 1 if layout == RS.GC_REAL else 2
    full = comp * S.samplingFreq * S.intTime * 28.0 * (5.0 if rec.dtype == np.int16 else 1.0)
    assert [t.status for t in tr] == [str(s) for s in z["status"]]
    for k, t in enumerate(tr):
        have = {f for f in vars(t) if isinstance(getattr(t, f), (np.ndarray, float)) and f != "PRN"}
        assert have == set(ref_fields), (sc.name, sorted(have ^ set(ref_fields)))
        for f in ref_fields:
            want, got = z["f_" + f][k], np.atleast_1d(np.asarray(getattr(t, f), dtype=np.float64))
            assert got.shape == want.shape, (sc.name, f)
            assert np.array_equal(np.isinf(got), np.isinf(want)), (sc.name, k, f)
            m = np.isfinite(want)
            if not m.any():
                continue
            d = float(np.max(np.abs(got[m] - want[m])))
            if f == "absoluteSample":
                assert d < 1e-6, (sc.name, k, d)
            elif f[-3:] in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L"):
                assert d < 1e-5 * full, (sc.name, k, f, d / full)
            elif f in ("carrFreq", "codeFreq", "DataCNo", "PilotCNo", "B2a_CNo", "B1C_CNo"):
                assert d < 1e-3, (sc.name, k, f, d)
            elif f == "remCarrPhase":
                dd = np.abs(got[m] - want[m])
                assert np.max(np.minimum(dd, np.abs(dd - 2 * np.pi))) < 1e-5
            else:
                assert d < 2e-5 * max(1.0, float(np.max(np.abs(want[m])))), (sc.name, k, f, d)
        if "cno_VSMValue" in z.files and t.status == "T":
            assert np.allclose(np.atleast_1d(t.CNo.VSMValue), z["cno_VSMValue"][k], atol=1e-3)
            assert np.array_equal(np.atleast_1d(t.CNo.VSMIndex), z["cno_VSMIndex"][k])
        if bool(z["PRN_set"][k]):
            assert float(t.PRN) == float(z["PRN"][k])


_ACQ_WRAPPED = ("GPS_L1CA", "GPS_L5C", "GAL_E5a", "BDS_B2a", "GAL_E5b", "BDS_B3I", "GAL_E1C", "GLO_GL1", "GLO_GL2", "BDS_B1I", "GPS_L2C", "BDS_B1C", "GPS_L1CA_resampled", "GPS_L5C_resampled", "GAL_E5b_resampled", "GLO_GL1_resampled", "BDS_B3I_resampled", "GAL_E1C_resampled", "BDS_B1C_resampled")


@pytest.mark.parametrize("sc", [s for s in RS.ACQ_SCENES if s.name in _ACQ_WRAPPED], ids=[s.name for s in RS.ACQ_SCENES if s.name in _ACQ_WRAPPED])
def test_matlab_drop_in_reproduces_the_references_acquisition_m(gateway, sc):
    """acqResults = acquisition(longSignal, settings) - the reference's signature, longSignal a complex double row - through
    matlab/packages/<package>/acquisition.m and the compiled gateway, against the reference's own acquisition.m."""
    import bridge
    import cu_sdr_collection_amd as P
    from oracle import mlab
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    assert RS.crc(rec) == int(z["record_crc32"][0])
    x = rec.astype(np.float64)
    long_signal = (x[0::2] + 1j * x[1::2]).reshape(1, -1)
    pkg = sc.name.replace("_resampled", "")
    signal = {"BDS_B1C": "BDS_B1C_NB"}.get(pkg, pkg)
    I = bridge.install(bridge.interpreter_for(pkg), gateway, P, signal)
    try:
        acq = mlab.from_matlab(I.call("acquisition", long_signal, mlab.to_matlab(S)))
    finally:
        I.call("gnsscorr_context", "", "clear")
    for f in sc.fields:
        want, got = z["f_" + f], np.asarray(getattr(acq, f), dtype=np.float64).reshape(-1)
        assert got.shape == want.shape, (sc.name, f, got.shape, want.shape)
        if f == "peakMetric":
            assert np.max(np.abs(got - want)) <= sc.metric_rtol * np.max(np.abs(want)), sc.name
        else:
            assert np.array_equal(got, want), (sc.name, f, got[got != want], want[got != want])


def test_matlab_drop_in_tracks_a_record_larger_than_the_window(gateway, tmp_path):
    """settings.gnsscorrWindowSamples: the drop-in tracks the file window by window (gnsscorr_mex('track_file') -> gc_track_file)
    and returns exactly what it returns with the whole record resident."""
    import bridge
    import cu_sdr_collection_amd as P
    from oracle import mlab
    from types import SimpleNamespace
    sc = next(s for s in RS.TRACK_SCENES if s.name == "GPS_L1CA")
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    path = str(tmp_path / "record.bin")
    rec.tofile(path)
    mch = mlab.to_matlab([SimpleNamespace(**{k: (v if isinstance(v, str) else float(v)) for k, v in vars(c).items()}) for c in ch])
    out = []
    for window in (0, int(S.samplingFreq * S.intTime * 14.5)):
        I = bridge.install(bridge.interpreter_for("GPS_L1CA"), gateway, P, sc.signal)
        fid = mlab.register_file(I, rec.tobytes(), path)
        if window:
            S.gnsscorrWindowSamples = float(window)
        Sm = mlab.to_matlab(S)
        try:
            tr, _ = I.call("tracking", fid, mch, Sm, nargout=2)
        finally:
            I.call("gnsscorr_context", "", "clear")
        out.append(mlab.from_matlab(tr))
    for a, b in zip(*out):
        assert a.status == b.status
        for f in vars(a):
            if isinstance(getattr(a, f), np.ndarray):
                assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_matlab_drop_in_keeps_one_context_per_record_not_per_file_name(gateway, tmp_path):
    """ADVICE r2 (gnsscorr_tracking.m:69): the cached context is keyed by what decides the bytes in HBM and reloads a record it does
    not hold.  A windowed call first, then a resident one on the same file (was: 'no IF record'); the resident call again after a
    windowed one; and the same file NAME rewritten as an int16 record (was: the stale int8 record, silently)."""
    import bridge
    import cu_sdr_collection_amd as P
    from oracle import mlab
    from types import SimpleNamespace
    sc = next(s for s in RS.TRACK_SCENES if s.name == "GPS_L1CA")
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    path = str(tmp_path / "record.bin")
    rec.tofile(path)
    mch = mlab.to_matlab([SimpleNamespace(**{k: (v if isinstance(v, str) else float(v)) for k, v in vars(c).items()}) for c in ch])
    I = bridge.install(bridge.interpreter_for("GPS_L1CA"), gateway, P, sc.signal)
    fid = mlab.register_file(I, rec.tobytes(), path)
    window = float(int(S.samplingFreq * S.intTime * 14.5))
    runs = []
    try:
        for w in (window, 0.0, window, 0.0):
            S.gnsscorrWindowSamples = w
            tr, _ = I.call("tracking", fid, mch, mlab.to_matlab(S), nargout=2)
            runs.append(mlab.from_matlab(tr))
        # the same name, another record: int16 samples with the same values -> the same results, through a context of its own
        rec.astype(np.int16).tofile(path)
        fid16 = mlab.register_file(I, rec.astype(np.int16).tobytes(), path)
        S.gnsscorrWindowSamples = 0.0
        S.dataType = "int16"
        tr16, _ = I.call("tracking", fid16, mch, mlab.to_matlab(S), nargout=2)
        runs.append(mlab.from_matlab(tr16))
    finally:
        I.call("gnsscorr_context", "", "clear")
    for k, other in enumerate(runs[1:], 1):
        for a, b in zip(runs[0], other):
            assert a.status == b.status and np.array_equal(a.absoluteSample, b.absoluteSample)
            for f in ("I_P", "Q_P", "carrFreq", "codeFreq"):
                if k < 4:       # the same int8 record, windowed or resident: bit for bit
                    assert np.array_equal(getattr(a, f), getattr(b, f)), (k, f)
                else:           # the int16 record runs the 8-sample-chunk kernels: the same sums in another order of float additions
                    assert np.allclose(getattr(a, f), getattr(b, f), rtol=0, atol=1e-5 * 2 * 18000 * 28 if f in ("I_P", "Q_P") else 1e-3), (k, f)


@pytest.mark.parametrize("name", ["GPS_L1CA", "GPS_L5C", "BDS_B1C_WB"])
def test_matlab_drop_in_with_the_loop_closed_on_the_gpu(gateway, name, tmp_path):
    """settings.gnsscorrDeviceLoop: the drop-in's loops run in one persistent launch (gnsscorr_mex('track_device') -> gc_track_device)
    and the reference's own trackResults come back within the tolerances of the device-loop tests."""
    import bridge
    import cu_sdr_collection_amd as P
    from oracle import mlab
    from types import SimpleNamespace
    sc = next(s for s in RS.TRACK_SCENES if s.name == name)
    z = np.load(os.path.join(GOLD, f"ref_track_{sc.name}.npz"))
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    S.gnsscorrDeviceLoop = 1.0
    path = str(tmp_path / "record.bin")
    rec.tofile(path)
    I = bridge.install(bridge.interpreter_for(_WRAPPER_DIR[sc.signal]), gateway, P, sc.signal)
    fid = mlab.register_file(I, rec.tobytes(), path)
    mch = mlab.to_matlab([SimpleNamespace(**{k: (v if isinstance(v, str) else float(v)) for k, v in vars(c).items()}) for c in ch])
    try:
        tr, _ = I.call(sc.fn, fid, mch, mlab.to_matlab(S), nargout=2)
    finally:
        I.call("gnsscorr_context", "", "clear")
    tr = mlab.from_matlab(tr)
    calls = [c for c, _ in gateway.calls] if hasattr(gateway, "calls") else None
    assert calls is None or "track_device" in calls
    full = 2 * S.samplingFreq * S.intTime * 28.0
    for k in range(2):
        assert tr[k].status == str(z["status"][k])
        assert np.array_equal(np.asarray(tr[k].absoluteSample, dtype=np.float64).reshape(-1), z["f_absoluteSample"][k])
        assert np.max(np.abs(np.asarray(tr[k].carrFreq).reshape(-1) - z["f_carrFreq"][k])) < 1e-3
        assert np.max(np.abs(np.asarray(tr[k].I_P).reshape(-1) - z["f_I_P"][k])) < 1e-5 * full


_ACQ_FORMATS = ("GPS_L1CA", "GPS_L5C", "GLO_GL1", "BDS_B1I", "GPS_L2C", "BDS_B1C", "GPS_L5C_resampled")


@pytest.mark.parametrize("scale", [100.0, 0.37], ids=["int16-valued", "any-complex-row"])
@pytest.mark.parametrize("sc", [s for s in RS.ACQ_SCENES if s.name in _ACQ_FORMATS], ids=[s.name for s in RS.ACQ_SCENES if s.name in _ACQ_FORMATS])
def test_matlab_drop_in_takes_longsignal_as_the_reference_does(gateway, sc, scale):
    """acquisition(longSignal, settings) takes a complex double row, whatever the file's dataType was (postProcessing.m:61-96 reads
    int8 or int16).  The results do not depend on the signal's scale (codePhase and carrFreq are positions, peakMetric a ratio), so
    the reference's results for the int8 record must come back for the same samples times 100 - int16 values, uploaded as an int16
    record whose float copy the searches read - and times 0.37 - no integers at all, longSignal itself goes up in single precision
    (the conditioning block, which wants a record, then refuses)."""
    import bridge
    import cu_sdr_collection_amd as P
    from oracle import mlab
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    x = rec.astype(np.float64) * scale
    long_signal = (x[0::2] + 1j * x[1::2]).reshape(1, -1)
    pkg = sc.name.replace("_resampled", "")
    signal = {"BDS_B1C": "BDS_B1C_NB"}.get(pkg, pkg)
    I = bridge.install(bridge.interpreter_for(pkg), gateway, P, signal)
    try:
        if sc.name.endswith("_resampled") and scale != round(scale):
            with pytest.raises(mlab.MError) as e:
                I.call("acquisition", long_signal, mlab.to_matlab(S))
            assert "gnsscorr:acquisition" in str(e.value)      # (the interpreter keeps the identifier; the text names the conditioning block)
            return
        acq = mlab.from_matlab(I.call("acquisition", long_signal, mlab.to_matlab(S)))
    finally:
        I.call("gnsscorr_context", "", "clear")
    for f in sc.fields:
        want, got = z["f_" + f], np.asarray(getattr(acq, f), dtype=np.float64).reshape(-1)
        assert got.shape == want.shape, (sc.name, f, got.shape, want.shape)
        if f == "peakMetric":
            # samples times 0.37 go up in single precision: the signal itself is rounded at 6e-8 (the guard's float64 cannot undo that)
            rtol = sc.metric_rtol if scale == round(scale) else max(sc.metric_rtol, 5e-7)
            assert np.max(np.abs(got - want)) <= rtol * np.max(np.abs(want)), (sc.name, np.max(np.abs(got - want)) / np.max(np.abs(want)))
        else:
            assert np.array_equal(got, want), (sc.name, f, got[got != want], want[got != want])
