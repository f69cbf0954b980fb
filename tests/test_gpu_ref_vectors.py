"""The HIP path against what the REFERENCE'S OWN .m FILES computed on the same inputs (tests/golden/ref_*.npz, generated in the
build container by executing tracking.m / NB_tracking.m / WB_tracking.m / acquisition.m of every package with oracle/mlab —
tests/golden/make_ref_vectors.py).  No oracle in between: receiver.tracking / the acquisition entry points call the C-ABI,
the expected values are the reference's.

Tolerances (north_star: correlator I/Q within a stated float tolerance, acquired code-phase indices bit-exact):
  block geometry (absoluteSample), acquisition codePhase / carrFreq / CLCodePhase       identical
  correlator sums                                                                      1e-5 of full scale 2*N*28 (f32 accumulate)
  loop state: carrFreq, codeFreq 1e-3 Hz; remCodePhase 1e-7 chip; discriminators 1e-5   (they integrate the f32 sums)
  C/N0 records 1e-3 dB; peakMetric 1e-9 relative (float32 FFTs find the cell, the float64 guard evaluates it: csrc/acq_guard.h; 2e-7 on a float32 conditioned signal)
  which entries are Inf / 0 (epochs and channels never processed)                       identical"""
import os

import numpy as np
import pytest

import ref_scenes as RS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_SUMS = ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")


# scenes the device-closed loop (gc_track_device) is instantiated for: all of them
_DEVICE_LOOP_SCENES = ("GPS_L1CA", "GPS_L1CA_int16_skip", "GPS_L1CA_real", "GPS_L5C", "GPS_L5C_data_only", "BDS_B2a", "BDS_B3I", "BDS_B1I",
                       "GAL_E1C", "GAL_E5a", "GAL_E5b", "GLO_GL1", "GLO_GL2", "BDS_B1C_NB", "BDS_B1C_WB", "GPS_L2C")


@pytest.mark.parametrize("sc", [s for s in RS.TRACK_SCENES if s.name in _DEVICE_LOOP_SCENES], ids=[s.name for s in RS.TRACK_SCENES if s.name in _DEVICE_LOOP_SCENES])
def test_device_closed_loop_equals_the_references_tracking_m(engine, sc):
    """The same comparison with the loop closed on the GPU (one persistent launch, tracking.m:302-335 executed by the kernel)."""
    _tracking_against_the_reference(engine, sc, device_loop=True)


@pytest.mark.parametrize("sc", RS.TRACK_SCENES, ids=[s.name for s in RS.TRACK_SCENES])
def test_hip_tracking_equals_the_references_tracking_m(engine, sc):
    _tracking_against_the_reference(engine, sc, device_loop=False)


def _tracking_against_the_reference(engine, sc, device_loop):
    import cu_sdr_collection_amd as P
    z = np.load(os.path.join(GOLD, f"ref_track_{sc.name}.npz"))
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    assert RS.crc(rec) == int(z["record_crc32"][0]), "the synthetic record is not the one the fixture was generated from"
    engine.load_if(rec, layout=layout, fs=S.samplingFreq)
    tr, _ = P.tracking(engine, ch, S, signal=sc.signal, device_loop=device_loop)
    ref_fields = [k[2:] for k in z.files if k.startswith("f_") and k != "f_PRN"]
    comp = 1 if layout == RS.GC_REAL else 2
    amp = 5.0 if rec.dtype == np.int16 else 1.0
    n_blk = S.samplingFreq * S.intTime
    full = comp * n_blk * 28.0 * amp
    assert [t.status for t in tr] == [str(s) for s in z["status"]]
    for k, t in enumerate(tr):
        have_fields = {f for f in vars(t) if isinstance(getattr(t, f), np.ndarray)}
        assert have_fields == set(ref_fields), (sc.name, sorted(have_fields ^ set(ref_fields)))     # exactly the reference's trackResults fields
        for f in ref_fields:
            want, have = z["f_" + f][k], getattr(t, f)
            assert have.shape == want.shape, (sc.name, f)
            assert np.array_equal(np.isinf(have), np.isinf(want)), (sc.name, k, f)               # untouched entries: inf / 0 as tracking.m:47-86
            m = np.isfinite(want)
            d = float(np.max(np.abs(have[m] - want[m]))) if m.any() else 0.0
            if f == "absoluteSample":
                assert d == 0.0 or (sc.signal == "GPS_L2C" and d < 1e-6), (sc.name, k, d)        # L2C records a fractional sample position
            elif f in _SUMS or f.startswith("Pilot_"):
                assert d < 1e-5 * full, (sc.name, k, f, d / full)
            elif f in ("carrFreq", "codeFreq"):
                assert d < 1e-3, (sc.name, k, f, d)
            elif f == "remCodePhase":
                assert d < 1e-7, (sc.name, k, f, d)
            elif f == "remCarrPhase":
                dd = np.abs(have[m] - want[m])
                assert not m.any() or np.max(np.minimum(dd, np.abs(dd - 2 * np.pi))) < 1e-5, (sc.name, k, f)      # rem(x, 2*pi) of a phase 1e-6 apart may wrap
            elif f in ("dllDiscr", "pllDiscr", "dllDiscrFilt", "pllDiscrFilt"):
                assert d < 2e-5 * max(1.0, float(np.max(np.abs(want[m]))) if m.any() else 1.0), (sc.name, k, f, d)
            elif f.endswith("CNo"):
                assert d < 1e-3, (sc.name, k, f, d)
            elif f.endswith("PLD"):
                assert d < 1e-5, (sc.name, k, f, d)
            else:
                raise AssertionError(f"unexpected field {f}")
        if "cno_VSMValue" in z.files and t.status == "T":
            assert np.allclose(t.CNo.VSMValue, z["cno_VSMValue"][k], atol=1e-3)
            assert np.array_equal(np.asarray(t.CNo.VSMIndex, dtype=np.float64), z["cno_VSMIndex"][k])
    sat = [getattr(c, "K", getattr(c, "PRN", 0)) for c in ch]
    for k, t in enumerate(tr):
        if bool(z["PRN_set"][k]):
            assert t.PRN == z["PRN"][k] == sat[k]


@pytest.mark.parametrize("device_loop", [False, True], ids=["host_loop", "device_loop"])
@pytest.mark.parametrize("sc", RS.LONG_TRACK_SCENES, ids=[s.name for s in RS.LONG_TRACK_SCENES])
def test_long_closed_loop_stays_with_the_references_tracking_m(engine, sc, device_loop):
    """1200 epochs of the reference's own tracking.m (tests/golden/ref_track_GPS_L1CA_long.npz; 800 of GPS L5's with its 3-state PLL
    and the data + pilot pair, ref_track_GPS_L5C_long.npz) against the HIP closed loops.  The
    GPU sums are float32-accumulated, so the loop state carries a small noise-like difference; what must hold over the whole run:
    block starts within one sample (two float implementations of ceil((L - rem)/step) may split a knife edge differently), NCOs
    within a small fraction of the loops' own jitter, identical lock, data bits and C/N0."""
    import cu_sdr_collection_amd as P
    z = np.load(os.path.join(GOLD, f"ref_track_{sc.name}.npz"))
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    assert RS.crc(rec) == int(z["record_crc32"][0])
    engine.load_if(rec, layout=layout, fs=S.samplingFreq)
    tr, _ = P.tracking(engine, ch, S, signal=sc.signal, device_loop=device_loop)
    assert [t.status for t in tr] == [str(s) for s in z["status"]]
    n_ep = tr[0].carrFreq.shape[0]
    for k in range(2):
        t = tr[k]
        assert np.max(np.abs(t.absoluteSample - z["f_absoluteSample"][k])) <= 1.0
        same = t.absoluteSample == z["f_absoluteSample"][k]
        assert same.mean() > 0.99
        jitter = float(np.std(z["f_carrFreq"][k][200:]))
        assert np.max(np.abs(t.carrFreq - z["f_carrFreq"][k])[same]) < max(2e-2, 0.02 * jitter), (k, jitter)
        # The code NCO: 1e-3 Hz for the 1-ms packages.  A 4-ms E1 block has 72 000 samples x 3 taps on half-chip entries: once the
        # code phase has drifted 5e-8 chip from the reference's (float32 partial sums) a sample within that distance of a table edge -
        # one in twenty-five epochs - lands on the other side, +-25 counts on one tap's sum; the DLL answers it, the code phases
        # move apart a little more, more samples flip: the two loops decorrelate their sampling noise (epoch 261 of this scene) and
        # from there differ like two noise realisations of it - up to 0.13 sigma of the loop's own code-NCO jitter (0.95 Hz), 0.4 % of
        # the prompt sum - while every single block replayed from the reference's state still equals the oracle's to 0.02 counts.
        code_jitter = float(np.std(z["f_codeFreq"][k][z["f_codeFreq"][k].shape[0] // 5:]))
        tol_code = 1e-3 if S.intTime <= 0.001 else 0.25 * code_jitter
        assert np.max(np.abs(t.codeFreq - z["f_codeFreq"][k])[same]) < tol_code, (k, code_jitter)
        full = 2 * S.samplingFreq * S.intTime * 28.0
        tol_sum = 1e-4 * full if S.intTime <= 0.001 else 0.01 * float(np.median(np.abs(z["f_I_P"][k])))
        assert np.max(np.abs(t.I_P - z["f_I_P"][k])[same]) < tol_sum and np.max(np.abs(t.Q_P - z["f_Q_P"][k])[same]) < tol_sum
        assert np.array_equal(np.sign(t.I_P[100:]), np.sign(z["f_I_P"][k][100:]))                  # the same navigation bits
        assert np.allclose(t.CNo.VSMValue, z["cno_VSMValue"][k], atol=1e-2) and len(t.CNo.VSMValue) == n_ep // int(S.CNo.VSMinterval)
        if "f_Pilot_I_P" in z.files:
            assert np.max(np.abs(t.Pilot_I_P - z["f_Pilot_I_P"][k])[same]) < tol_sum and np.max(np.abs(t.Pilot_Q_P - z["f_Pilot_Q_P"][k])[same]) < tol_sum


@pytest.mark.parametrize("sc", RS.ACQ_SCENES, ids=[s.name for s in RS.ACQ_SCENES])
def test_hip_acquisition_equals_the_references_acquisition_m(engine, sc):
    import cu_sdr_collection_amd as P
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    assert RS.crc(rec) == int(z["record_crc32"][0])
    engine.load_if(rec, fs=S.samplingFreq)
    got = sc.product(P, engine, S)
    for f in sc.fields:
        want = z["f_" + f]
        have = np.asarray(getattr(got, f), dtype=np.float64)
        assert have.shape == want.shape, (sc.name, f, have.shape, want.shape)
        if f == "peakMetric":
            assert np.max(np.abs(have - want)) <= sc.metric_rtol * np.max(np.abs(want)), (sc.name, np.max(np.abs(have - want)) / np.max(np.abs(want)))
        else:
            assert np.array_equal(have, want), (sc.name, f, np.flatnonzero(have != want), have[have != want], want[have != want])
    assert np.count_nonzero(z["f_carrFreq"]) >= 1


@pytest.mark.parametrize("sc", RS.GUARD_ACQ_SCENES, ids=[s.name for s in RS.GUARD_ACQ_SCENES])
def test_hip_acquisition_on_constructed_near_ties_and_near_threshold_metrics(engine, sc):
    """The float64 guard (csrc/acq_guard.h).  The searches transform in float32; the reference decides `max(max(results))` and
    `peakMetric > acqThreshold` in float64 (GPS_L1CA/include/acquisition.m:196-206).  Records with two cells of `results` 2e-7 apart
    (two columns: the same satellite twice, one sample apart; two bins: a real carrier, whose +-500 Hz bins are conjugates) and a
    threshold 2.5e-7 below / above a satellite's metric: codePhase, carrFreq AND the detected set identical to the reference's own
    acquisition.m, the metric to 1e-9 - the winner's cell is re-evaluated in float64 for every PRN, all cells within eps of a
    near-tied winner are, and the reference's first-occurrence rule runs on those values."""
    import cu_sdr_collection_amd as P
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    assert RS.crc(rec) == int(z["record_crc32"][0])
    engine.load_if(rec, fs=S.samplingFreq)
    got = sc.product(P, engine, S)
    st = engine.acq_guard_stats()
    for f in sc.fields:
        want, have = z["f_" + f], np.asarray(getattr(got, f), dtype=np.float64)
        if f == "peakMetric":
            assert np.max(np.abs(have - want)) <= 1e-9 * np.max(np.abs(want)), (sc.name, np.max(np.abs(have - want)) / np.max(np.abs(want)))
        else:
            assert np.array_equal(have, want), (sc.name, f, have[have != want], want[have != want], st)
    assert np.array_equal(np.asarray(got.carrFreq) != 0, z["f_carrFreq"] != 0)          # the detected set
    assert 5e-6 < st["eps"] < 2e-5 and st["max_dev"] < st["eps"] / 8, st                  # float32 stayed well inside the band the guard assumes
    if "tie" in sc.name:
        assert st["ties"] >= 1, st                                                       # the slow path did run for the constructed tie


def _compare_acq(sc, z, got, float32_metric=False):
    """float32_metric: a path without the float64 guard (the fused kernel of the tuning build, the circshift family PRN by PRN) - its
    peakMetric carries the float32 transforms' rounding, 2e-3 as before round 6."""
    for f in sc.fields:
        want = z["f_" + f]
        have = np.asarray(getattr(got, f), dtype=np.float64)
        assert have.shape == want.shape, (sc.name, f, have.shape, want.shape)
        if f == "peakMetric":
            rtol = 2e-3 if float32_metric else sc.metric_rtol
            assert np.max(np.abs(have - want)) <= rtol * np.max(np.abs(want)), (sc.name, np.max(np.abs(have - want)) / np.max(np.abs(want)))
        else:
            assert np.array_equal(have, want), (sc.name, f, np.flatnonzero(have != want), have[have != want], want[have != want])


@pytest.mark.parametrize("sc", RS.DEFAULT_ACQ_SCENES, ids=[s.name for s in RS.DEFAULT_ACQ_SCENES])
def test_hip_acquisition_at_the_references_default_search_sizes(engine, sc):
    """settings = initSettings() UNMODIFIED - the searches the packages ship with: GPS L1 C/A 32 PRNs x 29 bins x 20 hops
    (GPS_L1CA/initSettings.m:80-89), L5 25 hops, Galileo E5b 168 bins of 60 Hz x 15 hops x 36 PRNs, E1 94 bins x 144 000 points, BDS B3I
    63 PRNs, B1C 62 PRNs x 201 bins x 360 000 points, L2C 401 bins x 2 sub-bin shifts x 320 000 points, B1I 53 PRNs, GLONASS K = -7..6.
    The fixtures are the reference's own acquisition.m executed on the scene's record (oracle/mlab, minutes each, made once:
    tests/golden/make_ref_vectors.py acq_default); codePhase and carrFreq (coarse bin + fine stage) must be IDENTICAL
    (north_star: "acquired code-phase sample indices bit-exact"), the peak metric to 1e-9 (the winning cell re-evaluated in float64) - at the depth
    the hop groups, slot reductions and shifted spectra actually run at."""
    import cu_sdr_collection_amd as P
    if not os.path.exists(os.path.join(GOLD, f"ref_acq_{sc.name}.npz")):
        pytest.skip(f"tests/golden/ref_acq_{sc.name}.npz has not been generated (make_ref_vectors.py acq_default --only {sc.name})")
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    assert sc.overrides == {} and RS.crc(rec) == int(z["record_crc32"][0])
    engine.load_if(rec, fs=S.samplingFreq)
    got = sc.product(P, engine, S)
    _compare_acq(sc, z, got)
    assert np.count_nonzero(z["f_carrFreq"]) >= 2


_ACQ_FUSED_DEFAULT = ("GPS_L1CA_default", "GPS_L5C_default", "GAL_E5a_default", "BDS_B2a_default", "BDS_B3I_default", "GLO_GL1_default", "GLO_GL2_default")


@pytest.mark.tuning
@pytest.mark.parametrize("sc", [s for s in RS.DEFAULT_ACQ_SCENES if s.name in _ACQ_FUSED_DEFAULT], ids=[s.name for s in RS.DEFAULT_ACQ_SCENES if s.name in _ACQ_FUSED_DEFAULT])
def test_fused_inverse_transform_kernel_at_the_references_default_search_sizes(engine, sc, monkeypatch):
    """GC_ACQ_FUSED=1 (the single-launch inverse side) on the same default-size searches."""
    import cu_sdr_collection_amd as P
    monkeypatch.setenv("GC_ACQ_FUSED", "1")
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    engine.load_if(rec, fs=S.samplingFreq)
    _compare_acq(sc, z, sc.product(P, engine, S), float32_metric=True)


@pytest.mark.tuning
@pytest.mark.parametrize("name", ["GPS_L1CA_default", "GPS_L5C_default"])
def test_default_size_searches_in_one_chunk_of_bins(engine, monkeypatch, name):
    """The default GPS L1 C/A and L5 searches run their bins in two chunks (both lanes' intermediates then fit the last-level cache,
    DESIGN.md 4.4); GC_ACQ_BIN_CHUNKS=1 is the search in one piece, same results."""
    import cu_sdr_collection_amd as P
    sc = next(s for s in RS.DEFAULT_ACQ_SCENES if s.name == name)
    monkeypatch.setenv("GC_ACQ_BIN_CHUNKS", "1")
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    engine.load_if(rec, fs=S.samplingFreq)
    _compare_acq(sc, z, sc.product(P, engine, S))


@pytest.mark.tuning
@pytest.mark.parametrize("name", ["BDS_B1I_default", "GPS_L2C_default", "BDS_B1C_default"])
def test_circshift_family_at_the_default_sizes_prn_by_prn(engine, monkeypatch, name):
    """The circshift family's default-size searches run as ONE library call per package (gc_acq_shift_search_batch, the default path of
    test_hip_acquisition_at_the_references_default_search_sizes); GC_ACQ_SHIFT_PER_PRN=1 is the PRN loop it replaced - row maxima and
    the winning row read back per PRN, the selection rules on the host: the same fixtures, the same results."""
    import cu_sdr_collection_amd as P
    sc = next(s for s in RS.DEFAULT_ACQ_SCENES if s.name == name)
    monkeypatch.setenv("GC_ACQ_SHIFT_PER_PRN", "1")
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    engine.load_if(rec, fs=S.samplingFreq)
    _compare_acq(sc, z, sc.product(P, engine, S), float32_metric=True)      # (float32 rows on the host: the guard lives in the batch call)


_ACQ_KNOBS = [("GPS_L1CA", {"GC_ACQ_LANES": "1"}), ("GPS_L5C", {"GC_ACQ_LANES": "1"}), ("GAL_E1C", {"GC_ACQ_PEAK_KERNEL": "1"}),
              ("BDS_B1C", {"GC_ACQ_PEAK_KERNEL": "1"}), ("BDS_B1I", {"GC_ACQ_ROWMAX_KERNEL": "1"}), ("GPS_L2C", {"GC_ACQ_ROWMAX_KERNEL": "1"}),
              ("GPS_L1CA", {"GC_ACQ_NO_HOP_GROUPS": "1"}), ("GLO_GL1", {"GC_ACQ_NATURAL_ORDER": "1"}), ("GAL_E1C", {"GC_ACQ_GENERIC": "1"}),
              ("BDS_B1I", {"GC_ACQ_GENERIC": "1"}),
              # later in round 4: several bins per workgroup of the fused columns pass (forced here: the toy searches are too small to
              # get them by themselves; 3 and 4 leave a remainder), the pairing instead of XCD runs, the fine stage's code periods cut
              # into runs, and the hop groups that two lanes no longer need at the default size
              ("BDS_B1C", {"GC_ACQ_BINS_PER_WG": "3"}), ("GAL_E1C", {"GC_ACQ_BINS_PER_WG": "4"}), ("GPS_L2C", {"GC_ACQ_BINS_PER_WG": "2"}),
              ("BDS_B1I", {"GC_ACQ_BINS_PER_WG": "4"}), ("BDS_B1C", {"GC_ACQ_XCD_MAP": "pairs"}), ("GPS_L1CA", {"GC_ACQ_FINE_PARTS": "8"}),
              ("GPS_L1CA", {"GC_ACQ_HOP_GROUPS": "2"}), ("GAL_E5b", {"GC_ACQ_HOP_GROUPS": "3"}),
              # second half of round 4: bin spacings of q / den FFT bins read den x hops spectra shifted (Galileo E5b 3 / 25, E1 6 / 5) -
              # switched back to one spectrum per (bin, hop); the L1 C/A fine stage's hypothesis search on the host again
              ("GAL_E5b", {"GC_ACQ_NO_RATIONAL_SHIFT": "1"}), ("GAL_E1C", {"GC_ACQ_NO_RATIONAL_SHIFT": "1"}), ("GPS_L1CA", {"GC_ACQ_FINE_HOST": "1"}),
              # a PRN's bins in chunks dealt out to the two lanes (automatic at the default L1 C/A and L5 sizes only): forced here, with a
              # remainder chunk, one and two code arms, and the rational spacing on top
              ("GPS_L1CA", {"GC_ACQ_BIN_CHUNKS": "3"}), ("GPS_L5C", {"GC_ACQ_BIN_CHUNKS": "2"}), ("GAL_E5b", {"GC_ACQ_BIN_CHUNKS": "4"}),
              # the PRN lanes on the context's own stream pair instead of the device's search streams
              ("GPS_L1CA", {"GC_ACQ_LANE_STREAMS": "own"}), ("GAL_E5b", {"GC_ACQ_LANE_STREAMS": "own"}),
              # data + pilot searches arm by arm (two launches per arm, the second arm adding to the first one's sums) instead of both
              # arms in one launch pair
              # arms in one launch pair (automatic for Galileo E1's one hop per bin; forced for the others, with and without chunks)
              ("GAL_E1C", {"GC_ACQ_ARMS_SEPARATE": "1"}), ("GPS_L5C", {"GC_ACQ_ARMS_MERGE": "1"}), ("GAL_E5b", {"GC_ACQ_ARMS_MERGE": "1"}),
              ("BDS_B2a", {"GC_ACQ_ARMS_MERGE": "1", "GC_ACQ_BIN_CHUNKS": "2"}), ("GAL_E5a", {"GC_ACQ_ARMS_MERGE": "1", "GC_ACQ_BIN_CHUNKS": "3"}),
              # BDS B1C's weighted data + pilot arms arm by arm again, and merged in small chunks of rows
              ("BDS_B1C", {"GC_ACQ_ARMS_SEPARATE": "1"}), ("BDS_B1C", {"GC_ACQ_SHIFT_CHUNK_MB": "40"}),
              # round 5: the circshift family's whole PRN list in one library call (gc_acq_shift_search_batch) - back to the PRN loop
              # with its two read-backs per PRN, and the batch with one / two PRN lanes
              ("BDS_B1I", {"GC_ACQ_SHIFT_PER_PRN": "1"}), ("GPS_L2C", {"GC_ACQ_SHIFT_PER_PRN": "1"}), ("BDS_B1C", {"GC_ACQ_SHIFT_PER_PRN": "1"}),
              ("BDS_B1I", {"GC_ACQ_SHIFT_LANES": "1"}), ("GPS_L2C", {"GC_ACQ_SHIFT_LANES": "2"}), ("BDS_B1C", {"GC_ACQ_SHIFT_LANES": "2"})]


@pytest.mark.tuning
@pytest.mark.parametrize("name,env", _ACQ_KNOBS, ids=[f"{n}-{'+'.join(e)}" for n, e in _ACQ_KNOBS])
def test_acquisition_paths_behind_the_tuning_knobs_return_the_references_results_too(engine, monkeypatch, name, env):
    """The round-4 additions of the search each have a switch back to what they replaced - one PRN lane instead of two streams
    (GC_ACQ_LANES=1), the separate peak kernel over the written sums instead of the last pass's per-workgroup candidates
    (GC_ACQ_PEAK_KERNEL=1: one-hop searches), the row-maxima kernel instead of candidates + the winning row transformed again
    (GC_ACQ_ROWMAX_KERNEL=1: circshift family), no hop groups, the natural-order intermediate, the run-time pass kernel: every one
    of them against the same reference-executed fixture as the default path.  (GC_ACQ_ONE_BIN=1, the other switch of the bins per
    workgroup, is what these toy sizes run by default.)"""
    import cu_sdr_collection_amd as P
    sc = next(s for s in RS.ACQ_SCENES if s.name == name)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    with P.Engine(0) as eng:                      # a context of its own: the knobs that are read once per scratch see a fresh one
        eng.load_if(rec, fs=S.samplingFreq)
        # (GC_ACQ_SHIFT_PER_PRN: the circshift family PRN by PRN from the caller's side - float32 rows, no guard; without specialised passes the
        # batch call itself runs PRN after PRN on written rows, guarded)
        _compare_acq(sc, z, sc.product(P, eng, S), float32_metric=name in ("BDS_B1I", "GPS_L2C", "BDS_B1C") and "GC_ACQ_SHIFT_PER_PRN" in env)


_WIDE_BAND = [s for s in RS.ACQ_SCENES + RS.GUARD_ACQ_SCENES] + [s for s in RS.DEFAULT_ACQ_SCENES if s.name in ("GPS_L1CA_default", "GPS_L5C_default", "GLO_GL1_default", "BDS_B1I_default")]


@pytest.mark.tuning
@pytest.mark.parametrize("sc", _WIDE_BAND, ids=[s.name for s in _WIDE_BAND])
def test_guards_slow_path_on_every_search_returns_the_references_results(sc, monkeypatch):
    """GC_ACQ_GUARD_EPS=0.02 (tuning build): every PRN whose runner-up lies within 2 % of its winner - most noise-only PRNs, many rows of
    the circshift searches - goes through the float64 guard's SLOW path: the PRN searched again with its sums written, the cells within
    the band collected and re-evaluated as float64 correlations, the reference's first-occurrence rule on those values.  Data + pilot
    arms with weights, GLONASS' per-row centre frequencies, merged arms, bins in chunks, conditioned (float32) signals, padded
    transforms, the circshift family's row ties and second peaks: codePhase / carrFreq identical to the reference's acquisition.m,
    peakMetric to the scene's (float64) tolerance, as on the fast path."""
    import cu_sdr_collection_amd as P
    path = os.path.join(GOLD, f"ref_acq_{sc.name}.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    monkeypatch.setenv("GC_ACQ_GUARD_EPS", "0.02")
    z = np.load(path)
    S, rec = RS.acq_inputs(P, sc)
    with P.Engine(0) as eng:
        eng.load_if(rec, fs=S.samplingFreq)
        got = sc.product(P, eng, S)
        st = eng.acq_guard_stats()
    _compare_acq(sc, z, got)
    assert st["eps"] == 0.02
    if sc.name in ("GPS_L1CA_default", "GPS_L5C_default", "BDS_B1I_default", "GPS_L1CA", "BDS_B1I_tie_rows"):
        assert st["ties"] >= 1, st          # (the default lists' noise-only PRNs: among 25 - 50 of them some have their two largest cells within 2 %)


_ACQ_FUSED = ("GPS_L1CA", "GPS_L5C", "GAL_E5a", "GAL_E5b", "BDS_B2a", "BDS_B3I", "GLO_GL1")   # searches of 36 000 / 24 000 points


@pytest.mark.tuning
@pytest.mark.parametrize("sc", [s for s in RS.ACQ_SCENES if s.name in _ACQ_FUSED], ids=[s.name for s in RS.ACQ_SCENES if s.name in _ACQ_FUSED])
def test_fused_inverse_transform_kernel_equals_the_references_acquisition_m(engine, sc, monkeypatch):
    """GC_ACQ_FUSED=1: the inverse side of the search in one launch (acq_fused_kernel: radix-4 decimation-in-frequency step, four
    N / 4-point transforms in LDS, |.| and hop sums in registers, no intermediate in memory) - the alternative DESIGN.md 4.4 measures
    against the default two passes - returns the reference's codePhase / carrFreq / peakMetric as well (data + pilot arms, shifted
    and unshifted hop spectra, GLONASS' 24 000 points)."""
    import cu_sdr_collection_amd as P
    monkeypatch.setenv("GC_ACQ_FUSED", "1")
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    engine.load_if(rec, fs=S.samplingFreq)
    got = sc.product(P, engine, S)
    for f in sc.fields:
        want = z["f_" + f]
        have = np.asarray(getattr(got, f), dtype=np.float64)
        if f == "peakMetric":
            assert np.max(np.abs(have - want)) <= 2e-3 * np.max(np.abs(want)), (sc.name, f)     # (no float64 guard on the fused kernel's path)
        else:
            assert np.array_equal(have, want), (sc.name, f, have[have != want], want[have != want])


_ACQ_INT16 = ("GPS_L1CA", "GPS_L5C", "GAL_E1C", "GLO_GL1", "BDS_B1I", "GPS_L2C", "BDS_B1C", "GPS_L1CA_resampled", "BDS_B1C_resampled")


@pytest.mark.parametrize("sc", [s for s in RS.ACQ_SCENES if s.name in _ACQ_INT16], ids=[s.name for s in RS.ACQ_SCENES if s.name in _ACQ_INT16])
def test_hip_acquisition_of_an_int16_record_equals_the_references_acquisition_m(engine, sc):
    """settings.dataType = 'int16' (postProcessing.m:61-96): the record's samples times 100 as an int16 record.  The searches read
    its float copy on the device (gc_acq_signal_from_record), the conditioning block reads the int16 record itself; positions and
    the scale-free metric must be the reference's for the int8 record."""
    import cu_sdr_collection_amd as P
    z = np.load(os.path.join(GOLD, f"ref_acq_{sc.name}.npz"))
    S, rec = RS.acq_inputs(P, sc)
    engine.load_if(rec.astype(np.int16) * 100, fs=S.samplingFreq)
    got = sc.product(P, engine, S)
    for f in sc.fields:
        want = z["f_" + f]
        have = np.asarray(getattr(got, f), dtype=np.float64)
        assert have.shape == want.shape, (sc.name, f)
        if f == "peakMetric":
            assert np.max(np.abs(have - want)) <= sc.metric_rtol * np.max(np.abs(want)), (sc.name, np.max(np.abs(have - want)) / np.max(np.abs(want)))
        else:
            assert np.array_equal(have, want), (sc.name, f, have[have != want], want[have != want])


_REPLAY_SCENES = RS.TRACK_SCENES + RS.LONG_TRACK_SCENES


@pytest.mark.parametrize("sc", _REPLAY_SCENES, ids=[s.name for s in _REPLAY_SCENES])
def test_correlator_replayed_from_the_references_own_state_returns_the_references_sums(engine, sc):
    """tracking.m records, per epoch, the state its block was cut from (absoluteSample, remCodePhase, codeFreq, carrFreq,
    remCarrPhase: :212-216,249,277,314,332) next to the six sums: the HIP correlator fed with the REFERENCE's state - every epoch of
    the reference-executed runs (fifteen packages / record formats and the three long runs), one batched launch each - must return
    the reference's sums.  This takes the loop out of the
    comparison: closed loops that differ by 5e-8 chip of code phase end up with different samples on the table edges and drift apart
    like two noise realisations (the long closed-loop test above), the correlator itself does not."""
    import cu_sdr_collection_amd as P
    from cu_sdr_collection_amd import signals
    z = np.load(os.path.join(GOLD, f"ref_track_{sc.name}.npz"))
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    engine.load_if(rec, layout=layout, fs=S.samplingFreq)
    spec = signals.SIGNALS[sc.signal]
    nch, n_ep = sum(1 for s in z["status"] if str(s) == "T"), z["f_carrFreq"].shape[1]
    assert nch == 2 and np.all(np.isfinite(z["f_absoluteSample"][:nch]))
    for k in range(nch):
        engine.set_channel(k, spec.tables(int(z["PRN"][k]), S), index_scale=spec.index_scale, arm_mult=spec.arm_mult, windows=spec.windows)
    blocks = engine.make_blocks(nch * n_ep)
    fs = S.samplingFreq
    l2c = spec.doubled_code
    for e in range(n_ep):
        for k in range(nch):
            b = blocks[e * nch + k]
            step = float(z["f_codeFreq"][k][e]) / fs
            rem = float(z["f_remCodePhase"][k][e])
            pos = float(z["f_absoluteSample"][k][e])
            length, spacing = S.codeLength, S.dllCorrelatorSpacing
            if l2c:
                # GPS L2C works on the RZ-doubled code (GPS_L2C tracking.m:107-109,171) and RECORDS in single-code units, the position
                # pushed back by the remainder in samples (:223,250,376,382-383): undo that to get the block the sums were taken over
                step, rem, length, spacing = 2 * step, 2 * rem, 2 * S.codeLength, 2 * S.dllCorrelatorSpacing
                pos = float(np.rint(pos - 1 + rem / step))
                b.table_offset[1] = int(length) * ((int(ch[k].CLCodePhase) - 1 + e) % 75)       # :261,357-360
            b.channel = k
            b.first_sample = int(pos)
            b.rem_code_phase = rem
            b.code_phase_step = step
            b.blksize = int(np.ceil((length - rem) / step))                                         # tracking.m:219-222
            b.el_spacing = spacing
            b.carr_freq = float(z["f_carrFreq"][k][e])
            b.rem_carr_phase = float(z["f_remCarrPhase"][k][e])
    engine.replay_prepare(blocks)
    engine.replay_launch()
    out = engine.replay_fetch().reshape(n_ep, nch, -1, 6)
    comp = 1 if layout == RS.GC_REAL else 2
    full = comp * fs * S.intTime * 28.0 * (5.0 if rec.dtype == np.int16 else 1.0)
    names = ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")
    for k in range(nch):
        want = np.stack([z["f_" + n][k] for n in names], axis=1)
        d = np.max(np.abs(out[:, k, 0, :] - want)) / full
        assert d < 2e-6, (sc.name, k, d)
        if "f_Pilot_I_P" in z.files and spec.pilot_combine in (1, 2, 3):
            # GPS L5 records the pilot arm as correlated (Pilot_I_P / Pilot_Q_P, GPS_L5C/include/tracking.m:321-326)
            dp = max(np.max(np.abs(out[:, k, 1, 2] - z["f_Pilot_I_P"][k])), np.max(np.abs(out[:, k, 1, 3] - z["f_Pilot_Q_P"][k]))) / full
            assert dp < 2e-6, (sc.name, k, dp)
        if spec.pilot_combine == 4:
            # BDS B1C wide-band records ONE pilot: the BOC(6,1) and BOC(1,1) pilot arms folded sqrt(4/33) : sqrt(29/33) in
            # quadrature (WB_tracking.m:364-369) - formed here from the two arms the correlator returns
            a61, a11 = -np.sqrt(4.0 / 33.0), np.sqrt(29.0 / 33.0)
            for x, tap in enumerate(("E", "P", "L")):
                i11, q11, i61, q61 = out[:, k, 1, 2 * x], out[:, k, 1, 2 * x + 1], out[:, k, 2, 2 * x], out[:, k, 2, 2 * x + 1]
                di = np.max(np.abs(a61 * i61 + a11 * q11 - z["f_Pilot_I_" + tap][k])) / full
                dq = np.max(np.abs(a61 * q61 - a11 * i11 - z["f_Pilot_Q_" + tap][k])) / full
                assert di < 2e-6 and dq < 2e-6, (sc.name, k, tap, di, dq)


@pytest.mark.parametrize("sc", RS.NAVSYNC_SCENES, ids=[s.name for s in RS.NAVSYNC_SCENES])
def test_hip_bit_sync_equals_the_references_navdecoding_m(engine, sc):
    """SURVEY §8f.4, no oracle in between: gc_sync_xcorr + nav_sync's table against what the synchronisation block of the package's
    own NAVdecoding.m computes on the same prompt stream (tests/golden/ref_navsync_*: the file's lines executed in place) -
    tlmXcorrResult over the non-negative lags bit for bit, `index`, and the verified start where the block has a verification."""
    import hashlib
    import cu_sdr_collection_amd as P
    from oracle import gnss_oracle as O
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"ref_navsync_{sc.name}.npz"), allow_pickle=False)
    x = RS.navsync_stream(sc, P.nav_sync.SYNC[sc.package].pattern(sc.prn), parity_check=O.nav_parity_check)
    assert RS.crc(x) == int(ref["stream_crc32"][0])
    got = P.nav_sync.find_sync(engine, sc.package, x, sc.ms_to_process, prn=sc.prn)
    r = got.xcorr
    assert r.dtype == np.float32 and r.shape[0] == int(ref["xcorr_len"][0]) and np.array_equal(r, np.rint(r))
    assert hashlib.sha256(r.astype(np.int16).tobytes()).hexdigest() == str(ref["xcorr_sha256"])
    assert np.array_equal(r[:4096].astype(np.int16), ref["xcorr_head"])
    assert np.array_equal(got.index, ref["index"]) and got.index.size > 0
    if sc.loop_var:
        assert (got.first if got.first is not None else -1) == int(ref["first"][0])
    # the spacing rule of the packages whose verification is the navigation decoder proper: the oracle's restatement
    _, _, cand, _ = O.nav_sync(sc.package, x, sc.ms_to_process, sc.prn)
    assert np.array_equal(got.candidates, cand)


def test_sync_xcorr_edges_and_arguments(engine):
    """Streams shorter than the pattern, lengths around the kernel's 1024-lag tiles, the two zero rules, argument errors."""
    import cu_sdr_collection_amd as P
    from cu_sdr_collection_amd import _lib as L
    from oracle import gnss_oracle as O
    rng = np.random.default_rng(77)
    for n, m in ((5, 160), (1023, 300), (1024, 10), (1025, 240), (3000, 1), (4097, 8192)):
        x = rng.standard_normal(n)
        x[rng.integers(0, n, 3)] = 0.0
        pat = rng.integers(-1, 2, m).astype(np.int8)
        for zp in (False, True):
            bits = (1.0 - 2.0 * (x < 0)) if zp else np.where(x > 0, 1.0, -1.0)
            assert np.array_equal(engine.sync_xcorr(x, pat, zero_is_plus=zp), O.xcorr_nonneg(bits, pat.astype(np.float64)).astype(np.float32)), (n, m, zp)
    with pytest.raises(L.GnssCorrError) as e:
        engine.sync_xcorr(np.ones(10), np.ones(8193, dtype=np.int8))
    assert e.value.status == L.GC_E_INVALID
    with pytest.raises(ValueError):
        P.nav_sync.find_sync(engine, "BDS_B1I", np.ones(900), prn=8)          # the stream ends before searchStartOffset = 1000
