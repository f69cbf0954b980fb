"""The oracle against what the REFERENCE'S OWN .m FILES computed (tests/golden/ref_*.npz: the reference's source text executed
by the mini-MATLAB interpreter of oracle/mlab in the build container, tests/golden/make_ref_vectors.py).  This is the pin of the
oracle that round 1 lacked: a misreading of tracking.m / acquisition.m / preRun.m / initSettings.m / a code generator in
oracle/gnss_oracle.py (or in the product's host side) shows up here as a difference from the reference's executed statements."""
import json
import os

import numpy as np
import pytest

import ref_scenes as RS
from oracle import gnss_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.mark.parametrize("sc", RS.TRACK_SCENES, ids=[s.name for s in RS.TRACK_SCENES])
def test_oracle_tracking_equals_the_references_tracking_m(sc):
    """[trackResults, channel] = tracking(fid, channel, settings) of every package: every recorded field of every epoch.
    Tolerance 1e-12 relative (the float64 restatement differs from the interpreter's NumPy evaluation by summation order at most;
    measured: bit-identical for eight packages, <= 5e-16 for the rest)."""
    import cu_sdr_collection_amd as P
    if sc.oracle is None:
        pytest.skip("no oracle runner for this scene (checked on the GPU against the fixture directly)")
    z = _load(f"ref_track_{sc.name}.npz")
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    assert RS.crc(rec) == int(z["record_crc32"][0]), "the synthetic record is not the one the fixture was generated from"
    got = sc.oracle(O, rec, ch, S)
    assert [t.status for t in got] == [str(s) for s in z["status"]]
    n_active = sum(1 for c in ch if c.status == "T")
    for k in range(n_active):
        for f in RS.TRACK_FIELDS + RS.PILOT_FIELDS:
            if "f_" + f not in z.files:
                continue
            want = z["f_" + f][k]
            have = np.asarray(getattr(got[k], f))
            assert have.shape == want.shape, f
            if f == "absoluteSample":
                assert np.array_equal(have, want), (sc.name, k)
            else:
                scale = np.max(np.abs(want[np.isfinite(want)])) + 1e-300
                assert np.all(np.isfinite(want)) and np.max(np.abs(have - want)) <= 1e-12 * scale, (sc.name, k, f)
        assert float(z["PRN"][k]) == float(getattr(ch[k], "K", getattr(ch[k], "PRN", 0)))


def test_reference_trackresults_field_sets_and_initial_values():
    """tracking.m:47-86 and its per-package variants: which Pilot_* fields exist, which fields start as inf."""
    from cu_sdr_collection_amd import receiver, signals
    for sc in RS.TRACK_SCENES:
        z = _load(f"ref_track_{sc.name}.npz")
        fields = {k[2:] for k in z.files if k.startswith("f_")}
        spec = signals.SIGNALS[sc.signal]
        want_pilot = set(receiver._recorded_pilot_fields(spec, sc.pilot, "reference"))
        assert {f for f in fields if f.startswith("Pilot_")} == want_pilot, sc.name
        idle = -1                                      # the last channel of every scene is never assigned
        for f in RS.TRACK_FIELDS:
            v = z["f_" + f][idle]
            assert np.all(np.isinf(v)) if f in receiver._INF_FIELDS else np.all(v == 0), (sc.name, f)
        assert not bool(z["PRN_set"][idle])            # trackResults(k).PRN stays [] for an idle channel
