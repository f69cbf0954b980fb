"""The MATLAB drop-ins (matlab/*.m) on CPU: every file parses, and each package's tracking.m wrapper, executed by the mini-MATLAB
interpreter against a MOCK gateway (no GPU), builds exactly the trackResults struct array the reference's tracking.m builds (field
sets, sizes, inf / 0 initial values - tests/golden/ref_track_*.npz) and passes the gateway what the C-ABI expects.  The real
gateway and the GPU are exercised by tests/test_gpu_mex_gateway.py."""
import glob
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

import ref_scenes as RS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "mexstub"))


def test_every_wrapper_parses():
    from oracle.mlab.parser import parse
    files = glob.glob(os.path.join(ROOT, "matlab", "*.m")) + glob.glob(os.path.join(ROOT, "matlab", "packages", "*", "*.m"))
    assert len(files) >= 16
    for f in files:
        funcs, script = parse(open(f).read(), f)
        assert funcs and script is None, f
    # one tracking stub per package with the reference's signature
    for d in glob.glob(os.path.join(ROOT, "matlab", "packages", "*")):
        for f in glob.glob(os.path.join(d, "*tracking.m")):
            fn = parse(open(f).read(), f)[0][0]
            assert fn[2] == ["fid", "channel", "settings"] and fn[3] == ["trackResults", "channel"], f


class _MockGateway:
    def __init__(self):
        self.calls = []

    def call(self, cmd, *a, nargout=1):
        self.calls.append((cmd, a))
        if cmd == "create":
            return np.array([[0.0]])
        if cmd == "if_info":                       # a fresh context holds no record
            return np.zeros((1, 3))
        if cmd == "track":
            p, chan = a[1], np.asarray(a[2])
            n = int(np.asarray(p["numEpochs"]).flat[0])
            nch = chan.shape[1]
            done = np.full((1, nch), float(n))
            done[0, -1] = n - 3                    # the last active channel stops three epochs early (a short read)
            # 4th output: C/N0 per interval and channel (CNoVSM: one value; Calc_CNo_PLD modes: five), as gnsscorr_mex.c lays it out
            k = int(np.asarray(p["cnoInterval"]).flat[0])
            nv = 5 if int(np.asarray(p["cnoMode"]).flat[0]) else 1
            return np.ones((n, 21 * nch)), done, np.array([[-2.0]]), np.full((nv * (n // k), nch), 45.0)
        return None


@pytest.mark.parametrize("sc", RS.TRACK_SCENES, ids=[s.name for s in RS.TRACK_SCENES])
def test_wrapper_builds_the_references_trackresults(sc):
    import bridge
    import cu_sdr_collection_amd as P
    from oracle import mlab
    S, rec, layout, ch = RS.scene_inputs(P, sc)
    d = {"BDS_B1C_NB": "BDS_B1C", "BDS_B1C_WB": "BDS_B1C"}.get(sc.signal, sc.signal)
    gw = _MockGateway()
    I = bridge.install(bridge.interpreter_for(d), gw, P, sc.signal)
    fid = mlab.register_file(I, b"", "/data/record.bin")
    mch = mlab.to_matlab([SimpleNamespace(**{k: (v if isinstance(v, str) else float(v)) for k, v in vars(c).items()}) for c in ch])
    with np.errstate(all="ignore"):
        tr, chout = I.call(sc.fn, fid, mch, mlab.to_matlab(S), nargout=2)
    tr = mlab.from_matlab(tr)
    z = np.load(os.path.join(ROOT, "tests", "golden", f"ref_track_{sc.name}.npz"))
    ref_fields = {k[2:] for k in z.files if k.startswith("f_") and k != "f_PRN"}
    n = z["f_I_P"].shape[1]
    for k, t in enumerate(tr):
        have = {f for f in vars(t) if isinstance(getattr(t, f), (np.ndarray, float)) and f != "PRN"}
        assert have == ref_fields, sorted(have ^ ref_fields)
        for f in ref_fields:
            assert np.atleast_1d(getattr(t, f)).shape == z["f_" + f][k].shape, f
    for f in ref_fields:        # every recorded field of an active channel took the gateway's values (ones), pilot fields included
        if f.startswith("Pilot_") or f in ("I_P", "Q_E", "carrFreq"):
            if not (sc.signal == "GPS_L2C" and f in ("codeFreq",)):
                assert np.all(np.atleast_1d(getattr(tr[0], f)) == 1), f
    idle = tr[-1]
    assert idle.status == "-" and np.all(np.isinf(idle.codeFreq)) and not np.any(idle.I_P)
    # the short-read channel keeps its '-' status and its untouched tail (tracking.m:241-245,365)
    assert tr[0].status == "T" and tr[1].status == "-"
    assert np.all(tr[1].I_P[:n - 3] == 1) and np.all(tr[1].I_P[n - 3:] == 0) and np.all(np.isinf(tr[1].carrFreq[n - 3:]))
    assert "Not able to read the specified number of samples" in "".join(I.out)
    # what went through the gateway: the file is opened once by name, from byte 0; the start sample follows tracking.m:145-153
    opens = [a for c, a in gw.calls if c == "open_if_file"]
    assert len(opens) == 1 and opens[0][1] == "/data/record.bin" and float(np.asarray(opens[0][2]).flat[0]) == 0
    p = [a for c, a in gw.calls if c == "track"][0][1]
    skip = getattr(S, "skipNumberOfSamples", getattr(S, "skipNumberOfBytes", 0))
    want_skip = skip / 2 if S.dataType == "int16" else skip
    assert float(np.asarray(p["skipSamples"]).flat[0]) == want_skip + (1 if sc.signal == "GPS_L2C" else 0)
    sets = [a for c, a in gw.calls if c == "set_channel"]
    assert len(sets) == 2 and all(len(s[2]) == (3 if sc.signal == "BDS_B1C_WB" else 2 if sc.pilot else 1) for s in sets)
