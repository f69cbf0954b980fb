import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "tuning: switches kernels through GC_* tuning variables, which only libgnsscorr_tuning.so reads "
                                       "(GC_LIB_PATH; tests/test_gpu_tuning_build.py runs these in a process that loads it)")


def pytest_collection_modifyitems(config, items):
    """Knob tests need the tuning build: with the library that ships the GC_* switches are compiled out (gc_internal.h GC_TUNE_ENV) and
    setting them changes nothing.  They run in the child process of tests/test_gpu_tuning_build.py; here they are skipped."""
    tuned = [it for it in items if it.get_closest_marker("tuning")]
    if not tuned:
        return
    try:
        from cu_sdr_collection_amd import _lib as L
        is_tuning = L.is_tuning_build()
    except Exception:
        return                                    # no library / no GPU: the tests fail on their own, loudly
    if not is_tuning:
        skip = pytest.mark.skip(reason="needs libgnsscorr_tuning.so (run by tests/test_gpu_tuning_build.py with GC_LIB_PATH set)")
        for it in tuned:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    """A live gc_context on cuda:0.  GPU tests FAIL (not skip) when the HIP library is missing
    or no MI355X is visible: there is no CPU fallback to hide behind."""
    import cu_sdr_collection_amd as P
    eng = P.Engine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def l1ca_scene():
    """0.3 s, 4-satellite GPS L1 C/A scene at the reference's default front end
    (18 Msps int8 I/Q, IF 20 kHz: GPS/GPS_L1CA/initSettings.m:60-69)."""
    import numpy as np
    import cu_sdr_collection_amd as P
    S = P.initSettings()
    sats = P.synth.scene(4, 20241008 + 2, S.samplingFreq)
    iq = P.synth.generate_if(sats, int(0.3 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode,
                             S.codeFreqBasis, 1023, seed=77)
    return S, sats, iq
