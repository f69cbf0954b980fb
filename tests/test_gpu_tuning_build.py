"""The knob tests on the tuning build.

libgnsscorr.so reads no tuning variable: every A/B switch of docs/KNOBS.md goes through GC_TUNE_ENV (csrc/gc_internal.h), a null
pointer in the library that ships and std::getenv in libgnsscorr_tuning.so (-DGC_TUNING=1, same sources, plus the kernels that lost
their A/B: csrc/corr_cboc.hip).  Tests marked `tuning` are skipped in a process that loaded the product library (tests/conftest.py);
this file runs them - and nothing else - in a child process that loads the tuning build through GC_LIB_PATH, so one `pytest -m gpu`
covers both builds."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_library_that_ships_reads_no_tuning_variable():
    from cu_sdr_collection_amd import _lib as L
    if L.is_tuning_build():
        pytest.skip("this process loaded the tuning build")
    data = open(L.LIB_PATH, "rb").read()
    import glob
    knobs = set()
    for src in glob.glob(os.path.join(ROOT, "cu-sdr-collection_amd", "csrc", "*")):
        knobs |= set(re.findall(r'GC_TUNE_ENV\("(GC_[A-Z0-9_]+)"\)', open(src).read()))
    assert len(knobs) >= 55                                                # every A/B switch of docs/KNOBS.md goes through the macro ...
    assert not [k for k in sorted(knobs) if k.encode() in data]            # ... and none of their names is in the library that ships
    assert os.path.getsize(L.LIB_PATH) < 7 * 1024 * 1024
    tuned = open(L.TUNING_LIB_PATH, "rb").read()
    assert all(k.encode() in tuned for k in knobs)


def test_knob_tests_on_the_tuning_build():
    from cu_sdr_collection_amd import _lib as L
    if L.is_tuning_build():
        pytest.skip("this process loaded the tuning build: the knob tests run in it directly")
    assert os.path.exists(L.TUNING_LIB_PATH), "libgnsscorr_tuning.so not built (cu_sdr_collection_amd.build)"
    env = dict(os.environ, GC_LIB_PATH=L.TUNING_LIB_PATH)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-x", "-m", "gpu and tuning", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=3000, env=env, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 50, tail        # the acquisition knob matrix alone has 50 cases
    assert "skipped" not in r.stdout.splitlines()[-1], tail
    print(r.stdout.splitlines()[-1])
