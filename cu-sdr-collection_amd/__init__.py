"""cu-sdr-collection_amd — MI355X-native acquisition + tracking hot path for the
CU-SDR-Collection GNSS software receivers.

Layout: csrc/ (HIP kernels + the C-ABI of include/gnsscorr.h), _lib.py (ctypes binding),
engine.py (context wrapper), receiver.py (acquisition / preRun / tracking with the reference's
names and struct fields), codes.py (code generators), settings.py (initSettings mirror),
synth.py (synthetic IF records).  Nothing here imports `oracle/`; there is no CPU fallback.
"""
from . import _lib, acq_family, acq_shift, codes, nav_sync, settings, signals, synth  # noqa: F401
from ._lib import GnssCorrError  # noqa: F401
from .engine import Engine, device_count  # noqa: F401
from .receiver import CNoVSM, acquisition, preRun, tracking, tracking_file, tracking_multi  # noqa: F401
from .settings import initSettings  # noqa: F401

__all__ = ["Engine", "device_count", "GnssCorrError", "acquisition", "preRun", "tracking", "tracking_file", "tracking_multi", "CNoVSM", "initSettings",
           "codes", "settings", "synth"]
