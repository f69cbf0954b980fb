"""ctypes binding of libgnsscorr.so (include/gnsscorr.h).  There is no Python or CPU fallback:
if the HIP library is missing or no MI355X is visible, calls raise GnssCorrError."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GC_LIB_PATH") or os.path.join(HERE, "lib", "libgnsscorr.so")  # env override: tuning builds

GC_OK, GC_E_INVALID, GC_E_RANGE, GC_E_NOMEM, GC_E_HIP, GC_E_STATE, GC_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
GC_I8, GC_I16 = 0, 1
GC_REAL, GC_IQ, GC_QI = 0, 1, 2
GC_MAX_ARMS = 3
GC_SYNC_ZERO_IS_PLUS = 1   # gc_sync_xcorr flag
GC_OUT_STRIDE = 6 * GC_MAX_ARMS
GC_PLL_2ND_ORDER, GC_PLL_3_STATE = 0, 1
GC_CNO_VSM, GC_CNO_PLD, GC_CNO_PLD_PILOT_SWAPPED, GC_CNO_PLD_PILOT = 0, 1, 2, 3   # gc_cno_mode
GC_CNO_NPLD = 5

TRK_FIELDS = ["absoluteSample", "codeFreq", "carrFreq", "I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L",
              "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt", "remCodePhase", "remCarrPhase",
              "Pilot_I_E", "Pilot_Q_E", "Pilot_I_P", "Pilot_Q_P", "Pilot_I_L", "Pilot_Q_L"]
GC_TRK_NFIELDS = len(TRK_FIELDS)


class GnssCorrError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"gnsscorr status {status}: {message}")
        self.status = status


class gc_block(C.Structure):
    _fields_ = [("channel", C.c_int32), ("blksize", C.c_int32), ("first_sample", C.c_int64),
                ("rem_code_phase", C.c_double), ("code_phase_step", C.c_double),
                ("el_spacing", C.c_double), ("carr_freq", C.c_double), ("rem_carr_phase", C.c_double),
                ("table_offset", C.c_int32 * GC_MAX_ARMS), ("reserved", C.c_int32)]


class gc_track_params(C.Structure):
    _fields_ = [("sampling_freq", C.c_double), ("code_freq_basis", C.c_double), ("code_length", C.c_double),
                ("el_spacing", C.c_double), ("int_time", C.c_double),
                ("dll_noise_bw", C.c_double), ("dll_damping", C.c_double),
                ("pll_noise_bw", C.c_double), ("pll_damping", C.c_double),
                ("pll_kind", C.c_int32), ("pilot_combine", C.c_int32),
                ("pf1", C.c_double), ("pf2", C.c_double), ("pf3", C.c_double),
                ("skip_samples", C.c_int64), ("n_epochs", C.c_int32), ("table_phase_count", C.c_int32),
                ("pll_weight", C.c_double * 2), ("dll_weight", C.c_double * 2), ("dll_scale", C.c_double),
                ("cno_interval", C.c_int32), ("cno_mode", C.c_int32), ("cno_acc_time", C.c_double)]


class gc_channel_init(C.Structure):
    _fields_ = [("channel", C.c_int32), ("prn", C.c_int32), ("acquired_freq", C.c_double),
                ("code_freq", C.c_double), ("code_phase", C.c_int64), ("table_phase", C.c_int32), ("reserved", C.c_int32)]


class gc_acq_front_params(C.Structure):
    """acquisition.m:46-111 input conditioning (gc_acq_condition)."""
    _fields_ = [("sampling_freq", C.c_double), ("intermediate_freq", C.c_double), ("bandwidth", C.c_double), ("first_sample", C.c_int64),
                ("n_samples", C.c_int64), ("fir_order", C.c_int32), ("reserved", C.c_int32),
                ("band_margin", C.c_double)]


class gc_acq_front_result(C.Structure):
    _fields_ = [("sampling_freq", C.c_double), ("intermediate_freq", C.c_double), ("n_samples", C.c_int64)]


class gc_channel_state(C.Structure):
    """What tracking.m keeps between two blocks of a channel (include/gnsscorr.h: gc_track_resume / gc_track_file)."""
    _fields_ = [("next_sample", C.c_int64), ("code_freq", C.c_double), ("rem_code_phase", C.c_double), ("carr_freq", C.c_double),
                ("rem_carr_phase", C.c_double), ("old_code_nco", C.c_double), ("old_code_error", C.c_double),
                ("old_carr_nco", C.c_double), ("old_carr_error", C.c_double), ("d_carr_error", C.c_double),
                ("d2_carr_error", C.c_double), ("table_phase", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int64)]


class gc_track_job(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("params", C.POINTER(gc_track_params)), ("init", C.POINTER(gc_channel_init)),
                ("out", C.POINTER(C.c_double)), ("epochs_done", C.POINTER(C.c_int32)), ("nch", C.c_int32),
                ("device_loop", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32), ("error", C.c_char * 240)]


class gc_acq_params(C.Structure):
    _fields_ = [("sampling_freq", C.c_double), ("code_freq_basis", C.c_double), ("code_length", C.c_double),
                ("intermediate_freq", C.c_double), ("search_band", C.c_double), ("search_step", C.c_double),
                ("non_coh_time", C.c_int32), ("source", C.c_int32), ("first_sample", C.c_int64),
                ("block_len", C.c_int32), ("code_samples", C.c_int32), ("n_bins", C.c_int32), ("reserved", C.c_int32),
                ("arm_weight", C.c_double * 4)]


class gc_fine_params(C.Structure):
    _fields_ = [("sampling_freq", C.c_double), ("code_freq", C.c_double), ("f0", C.c_double), ("fstep", C.c_double),
                ("first_sample", C.c_int64), ("spc", C.c_int32), ("ncodes", C.c_int32), ("nbins", C.c_int32),
                ("code_len", C.c_int32), ("index_offset", C.c_int32), ("source", C.c_int32),
                ("dc_re", C.c_double), ("dc_im", C.c_double)]


class gc_acq_shift_params(C.Structure):
    _fields_ = [("sampling_freq", C.c_double), ("carrier_f0", C.c_double), ("carrier_step", C.c_double),
                ("first_sample", C.c_int64), ("n", C.c_int32), ("n_signals", C.c_int32), ("n_carriers", C.c_int32),
                ("n_bins", C.c_int32), ("n_arms_max", C.c_int32), ("source", C.c_int32)]


class gc_acq_shift_pick(C.Structure):
    _fields_ = [("row", C.c_int32), ("code_phase", C.c_int32), ("peak", C.c_double), ("second_peak", C.c_double)]


GC_SHIFT_PICK_GLOBAL, GC_SHIFT_PICK_SEQUENTIAL, GC_SHIFT_PICK_SEQUENTIAL_PAIRS = 0, 1, 2


class gc_acq_result(C.Structure):
    _fields_ = [("coarse_bin", C.c_int32), ("code_phase", C.c_int32), ("peak", C.c_double),
                ("peak_metric", C.c_double), ("coarse_freq", C.c_double)]


# every symbol include/gnsscorr.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "gc_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "gc_destroy": (C.c_int, [_P]),
    "gc_last_error": (C.c_char_p, []),
    "gc_api_version": (C.c_int, []),
    "gc_device_info": (C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "gc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gc_synchronize": (C.c_int, [_P]),
    "gc_load_if": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int]),
    "gc_load_if_packed2": (C.c_int, [_P, _P, C.c_uint64]),
    "gc_open_if_file": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]),
    "gc_attach_if": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int]),
    "gc_if_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "gc_alloc_if": (C.c_int, [_P, C.c_uint64, C.c_int, C.c_int]),
    "gc_read_if": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P]),
    "gc_set_channel": (C.c_int, [_P, C.c_int, C.c_int, C.c_double]),
    "gc_set_code": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_double]),
    "gc_set_code_window": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "gc_set_sampling_freq": (C.c_int, [_P, C.c_double]),
    "gc_force_generic_kernel": (C.c_int, [_P, C.c_int]),
    "gc_correlate": (C.c_int, [_P, C.c_int, C.POINTER(gc_block), C.POINTER(C.c_double)]),
    "gc_replay_prepare": (C.c_int, [_P, C.c_int64, C.POINTER(gc_block)]),
    "gc_replay_launch": (C.c_int, [_P]),
    "gc_replay_fetch": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "gc_timer_start": (C.c_int, [_P]),
    "gc_timer_stop": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "gc_track": (C.c_int, [_P, C.POINTER(gc_track_params), C.c_int, C.POINTER(gc_channel_init),
                           C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "gc_track_device": (C.c_int, [_P, C.POINTER(gc_track_params), C.c_int, C.POINTER(gc_channel_init),
                                  C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "gc_track_resume": (C.c_int, [_P, C.POINTER(gc_track_params), C.c_int, C.POINTER(gc_channel_init), C.POINTER(gc_channel_state),
                                  C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gc_track_file": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.POINTER(gc_track_params), C.c_int,
                                C.POINTER(gc_channel_init), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "gc_set_cno_output": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int64]),
    "gc_share_if": (C.c_int, [_P, _P]),
    "gc_track_multi": (C.c_int, [C.c_int, C.POINTER(gc_track_job)]),
    "gc_acquire_coarse": (C.c_int, [_P, C.POINTER(gc_acq_params), C.c_int, _P, C.POINTER(gc_acq_result)]),
    "gc_acquire_coarse_multi": (C.c_int, [_P, C.POINTER(gc_acq_params), C.c_int, C.c_int, _P, C.POINTER(gc_acq_result)]),
    "gc_acquire_coarse_offsets": (C.c_int, [_P, C.POINTER(gc_acq_params), C.c_int, C.c_int, _P, C.POINTER(C.c_double), C.POINTER(gc_acq_result)]),
    "gc_acquire_fine_l1ca": (C.c_int, [_P, C.POINTER(gc_acq_params), _P, C.c_int, C.c_double,
                                       C.POINTER(C.c_double)]),
    "gc_acquire_fine_sums": (C.c_int, [_P, C.POINTER(gc_fine_params), _P, C.POINTER(C.c_double)]),
    "gc_acq_condition": (C.c_int, [_P, C.POINTER(gc_acq_front_params), C.POINTER(gc_acq_front_result)]),
    "gc_acq_conditioned": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(C.c_float)]),
    "gc_acquire_fine_l1ca_batch": (C.c_int, [_P, C.POINTER(gc_acq_params), C.c_int, _P, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gc_if_format": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gc_acq_signal_from_record": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "gc_acq_set_signal": (C.c_int, [_P, C.POINTER(C.c_float), C.c_int64]),
    "gc_acq_signal_stats": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)]),
    "gc_acquire_fine_sums_batch": (C.c_int, [_P, C.POINTER(gc_fine_params), C.c_int, _P, C.POINTER(C.c_int64),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gc_acq_shift_prepare": (C.c_int, [_P, C.POINTER(gc_acq_shift_params)]),
    "gc_acq_shift_search": (C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "gc_acq_shift_row": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    "gc_acq_shift_search_batch": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                                            C.POINTER(gc_acq_shift_pick)]),
    "gc_acq_shift_dims": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gc_preamble_xcorr": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int64, _P, C.c_int, C.POINTER(C.c_float)]),
    "gc_sync_xcorr": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int64, _P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "gc_build_flags": (C.c_int, []),
    "gc_debug_first_sample_near_edge": (C.c_longlong, [C.c_double, C.c_double, C.c_longlong, C.c_double]),
    "gc_debug_last_kernel": (C.c_int, [_P]),
    "gc_debug_last_track_mode": (C.c_int, [_P]),
    "gc_debug_wave_transpose_sum": (C.c_int, [_P, C.c_int, _P, _P]),
    "gc_debug_tables_derivable": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "gc_debug_fft": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int]),
    "gc_acq_guard_stats": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Loads libgnsscorr.so (building is a separate, explicit step: cu_sdr_collection_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnssCorrError(GC_E_STATE, f"{LIB_PATH} not built — run `python -m cu_sdr_collection_amd.build` "
                                        "(hipcc, gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


GC_BUILD_TUNING = 1
TUNING_LIB_PATH = os.path.join(HERE, "lib", "libgnsscorr_tuning.so")


def is_tuning_build() -> bool:
    """True when the loaded library is libgnsscorr_tuning.so (GC_LIB_PATH): the GC_* switches of docs/KNOBS.md are read only there."""
    return bool(load().gc_build_flags() & GC_BUILD_TUNING)


def check(status: int):
    if status != GC_OK:
        msg = load().gc_last_error()
        raise GnssCorrError(status, msg.decode("utf-8", "replace") if msg else "")
