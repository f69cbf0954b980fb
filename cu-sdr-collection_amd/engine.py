"""Thin object wrapper over the C-ABI (one context = one GPU = one host thread)."""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib as L


def device_count() -> int:
    """gc_device_count: the HIP devices this process sees (valid Engine(device_id) are 0 .. n - 1); 0 without a GPU.  No context."""
    n = C.c_int(0)
    L.check(L.load().gc_device_count(C.byref(n)))
    return int(n.value)


class Engine:
    """Owns a gc_context on `device_id`.  Raises GnssCorrError when no MI355X is visible."""

    def __init__(self, device_id: int = 0):
        self._lib = L.load()
        self._ctx = C.c_void_p()
        L.check(self._lib.gc_create(C.byref(self._ctx), int(device_id)))
        self.device_id = device_id
        self.acq_stats = {}          # transforms of the searches since the caller last cleared it (_count_transforms)

    # ---- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.gc_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        name = C.create_string_buffer(128)
        cus = C.c_int()
        L.check(self._lib.gc_device_info(self._ctx, name, 128, C.byref(cus)))
        return name.value.decode(), cus.value

    def acq_guard_stats(self):
        """gc_acq_guard_stats of the last search: {ties, max_dev, eps} (include/gnsscorr.h)."""
        t, d, e = C.c_int32(), C.c_double(), C.c_double()
        L.check(self._lib.gc_acq_guard_stats(self._ctx, C.byref(t), C.byref(d), C.byref(e)))
        return {"ties": int(t.value), "max_dev": float(d.value), "eps": float(e.value)}

    def synchronize(self):
        L.check(self._lib.gc_synchronize(self._ctx))

    # ---- IF buffer -----------------------------------------------------------------------
    @staticmethod
    def _fmt(arr_dtype, layout):
        if arr_dtype == np.int8:
            dt = L.GC_I8
        elif arr_dtype == np.int16:
            dt = L.GC_I16
        else:
            raise TypeError("IF samples must be int8 or int16 (settings.dataType)")
        comp = 1 if layout == L.GC_REAL else 2
        return dt, comp

    def load_if(self, raw: np.ndarray, layout: int = L.GC_IQ, fs: float | None = None):
        """raw: the file content as a 1-D int8/int16 array (interleaved for IQ/QI)."""
        raw = np.ascontiguousarray(raw)
        dt, comp = self._fmt(raw.dtype, layout)
        n = raw.shape[0] // comp
        L.check(self._lib.gc_load_if(self._ctx, raw.ctypes.data_as(C.c_void_p), n, dt, layout))
        if fs is not None:
            self.set_sampling_freq(fs)

    def load_if_packed2(self, packed: np.ndarray, fs: float | None = None):
        """packed: uint8 array, two 2-bit sign-magnitude complex samples per byte (unpack_cplx.m's input format)."""
        b = np.ascontiguousarray(packed, dtype=np.uint8)
        L.check(self._lib.gc_load_if_packed2(self._ctx, b.ctypes.data_as(C.c_void_p), b.shape[0]))
        if fs is not None:
            self.set_sampling_freq(fs)

    def open_if_file(self, path: str, skip_bytes: int = 0, nsamples: int = 0, dtype=np.int8,
                     layout: int = L.GC_IQ, fs: float | None = None):
        dt, _ = self._fmt(np.dtype(dtype), layout)
        L.check(self._lib.gc_open_if_file(self._ctx, path.encode(), skip_bytes, nsamples, dt, layout))
        if fs is not None:
            self.set_sampling_freq(fs)

    def alloc_if(self, nsamples: int, dtype=np.int8, layout: int = L.GC_IQ):
        dt, _ = self._fmt(np.dtype(dtype), layout)
        L.check(self._lib.gc_alloc_if(self._ctx, nsamples, dt, layout))

    def attach_if(self, device_ptr: int, nsamples: int, dtype=np.int8, layout: int = L.GC_IQ):
        dt, _ = self._fmt(np.dtype(dtype), layout)
        L.check(self._lib.gc_attach_if(self._ctx, C.c_void_p(device_ptr), nsamples, dt, layout))

    def share_if(self, owner: "Engine"):
        """Read `owner`'s IF record from this context too (same GPU, no copy): gc_share_if."""
        L.check(self._lib.gc_share_if(self._ctx, owner._ctx))
        self._if_owner = owner  # keep the owner (and with it the device allocation) alive

    def if_buffer(self):
        p = C.c_void_p()
        n = C.c_uint64()
        L.check(self._lib.gc_if_buffer(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def if_format(self):
        """(dtype, layout) of the loaded record: GC_I8 / GC_I16, GC_IQ / GC_QI / GC_REAL."""
        dt, lay = C.c_int(), C.c_int()
        L.check(self._lib.gc_if_format(self._ctx, C.byref(dt), C.byref(lay)))
        return dt.value, lay.value

    def acq_input(self, first_sample: int, fs: float, n_long: int | None = None):
        """What an acquisition's searches read: (source, first_sample, samples available from there).  int8 I/Q records are read
        in place; any other format (int16 files, Q/I order, real samples) goes through a float copy of longSignal on the device
        (gc_acq_signal_from_record; n_long samples, by default what the record holds up to 0.3 s)."""
        dt, lay = self.if_format()
        avail = int(self.if_buffer()[1]) - int(first_sample)
        if dt == L.GC_I8 and lay == L.GC_IQ:
            return 0, int(first_sample), avail if n_long is None else min(avail, int(n_long))
        n = min(avail, int(n_long) if n_long is not None else int(0.3 * fs))
        L.check(self._lib.gc_acq_signal_from_record(self._ctx, int(first_sample), n))
        return 1, 0, n

    def acq_set_signal(self, x: np.ndarray):
        """gc_acq_set_signal: longSignal itself (any complex row) as the searches' source = 1."""
        z = np.ascontiguousarray(x, dtype=np.complex64)
        L.check(self._lib.gc_acq_set_signal(self._ctx, z.view(np.float32).ctypes.data_as(C.POINTER(C.c_float)), z.shape[0]))

    def read_if(self, first: int, n: int, dtype=np.int8, layout: int = L.GC_IQ) -> np.ndarray:
        comp = 1 if layout == L.GC_REAL else 2
        out = np.empty(n * comp, dtype=dtype)
        L.check(self._lib.gc_read_if(self._ctx, first, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def force_generic_kernel(self, on: bool):
        L.check(self._lib.gc_force_generic_kernel(self._ctx, int(bool(on))))

    def set_sampling_freq(self, fs: float):
        L.check(self._lib.gc_set_sampling_freq(self._ctx, float(fs)))

    # ---- code tables ---------------------------------------------------------------------
    def set_channel(self, channel: int, tables, index_scale: float = 1.0, arm_mult=None, windows=None):
        """tables: list of padded code tables ([c(end) c c(1)]), one per arm."""
        arms = len(tables)
        L.check(self._lib.gc_set_channel(self._ctx, channel, arms, float(index_scale)))
        for a, t in enumerate(tables):
            t8 = np.ascontiguousarray(np.asarray(t), dtype=np.int8)
            if not np.array_equal(t8, np.asarray(t)):
                raise ValueError("code tables must hold integers in {-1,0,+1}")
            m = 1.0 if arm_mult is None else float(arm_mult[a])
            L.check(self._lib.gc_set_code(self._ctx, channel, a, t8.ctypes.data_as(C.c_void_p), t8.shape[0], m))
            if windows is not None and windows[a]:
                L.check(self._lib.gc_set_code_window(self._ctx, channel, a, int(windows[a])))

    # ---- correlator ----------------------------------------------------------------------
    @staticmethod
    def make_blocks(n: int):
        return (L.gc_block * n)()

    def correlate(self, blocks) -> np.ndarray:
        """blocks: ctypes array of gc_block.  Returns float64 [nblocks, GC_MAX_ARMS, 6]."""
        n = len(blocks)
        out = np.zeros((n, L.GC_MAX_ARMS, 6))
        L.check(self._lib.gc_correlate(self._ctx, n, blocks, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def last_kernel(self) -> int:
        """gc_debug_last_kernel: 0 lane, 1 fast (one wave), 2 fast (four waves, int8 pairs), 3 fast (four waves, floats), 4 multi-transition
        (corr_multi.hip), -1 mixed."""
        return int(self._lib.gc_debug_last_kernel(self._ctx))

    def last_track_mode(self) -> int:
        """gc_debug_last_track_mode: 0 a launch per epoch, 1 persistent host-fed kernel, 2 device loop."""
        return int(self._lib.gc_debug_last_track_mode(self._ctx))

    def debug_wave_transpose_sum(self, values: np.ndarray) -> np.ndarray:
        """gc_debug_wave_transpose_sum: values [k, 64] float32 -> the k wave sums as the lane kernel's flush forms them."""
        v = np.ascontiguousarray(values, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != 64:
            raise ValueError("debug_wave_transpose_sum: [k, 64] values")
        out = np.empty(v.shape[0], dtype=np.float32)
        L.check(self._lib.gc_debug_wave_transpose_sum(self._ctx, v.shape[0], v.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def replay_prepare(self, blocks):
        self._replay_n = len(blocks)
        L.check(self._lib.gc_replay_prepare(self._ctx, len(blocks), blocks))

    def replay_launch(self):
        L.check(self._lib.gc_replay_launch(self._ctx))

    def replay_fetch(self) -> np.ndarray:
        out = np.zeros((self._replay_n, L.GC_MAX_ARMS, 6))
        L.check(self._lib.gc_replay_fetch(self._ctx, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def timer_start(self):
        L.check(self._lib.gc_timer_start(self._ctx))

    def timer_stop(self) -> float:
        ms = C.c_double()
        L.check(self._lib.gc_timer_stop(self._ctx, C.byref(ms)))
        return ms.value

    # ---- closed-loop tracking ------------------------------------------------------------
    def track(self, params: L.gc_track_params, inits, device_loop: bool = False):
        """Runs gc_track (or gc_track_device: loop closed on the GPU, one persistent launch).
        Returns (fields dict name -> [nch, n_epochs], epochs_done, status)."""
        nch = len(inits)
        arr = (L.gc_channel_init * nch)(*inits)
        n_ep = params.n_epochs
        out = np.zeros((nch, L.GC_TRK_NFIELDS, n_ep))
        done = (C.c_int32 * nch)()
        fn = self._lib.gc_track_device if device_loop else self._lib.gc_track
        cno = self._cno_buffer(params, nch)
        try:
            st = fn(self._ctx, C.byref(params), nch, arr,
                                    out.ctypes.data_as(C.POINTER(C.c_double)), done)
        finally:
            if cno is not None:
                self._lib.gc_set_cno_output(self._ctx, None, 0)
        if st not in (L.GC_OK, L.GC_E_RANGE):
            L.check(st)
        fields = {name: out[:, i, :] for i, name in enumerate(L.TRK_FIELDS)}
        if cno is not None:
            # [nch, n_epochs // cno_interval], 0 where an interval was not completed; Calc_CNo_PLD modes: [nch, nk, GC_CNO_NPLD]
            fields["CNoVSM" if cno.ndim == 2 else "CNoPLD"] = cno
        return fields, np.array(list(done)), st

    def _cno_buffer(self, params, nch):
        """Registers the C/N0 output of the next tracking call when the parameters ask for it (gc_set_cno_output)."""
        k = int(params.cno_interval)
        if k <= 1 or params.n_epochs // k == 0:
            return None
        cno = np.zeros((nch, params.n_epochs // k) + ((L.GC_CNO_NPLD,) if params.cno_mode != L.GC_CNO_VSM else ()))
        L.check(self._lib.gc_set_cno_output(self._ctx, cno.ctypes.data_as(C.POINTER(C.c_double)), cno.size))
        return cno

    def track_resume(self, params: L.gc_track_params, inits, state=None, origin: int = 0, pause_at_end: bool = False):
        """gc_track_resume on the window currently loaded (its first sample = record sample `origin`): continues from `state`
        (a ctypes array of gc_channel_state from a previous call; None starts from `inits`).
        Returns (fields, epochs_done, status, state, paused)."""
        nch = len(inits)
        arr = (L.gc_channel_init * nch)(*inits)
        flags = (1 if state is not None else 0) | (2 if pause_at_end else 0)
        if state is None:
            state = (L.gc_channel_state * nch)()
        out = np.zeros((nch, L.GC_TRK_NFIELDS, params.n_epochs))
        done = (C.c_int32 * nch)()
        paused = C.c_int32(0)
        st = self._lib.gc_track_resume(self._ctx, C.byref(params), nch, arr, state, flags, int(origin),
                                       out.ctypes.data_as(C.POINTER(C.c_double)), done, C.byref(paused))
        if st not in (L.GC_OK, L.GC_E_RANGE):
            L.check(st)
        return {name: out[:, i, :] for i, name in enumerate(L.TRK_FIELDS)}, np.array(list(done)), st, state, bool(paused.value)

    def track_file(self, path: str, params: L.gc_track_params, inits, window_samples: int, dtype: int = L.GC_I8,
                   layout: int = L.GC_IQ, skip_bytes: int = 0):
        """gc_track_file: tracking(fid, channel, settings) on a file of any size, at most 2 * window_samples samples resident
        (the next window is read and uploaded while the current one is tracked).  Returns as track()."""
        nch = len(inits)
        arr = (L.gc_channel_init * nch)(*inits)
        out = np.zeros((nch, L.GC_TRK_NFIELDS, params.n_epochs))
        done = (C.c_int32 * nch)()
        cno = self._cno_buffer(params, nch)
        try:
            st = self._lib.gc_track_file(self._ctx, os.fsencode(path), int(skip_bytes), int(dtype), int(layout), int(window_samples),
                                         C.byref(params), nch, arr, out.ctypes.data_as(C.POINTER(C.c_double)), done)
        finally:
            if cno is not None:
                self._lib.gc_set_cno_output(self._ctx, None, 0)
        if st not in (L.GC_OK, L.GC_E_RANGE):
            L.check(st)
        fields = {name: out[:, i, :] for i, name in enumerate(L.TRK_FIELDS)}
        if cno is not None:
            fields["CNoVSM" if cno.ndim == 2 else "CNoPLD"] = cno
        return fields, np.array(list(done)), st

    @staticmethod
    def track_multi(jobs, device_loop: bool = False):
        """gc_track_multi: jobs = [(engine, gc_track_params, [gc_channel_init, ...]), ...], one engine (context) per job,
        all tracking loops run concurrently.  Returns [(fields, epochs_done, status), ...] in job order."""
        lib = L.load()
        n = len(jobs)
        arr = (L.gc_track_job * n)()
        keep = []
        for k, (eng, params, inits) in enumerate(jobs):
            nch = len(inits)
            ia = (L.gc_channel_init * nch)(*inits)
            out = np.zeros((nch, L.GC_TRK_NFIELDS, params.n_epochs))
            done = (C.c_int32 * nch)()
            keep.append((ia, out, done, params, eng._cno_buffer(params, nch), eng))
            arr[k].ctx = eng._ctx
            arr[k].params = C.pointer(params)
            arr[k].init = ia
            arr[k].out = out.ctypes.data_as(C.POINTER(C.c_double))
            arr[k].epochs_done = done
            arr[k].nch = nch
            arr[k].device_loop = int(bool(device_loop))
        try:
            st = lib.gc_track_multi(n, arr)
        finally:
            for item in keep:
                if item[4] is not None:
                    lib.gc_set_cno_output(item[5]._ctx, None, 0)
        res = []
        for k, (ia, out, done, params, cno, _) in enumerate(keep):
            if arr[k].status not in (L.GC_OK, L.GC_E_RANGE):
                raise L.GnssCorrError(arr[k].status, f"job {k}: " + arr[k].error.decode("utf-8", "replace"))
            fields = {name: out[:, i, :] for i, name in enumerate(L.TRK_FIELDS)}
            if cno is not None:
                fields["CNoVSM" if cno.ndim == 2 else "CNoPLD"] = cno
            res.append((fields, np.array(list(done)), int(arr[k].status)))
        if st not in (L.GC_OK, L.GC_E_RANGE):
            L.check(st)
        return res

    # ---- acquisition ---------------------------------------------------------------------
    def acquire_coarse(self, params: L.gc_acq_params, sampled_codes: np.ndarray, freq_offset=None):
        """sampled_codes: int8 [nprn, spc], or [nprn, narms, spc] for a data+pilot search.  freq_offset: Hz per row added to
        params.intermediate_freq (gc_acquire_coarse_offsets: GLONASS' frequency numbers in one call); raises GnssCorrError with status
        GC_E_UNSUPPORTED when an offset is not a whole number of the search's FFT bins."""
        codes = np.ascontiguousarray(sampled_codes, dtype=np.int8)
        nprn = codes.shape[0]
        narms = codes.shape[1] if codes.ndim == 3 else 1
        res = (L.gc_acq_result * nprn)()
        if freq_offset is None:
            L.check(self._lib.gc_acquire_coarse_multi(self._ctx, C.byref(params), nprn, narms,
                                                      codes.ctypes.data_as(C.c_void_p), res))
        else:
            off = np.ascontiguousarray(freq_offset, dtype=np.float64)
            if off.shape != (nprn,):
                raise ValueError("acquire_coarse: one frequency offset per row")
            L.check(self._lib.gc_acquire_coarse_offsets(self._ctx, C.byref(params), nprn, narms, codes.ctypes.data_as(C.c_void_p),
                                                        off.ctypes.data_as(C.POINTER(C.c_double)), res))
        # bench bookkeeping, after the library has accepted the parameters (it must never change the call's error behaviour)
        n = int(params.block_len) if params.block_len else 2 * codes.shape[-1]
        bins = int(params.n_bins) if params.n_bins else int(math.floor(params.search_band * 2 / params.search_step + 0.5)) + 1
        self._count_transforms(n, forward=bins * int(params.non_coh_time), code=nprn * narms, inverse=nprn * narms * bins * int(params.non_coh_time))
        return list(res)

    def acquire_fine_sums(self, params: L.gc_fine_params, code: np.ndarray) -> np.ndarray:
        """Per-code-period complex sums [nbins, ncodes] of the generic fine-frequency stage."""
        c8 = np.ascontiguousarray(code, dtype=np.int8)
        out = np.empty((params.nbins, params.ncodes, 2))
        L.check(self._lib.gc_acquire_fine_sums(self._ctx, C.byref(params), c8.ctypes.data_as(C.c_void_p),
                                               out.ctypes.data_as(C.POINTER(C.c_double))))
        return out[..., 0] + 1j * out[..., 1]

    def acq_condition(self, sampling_freq: float, intermediate_freq: float, bandwidth: float, first_sample: int, n_samples: int,
                      fir_order: int = 700, band_margin: float = 0.0):
        """gc_acq_condition: the zero-phase band-pass + decimation front end of acquisition.m:46-111 on the record's samples
        [first_sample, first_sample + n_samples).  Returns (new sampling frequency, new IF, conditioned length); searches read
        the conditioned signal with gc_acq_params.source = 1."""
        fp = L.gc_acq_front_params(sampling_freq=sampling_freq, intermediate_freq=intermediate_freq, bandwidth=bandwidth,
                                   first_sample=int(first_sample), n_samples=int(n_samples), fir_order=int(fir_order),
                                   band_margin=float(band_margin))
        res = L.gc_acq_front_result()
        L.check(self._lib.gc_acq_condition(self._ctx, C.byref(fp), C.byref(res)))
        return res.sampling_freq, res.intermediate_freq, int(res.n_samples)

    def acq_conditioned(self, first: int, n: int) -> np.ndarray:
        """The conditioned signal back as complex64 (test hook)."""
        out = np.empty(2 * int(n), dtype=np.float32)
        L.check(self._lib.gc_acq_conditioned(self._ctx, int(first), int(n), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out[0::2] + 1j * out[1::2]

    def acquire_fine_sums_batch(self, params: L.gc_fine_params, codes: np.ndarray, first_sample, f0) -> np.ndarray:
        """The same for several detections in one launch: codes [ndet, code_len], first_sample [ndet], f0 [ndet]
        -> complex [ndet, nbins, ncodes]."""
        c8 = np.ascontiguousarray(codes, dtype=np.int8)
        ndet = c8.shape[0]
        fs = np.ascontiguousarray(first_sample, dtype=np.int64)
        ff = np.ascontiguousarray(f0, dtype=np.float64)
        if c8.ndim != 2 or c8.shape[1] != params.code_len or fs.shape != (ndet,) or ff.shape != (ndet,):
            raise ValueError("acquire_fine_sums_batch: codes [ndet, code_len], first_sample [ndet], f0 [ndet]")
        out = np.empty((ndet, params.nbins, params.ncodes, 2))
        L.check(self._lib.gc_acquire_fine_sums_batch(self._ctx, C.byref(params), ndet, c8.ctypes.data_as(C.c_void_p),
                                                     fs.ctypes.data_as(C.POINTER(C.c_int64)), ff.ctypes.data_as(C.POINTER(C.c_double)),
                                                     out.ctypes.data_as(C.POINTER(C.c_double))))
        return out[..., 0] + 1j * out[..., 1]

    def acq_signal_stats(self, first_sample: int, n: int, source: int = 0):
        """gc_acq_signal_stats: (mean(x), var(x)) of n samples of the record (or of the conditioned signal) as MATLAB's mean / var."""
        mr, mi, v = C.c_double(), C.c_double(), C.c_double()
        L.check(self._lib.gc_acq_signal_stats(self._ctx, int(first_sample), int(n), int(source), C.byref(mr), C.byref(mi), C.byref(v)))
        return complex(mr.value, mi.value), v.value

    def _count_transforms(self, n: int, forward: int = 0, code: int = 0, inverse: int = 0):
        """Bookkeeping for bench.py (acq_stats): the transforms the SEARCH needs once the signal spectra are hoisted out of the PRN
        loop (acquisition.m:167-192 recomputes them per PRN) - n-point forward transforms of the signal, of the sampled codes,
        and inverse transforms, since the last reset.  Counts only; nothing on the device depends on it."""
        st = self.acq_stats
        st["n_fft"] = max(st.get("n_fft", 0), int(n))
        for k, v in (("forward", forward), ("code", code), ("inverse", inverse)):
            st[k] = st.get(k, 0) + int(v)

    def acq_shift_prepare(self, params: L.gc_acq_shift_params):
        self._count_transforms(int(params.n), forward=int(params.n_signals) * int(params.n_carriers))
        self._shift = params
        L.check(self._lib.gc_acq_shift_prepare(self._ctx, C.byref(params)))

    def acq_shift_search(self, codes: np.ndarray, arm_weight=None):
        """codes: int8 [narms, n].  Returns (row_max float32[rows], row_argmax int32[rows])."""
        p = self._shift
        c8 = np.ascontiguousarray(codes, dtype=np.int8).reshape(-1, p.n)
        rows = p.n_carriers * p.n_signals * p.n_bins
        self._count_transforms(int(p.n), code=c8.shape[0], inverse=c8.shape[0] * rows)
        rmax = np.empty(rows, dtype=np.float32)
        rarg = np.empty(rows, dtype=np.int32)
        w = None
        if arm_weight is not None:
            w = (C.c_double * c8.shape[0])(*[float(v) for v in arm_weight])
        L.check(self._lib.gc_acq_shift_search(self._ctx, c8.shape[0], c8.ctypes.data_as(C.c_void_p), w,
                                              rmax.ctypes.data_as(C.POINTER(C.c_float)), rarg.ctypes.data_as(C.POINTER(C.c_int32))))
        return rmax, rarg

    def acq_shift_search_batch(self, codes: np.ndarray, arm_weight, rule: int, exclude: int = 0, period: int = 1, sample_index=None):
        """gc_acq_shift_search_batch: codes int8 [nprn, narms, n] (sampled replicas) or, with sample_index (0-based, one vector for all
        codes), [nprn, narms, chips] -> ctypes array of gc_acq_shift_pick [nprn], or None when the library answers GC_E_NOMEM (the batch's buffers
        do not fit; GC_E_UNSUPPORTED is kept for older libraries): the caller searches PRN by PRN."""
        p = self._shift
        c8 = np.ascontiguousarray(codes, dtype=np.int8)
        nprn, narms = c8.shape[0], c8.shape[1]
        rows = p.n_carriers * p.n_signals * p.n_bins
        picks = (L.gc_acq_shift_pick * nprn)()
        w = None
        if arm_weight is not None:
            w = (C.c_double * narms)(*[float(v) for v in arm_weight])
        idx, nidx = None, 0
        if sample_index is not None:
            i32 = np.ascontiguousarray(sample_index, dtype=np.int32)
            idx, nidx = i32.ctypes.data_as(C.c_void_p), int(i32.shape[0])
        rc = self._lib.gc_acq_shift_search_batch(self._ctx, nprn, narms, c8.ctypes.data_as(C.c_void_p), int(c8.shape[2]), idx, nidx, w, int(rule),
                                                 int(exclude), int(period), picks)
        if rc in (L.GC_E_UNSUPPORTED, L.GC_E_NOMEM):
            # GC_E_NOMEM: the batch holds every PRN's code spectra (nprn x narms x N x 8 bytes) and winning rows at once; the PRN-by-PRN
            # path needs one PRN's and still fits on a device that is short of memory (next to gc_track_multi contexts)
            return None
        L.check(rc)
        self._count_transforms(int(p.n), code=nprn * narms,
                               inverse=nprn * narms * rows + (narms * sum(1 for k in picks if k.row >= 0) if rule != L.GC_SHIFT_PICK_GLOBAL else 0))
        return picks

    def acq_shift_row(self, row: int) -> np.ndarray:
        out = np.empty(self._shift.n, dtype=np.float32)
        L.check(self._lib.gc_acq_shift_row(self._ctx, int(row), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def sync_xcorr(self, i_p: np.ndarray, pattern: np.ndarray, zero_is_plus: bool = False) -> np.ndarray:
        """gc_sync_xcorr: the non-negative lags of xcorr(hard-limited I_P, pattern) (NAVdecoding.m of every package)."""
        x = np.ascontiguousarray(i_p, dtype=np.float64)
        pat = np.ascontiguousarray(pattern, dtype=np.int8)
        out = np.empty(x.shape[0], dtype=np.float32)
        L.check(self._lib.gc_sync_xcorr(self._ctx, x.ctypes.data_as(C.POINTER(C.c_double)), x.shape[0], pat.ctypes.data_as(C.c_void_p),
                                        pat.shape[0], L.GC_SYNC_ZERO_IS_PLUS if zero_is_plus else 0, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def preamble_xcorr(self, i_p: np.ndarray, pattern: np.ndarray) -> np.ndarray:
        return self.sync_xcorr(i_p, pattern)

    def debug_fft(self, x: np.ndarray, inverse: bool = False) -> np.ndarray:
        """x: complex64 [nbatch, n].  The library's FFT (test hook)."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.empty_like(x)
        L.check(self._lib.gc_debug_fft(self._ctx, x.shape[1], x.shape[0], x.ctypes.data_as(C.c_void_p),
                                       out.ctypes.data_as(C.c_void_p), int(inverse)))
        return out

    def acquire_fine_l1ca(self, params: L.gc_acq_params, code: np.ndarray, code_phase: int,
                          coarse_freq: float) -> float:
        c8 = np.ascontiguousarray(code, dtype=np.int8)
        f = C.c_double()
        L.check(self._lib.gc_acquire_fine_l1ca(self._ctx, C.byref(params), c8.ctypes.data_as(C.c_void_p),
                                               int(code_phase), float(coarse_freq), C.byref(f)))
        return f.value

    def acquire_fine_l1ca_batch(self, params: L.gc_acq_params, codes: np.ndarray, code_phase, coarse_freq) -> np.ndarray:
        """Fine stage of all detected PRNs in one launch: codes [ndet, code_length] -> carrFreq [ndet]."""
        c8 = np.ascontiguousarray(codes, dtype=np.int8)
        ndet = c8.shape[0]
        cp = np.ascontiguousarray(code_phase, dtype=np.int32)
        cf = np.ascontiguousarray(coarse_freq, dtype=np.float64)
        if c8.ndim != 2 or c8.shape[1] != int(params.code_length) or cp.shape != (ndet,) or cf.shape != (ndet,):
            raise ValueError("acquire_fine_l1ca_batch: codes [ndet, code_length], code_phase [ndet], coarse_freq [ndet]")
        out = np.empty(ndet)
        L.check(self._lib.gc_acquire_fine_l1ca_batch(self._ctx, C.byref(params), ndet, c8.ctypes.data_as(C.c_void_p),
                                                     cp.ctypes.data_as(C.POINTER(C.c_int32)), cf.ctypes.data_as(C.POINTER(C.c_double)),
                                                     out.ctypes.data_as(C.POINTER(C.c_double))))
        return out
