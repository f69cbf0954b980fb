"""Synthetic IF record generator (NumPy) — stands in for the externally hosted data sets the
reference's README points to (README.md:10-11; there is no network here).

Signal model per satellite k, sample n (t = n/fs):
    x_k[n] = A_k * D_k(m) * c_k(chip) * exp(+j*(2*pi*(IF + fd_k)*t + phi_k))
    chip phase = (n - n0_k) * (fc + fd_k/1540 * fc/1.023e6 ...) / fs   (code Doppler = Doppler/1540 for L1)
    D_k flips every 20 code periods, aligned to the code epoch (50 bps)
plus complex AWGN of per-component sigma, rounded and clipped to int8, interleaved I,Q
(settings.fileType = 2, dataType 'schar').  A_k from C/N0: A = sigma*sqrt(2*10^(CN0/10)/fs).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class SatSpec:
    prn: int
    doppler: float           # Hz
    code_phase_samples: float  # sample index (0-based, may be fractional) at which a code period starts
    carrier_phase: float     # rad at n = 0
    cn0_dbhz: float = 45.0


def scene(n_sats: int, seed: int, fs: float, cn0=45.0, doppler_max=5e3, prn_pool=range(1, 33)):
    """Config-2 style scene: PRNs without replacement, Doppler U(-5,5) kHz, code phase U[0,1 ms)."""
    rng = np.random.default_rng(seed)
    prns = rng.choice(np.array(list(prn_pool)), size=n_sats, replace=False)
    sats = []
    for p in prns:
        sats.append(SatSpec(prn=int(p), doppler=float(rng.uniform(-doppler_max, doppler_max)),
                            code_phase_samples=float(rng.uniform(0, fs * 1e-3)),
                            carrier_phase=float(rng.uniform(0, 2 * np.pi)), cn0_dbhz=cn0))
    return sats


def generate_if(sats, n_samples: int, fs: float, intermediate_freq: float, code_fn, code_rate: float,
                code_len: int, seed: int, sigma: float = 20.0, carrier_ratio: float = 1540.0,
                bit_periods: int = 20, chunk: int = 1 << 21, noise: bool = True, pilot_fn=None,
                pilot_phase: float = 0.0) -> np.ndarray:
    """Returns int8[2*n_samples] interleaved I,Q.

    code_fn(prn) -> +-1 chips (length code_len).  carrier_ratio = f_carrier/f_chip * (chip-rate
    units): code Doppler = doppler / carrier_ratio * (code_rate / 1.023e6)."""
    rng = np.random.default_rng(seed)
    out = np.empty(2 * n_samples, dtype=np.int8)
    codes = {s.prn: np.asarray(code_fn(s.prn), dtype=np.float64) for s in sats}
    # optional pilot component (no data bits) at carrier phase offset `pilot_phase` (rad):
    # 0 for Galileo E1-C next to E1-B, +pi/2 for GPS L5-Q next to L5-I
    pilots = {s.prn: np.asarray(pilot_fn(s.prn), dtype=np.float64) for s in sats} if pilot_fn else None
    bits = {s.prn: rng.integers(0, 2, size=int(n_samples / fs * code_rate / code_len / bit_periods) + 8) * 2.0 - 1.0
            for s in sats}
    for start in range(0, n_samples, chunk):
        n = np.arange(start, min(n_samples, start + chunk), dtype=np.float64)
        acc = np.zeros(n.shape[0], dtype=np.complex128)
        for s in sats:
            amp = sigma * np.sqrt(2.0 * 10 ** (s.cn0_dbhz / 10.0) / fs)
            fcode = code_rate + s.doppler / carrier_ratio * (code_rate / 1.023e6)
            cp = (n - s.code_phase_samples) * (fcode / fs)      # chips since the reference code start
            chip = np.floor(cp).astype(np.int64)
            period = np.floor_divide(chip, code_len)
            bit_idx = np.floor_divide(period, bit_periods)
            b = bits[s.prn]
            data = b[np.mod(bit_idx, b.shape[0])]
            c = codes[s.prn][np.mod(chip, code_len)]
            theta = 2 * np.pi * (intermediate_freq + s.doppler) * (n / fs) + s.carrier_phase
            comp = data * c
            if pilots is not None:
                comp = comp + pilots[s.prn][np.mod(chip, code_len)] * np.exp(1j * pilot_phase)
            acc += amp * comp * np.exp(1j * theta)
        if noise:
            acc += sigma * (rng.standard_normal(n.shape[0]) + 1j * rng.standard_normal(n.shape[0]))
        i = np.clip(np.rint(acc.real), -127, 127).astype(np.int8)
        q = np.clip(np.rint(acc.imag), -127, 127).astype(np.int8)
        sl = slice(2 * start, 2 * (start + n.shape[0]))
        out[sl][0::2] = i
        out[sl][1::2] = q
    return out


@dataclass
class SignalGroup:
    """One signal family inside a record: the satellites that transmit it and how it is spread."""
    sats: list
    code_fn: object            # prn -> +-1 (or 0) chips, length code_len
    code_rate: float           # chips per second of that table (2 x 1.023e6 for BOC(1,1) half-chip tables)
    code_len: int
    bit_periods: int = 20      # code periods per data bit
    pilot_fn: object = None    # optional pilot component (no data bits)
    pilot_phase: float = 0.0   # carrier phase offset of the pilot (rad): 0 Galileo E1-C, +pi/2 GPS L5-Q
    carrier_ratio: float = 1540.0   # code Doppler = doppler / carrier_ratio * (code_rate / 1.023e6)
    intermediate_freq: float | None = None   # None: the record's IF


def _group_signal(g: SignalGroup, n: np.ndarray, fs: float, intermediate_freq: float, sigma: float, rng) -> np.ndarray:
    """Noise-free complex baseband-at-IF contribution of one signal family at sample indices n."""
    acc = np.zeros(n.shape[0], dtype=np.complex128)
    f_if = intermediate_freq if g.intermediate_freq is None else g.intermediate_freq
    for s in g.sats:
        key = (id(g), s.prn)
        if key not in rng["codes"]:
            rng["codes"][key] = np.asarray(g.code_fn(s.prn), dtype=np.float64)
            rng["pilots"][key] = np.asarray(g.pilot_fn(s.prn), dtype=np.float64) if g.pilot_fn else None
            rng["bits"][key] = rng["rng"].integers(0, 2, size=int(rng["n_total"] / fs * g.code_rate / g.code_len / g.bit_periods) + 8) * 2.0 - 1.0
        amp = sigma * np.sqrt(2.0 * 10 ** (s.cn0_dbhz / 10.0) / fs)
        fcode = g.code_rate + s.doppler / g.carrier_ratio * (g.code_rate / 1.023e6)
        cp = (n - s.code_phase_samples) * (fcode / fs)      # chips since the reference code start
        chip = np.floor(cp).astype(np.int64)
        period = np.floor_divide(chip, g.code_len)
        bit_idx = np.floor_divide(period, g.bit_periods)
        b = rng["bits"][key]
        data = b[np.mod(bit_idx, b.shape[0])]
        comp = data * rng["codes"][key][np.mod(chip, g.code_len)]
        theta = 2 * np.pi * (f_if + s.doppler) * (n / fs) + s.carrier_phase
        if rng["pilots"][key] is not None:
            comp = comp + rng["pilots"][key][np.mod(chip, g.code_len)] * np.exp(1j * g.pilot_phase)
        acc += amp * comp * np.exp(1j * theta)
    return acc


def generate_if_mix(groups, n_samples: int, fs: float, intermediate_freq: float, seed: int, sigma: float = 20.0,
                    chunk: int = 1 << 21, noise: bool = True) -> np.ndarray:
    """A record holding several signal families at once (BASELINE config 5: GPS L1 C/A, Galileo E1 and BDS B1C share the
    L1 band).  Returns int8[2*n_samples] interleaved I,Q."""
    r = np.random.default_rng(seed)
    state = {"rng": r, "codes": {}, "pilots": {}, "bits": {}, "n_total": n_samples}
    out = np.empty(2 * n_samples, dtype=np.int8)
    for start in range(0, n_samples, chunk):
        n = np.arange(start, min(n_samples, start + chunk), dtype=np.float64)
        acc = np.zeros(n.shape[0], dtype=np.complex128)
        for g in groups:
            acc += _group_signal(g, n, fs, intermediate_freq, sigma, state)
        if noise:
            acc += sigma * (r.standard_normal(n.shape[0]) + 1j * r.standard_normal(n.shape[0]))
        sl = slice(2 * start, 2 * (start + n.shape[0]))
        out[sl][0::2] = np.clip(np.rint(acc.real), -127, 127).astype(np.int8)
        out[sl][1::2] = np.clip(np.rint(acc.imag), -127, 127).astype(np.int8)
    return out


# ---------------------------------------------------------------------------------------------
# GPU generator (libgnsssynth.so, csrc/synth.hip) — same signal model, counter-based noise
# ---------------------------------------------------------------------------------------------
def generate_if_gpu(engine, sats, n_samples: int, fs: float, intermediate_freq: float, code_fn,
                    code_rate: float, code_len: int, seed: int, sigma: float = 20.0,
                    carrier_ratio: float = 1540.0, bit_periods: int = 20, attached: bool = False) -> None:
    """Allocates the engine's IF buffer (int8 I/Q) and fills it on the GPU.  attached=True: the engine already reads a buffer
    of n_samples int8 I/Q samples that the caller owns (Engine.attach_if, e.g. a torch tensor about to be broadcast) - fill that."""
    import ctypes as C
    import os

    class gs_sat(C.Structure):
        _fields_ = [("prn", C.c_int32), ("code_index", C.c_int32), ("doppler", C.c_double),
                    ("code_phase_samples", C.c_double), ("carrier_phase", C.c_double),
                    ("amplitude", C.c_double), ("code_rate", C.c_double)]

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgnsssynth.so")
    lib = C.CDLL(path)
    lib.gs_generate.restype = C.c_int
    lib.gs_generate.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int,
                                C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_uint64]
    if not attached:
        engine.alloc_if(n_samples, np.int8)
    engine.synchronize()
    ptr, n = engine.if_buffer()
    if n != n_samples:
        raise ValueError(f"generate_if_gpu: the engine's buffer holds {n} samples, {n_samples} wanted")
    codes = np.ascontiguousarray(np.stack([np.asarray(code_fn(s.prn), dtype=np.int8) for s in sats]))
    arr = (gs_sat * len(sats))()
    for i, s in enumerate(sats):
        arr[i].prn = s.prn
        arr[i].code_index = i
        arr[i].doppler = s.doppler
        arr[i].code_phase_samples = s.code_phase_samples
        arr[i].carrier_phase = s.carrier_phase
        arr[i].amplitude = sigma * np.sqrt(2.0 * 10 ** (s.cn0_dbhz / 10.0) / fs)
        arr[i].code_rate = code_rate + s.doppler / carrier_ratio * (code_rate / 1.023e6)
    rc = lib.gs_generate(C.c_void_p(ptr), n_samples, engine.device_id, fs, intermediate_freq,
                         codes.ctypes.data_as(C.c_void_p), len(sats), code_len, bit_periods, arr, len(sats),
                         sigma, seed)
    if rc != 0:
        raise RuntimeError(f"gs_generate failed with {rc}")


def generate_if_mix_gpu(engine, groups, n_samples: int, fs: float, intermediate_freq: float, seed: int, sigma: float = 20.0,
                        dtype=np.int8, qi_order: bool = False, attached: bool = False) -> None:
    """generate_if_mix on the GPU (libgnsssynth.so gs_generate2): allocates the engine's IF buffer (int8 or int16 I/Q)
    and fills it in HBM.  Same signal model; data bits and noise come from a counter-based hash instead of NumPy's
    generator, so the two generators agree in distribution, not sample for sample.  attached=True: fill the buffer the engine
    already reads (Engine.attach_if on caller-owned memory of the same size and format) instead of allocating one."""
    import ctypes as C
    import os

    class gs_sat2(C.Structure):
        _fields_ = [("prn", C.c_int32), ("code_len", C.c_int32), ("code_offset", C.c_int32), ("pilot_offset", C.c_int32),
                    ("bit_periods", C.c_int32), ("reserved", C.c_int32), ("doppler", C.c_double),
                    ("code_phase_samples", C.c_double), ("carrier_phase", C.c_double), ("amplitude", C.c_double),
                    ("code_rate", C.c_double), ("pilot_phase", C.c_double), ("intermediate_freq", C.c_double)]

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgnsssynth.so")
    lib = C.CDLL(path)
    lib.gs_generate2.restype = C.c_int
    lib.gs_generate2.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int,
                                 C.c_double, C.c_uint64, C.c_int]
    i16 = np.dtype(dtype) == np.int16
    from . import _lib as L
    if not attached:
        engine.alloc_if(n_samples, np.int16 if i16 else np.int8, layout=L.GC_QI if qi_order else L.GC_IQ)
    engine.synchronize()
    ptr, have = engine.if_buffer()
    if have != n_samples:
        raise ValueError(f"generate_if_mix_gpu: the engine's buffer holds {have} samples, {n_samples} wanted")
    flat, sats = [], []
    off = 0
    for g in groups:
        for s in g.sats:
            c = np.asarray(g.code_fn(s.prn), dtype=np.int8)
            assert c.shape[0] == g.code_len
            flat.append(c)
            e = gs_sat2(prn=s.prn, code_len=g.code_len, code_offset=off, pilot_offset=-1, bit_periods=g.bit_periods)
            off += g.code_len
            if g.pilot_fn is not None:
                flat.append(np.asarray(g.pilot_fn(s.prn), dtype=np.int8))
                e.pilot_offset = off
                off += g.code_len
            e.doppler, e.code_phase_samples, e.carrier_phase = s.doppler, s.code_phase_samples, s.carrier_phase
            e.amplitude = sigma * np.sqrt(2.0 * 10 ** (s.cn0_dbhz / 10.0) / fs)
            e.code_rate = g.code_rate + s.doppler / g.carrier_ratio * (g.code_rate / 1.023e6)
            e.pilot_phase = g.pilot_phase
            e.intermediate_freq = intermediate_freq if g.intermediate_freq is None else g.intermediate_freq
            sats.append(e)
    codes = np.ascontiguousarray(np.concatenate(flat))
    arr = (gs_sat2 * len(sats))(*sats)
    rc = lib.gs_generate2(C.c_void_p(ptr), n_samples, engine.device_id, fs, codes.ctypes.data_as(C.c_void_p), codes.shape[0],
                          arr, len(sats), sigma, seed, int(i16) | (2 if qi_order else 0))
    if rc != 0:
        raise RuntimeError(f"gs_generate2 failed with {rc}")
