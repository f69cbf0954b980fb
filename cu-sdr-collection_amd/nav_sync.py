"""Bit / frame synchronisation ahead of navigation decoding: what every package's NAVdecoding.m does first with a channel's
prompt in-phase stream.  The cross-correlation of the hard-limited stream with the package's sync
pattern runs on the GPU (gc_sync_xcorr); what follows it is restated here, package by package, as a table:

  package    pattern (samples)                          detection on the non-negative lags     candidates            file:lines
  GPS_L1CA   8-bit TLM preamble x 20           (160)    |r| > 153                              one 6000 ms later     GPS/GPS_L1CA/include/NAVdecoding.m:66-145
  GAL_E1C    10 sync symbols, 1 per symbol     (10)     round(|r|) >= 9.99, bits = (I_P < 0)   250 AND 500 later     GAL/GAL_E1C/include/NAVdecoding.m:59,79-108
  GAL_E5a    12 sync symbols x CS20            (240)    round(|r|) >= 239.99                   one 10 000 away       GAL/GAL_E5a/include/NAVdecoding.m:54,69-108
  GAL_E5b    10 preamble symbols x [-1 -1 -1 1] (40)    |r| > 39.99                            one 1000 later        GAL/GAL_E5b/include/NAVdecoding.m:59,80-119
  BDS_B1I    11-bit preamble x -NH20 (220) / x 2 (22)   |r| >= 10 codrPerD, offset 1000        300 codrPerD later    BDS/B1I/include/NAVdecoding.m:68-170
  BDS_B3I    as B1I, GEO = PRN 1-5 and 59-63            as B1I, + index < ms - 30000 + 300 c   as B1I                BDS/B3I/include/NAVdecoding.m:69-164
  GLO_GL1/2  30-bit time mark x 10             (300)    |r| > 271, index + 300                 one 2000 later        GLO/GLO_GL1/include/NAVdecoding.m:66-105

`find_sync` returns the detected indices (the reference's `index`, 1-based), the candidates that satisfy the package's spacing
rule, and - where the reference verifies a candidate with arithmetic that belongs to this step (GPS: the parity of the TLM and
HOW words, navPartyChk.m; BDS: the BCH(15,11) check of the first word's second half) - the first verified start
(`subFrameStart`).  Galileo's and GLONASS' verification is the navigation decoder proper (Viterbi + CRC-24Q, Hamming string
check): out of this path, their `first` is None.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable

import numpy as np

PREAMBLE_BITS = np.array([1, -1, -1, -1, 1, -1, 1, 1], dtype=np.int8)           # GPS_L1CA NAVdecoding.m:69
NH20 = np.array([-1, -1, -1, -1, -1, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, 1, 1, 1, -1], dtype=np.int8)     # BDS/B1I NAVdecoding.m:72
BDS_PREAMBLE = np.array([1, 1, 1, -1, -1, -1, 1, -1, -1, 1, -1], dtype=np.int8)                               # :71
E1_SYNC = np.array([1, -1, 1, -1, -1, 1, 1, 1, 1, 1], dtype=np.int8)                                          # GAL_E1C NAVdecoding.m:80
E5A_CS20 = (1 - 2 * np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 1, 1, 0, 1, 0, 0, 1])).astype(np.int8)  # GAL_E5a NAVdecoding.m:69 ("842E9")
E5A_SYNC = np.array([-1, 1, -1, -1, 1, -1, -1, -1, 1, 1, 1, 1], dtype=np.int8)                                # :71
E5B_CS4 = np.array([-1, -1, -1, 1], dtype=np.int8)                                                            # GAL_E5b NAVdecoding.m:80 ("E")
E5B_PREAMBLE = np.array([1, -1, 1, -1, -1, 1, 1, 1, 1, 1], dtype=np.int8)                                     # :84
GLO_TIME_MARK = np.array([1, 1, 1, 1, 1, -1, -1, -1, 1, 1, -1, 1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, 1, -1, -1, 1, -1, 1, 1, -1],
                         dtype=np.int8)                                                                       # GLO_GL1 NAVdecoding.m:69-70

# IS-GPS-200 table 20-XIV: which of D29*, D30*, d1..d24 enter D25..D30 (1-based positions in [D29* D30* d1..d24])
_PARITY_TAPS = (
    (1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22, 25),
    (2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23, 26),
    (1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24),
    (2, 4, 6, 7, 8, 10, 11, 15, 16, 17, 18, 19, 22, 23, 25),
    (2, 3, 5, 7, 8, 9, 11, 12, 16, 17, 18, 19, 20, 23, 24, 26),
    (1, 5, 7, 8, 10, 11, 12, 13, 15, 17, 21, 24, 25, 26),
)


def navPartyChk(ndat) -> int:
    """Parity of one 30-bit word preceded by the last two bits of the previous word, all as +-1 (32 values).
    Returns -1 or +1 (word polarity) when the six parity bits check, 0 otherwise (Common/navPartyChk.m)."""
    b = np.asarray(ndat, dtype=np.int64).copy()
    if b[1] != 1:                       # D30* set: the data bits arrive inverted
        b[2:26] = -b[2:26]
    parity = [int(np.prod(b[[t - 1 for t in taps]])) for taps in _PARITY_TAPS]
    return int(-b[1]) if parity == [int(v) for v in b[26:32]] else 0


def bch_15_11_ok(bits) -> bool:
    """`[~, cnumerr] = bchdec(gf(bits, 1), 15, 11); cnumerr == 0` (BDS/B1I NAVdecoding.m:151-158): the 15 bits, first = highest
    power, are a code word of the BCH(15,11) code with generator x^4 + x + 1 iff the polynomial division leaves no remainder."""
    r = 0
    for b in bits:
        r = (r << 1) | int(b)
        if r & 0x10:
            r ^= 0x13
    return r == 0


@dataclass(frozen=True)
class SyncSpec:
    pattern: Callable[[int], np.ndarray]     # PRN -> int8 pattern at the stream's rate
    threshold: float
    strict: bool                             # True: |r| > threshold; False: >= threshold
    rounded: bool = False                    # round(|r|) before the comparison
    zero_is_plus: bool = False               # Galileo E1: bits = (I_P < 0), i.e. I_P == 0 counts as +1
    search_start_offset: int = 0
    index_shift: int = 0                     # GLONASS: + 300, the index points behind the time mark
    per_bit: Callable[[int], int] | None = None   # BDS: codrPerD of the PRN


def _bds_geo_b1i(prn):
    return prn <= 5                                                          # BDS/B1I NAVdecoding.m:88


def _bds_geo_b3i(prn):
    return 1 <= prn <= 5 or 59 <= prn <= 63                                   # BDS/B3I NAVdecoding.m:88


def _bds_pattern(geo):
    def f(prn):
        if geo(prn):
            return np.kron(BDS_PREAMBLE, np.ones(2, dtype=np.int8)).astype(np.int8)          # preamble_D2, :77
        return np.kron(BDS_PREAMBLE, -NH20).astype(np.int8)                                    # preamble_D1, :76
    return f


SYNC = {
    "GPS_L1CA": SyncSpec(lambda prn: np.kron(PREAMBLE_BITS, np.ones(20, dtype=np.int8)).astype(np.int8), 153.0, True),
    "GAL_E1C": SyncSpec(lambda prn: E1_SYNC.copy(), 9.99, False, rounded=True, zero_is_plus=True),
    "GAL_E5a": SyncSpec(lambda prn: np.kron(E5A_SYNC, E5A_CS20).astype(np.int8), 239.99, False, rounded=True),
    "GAL_E5b": SyncSpec(lambda prn: np.kron(E5B_PREAMBLE, E5B_CS4).astype(np.int8), 39.99, True),
    "BDS_B1I": SyncSpec(_bds_pattern(_bds_geo_b1i), 10.0, False, search_start_offset=1000, per_bit=lambda prn: 2 if _bds_geo_b1i(prn) else 20),
    "BDS_B3I": SyncSpec(_bds_pattern(_bds_geo_b3i), 10.0, False, search_start_offset=1000, per_bit=lambda prn: 2 if _bds_geo_b3i(prn) else 20),
    "GLO_GL1": SyncSpec(lambda prn: np.kron(GLO_TIME_MARK, np.ones(10, dtype=np.int8)).astype(np.int8), 271.0, True, index_shift=300),
}
SYNC["GLO_GL2"] = SYNC["GLO_GL1"]            # GLO_GL2/include/NAVdecoding.m is the same file


def _matlab_round(x):
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def find_sync(engine, package: str, i_p, ms_to_process: int | None = None, prn: int = 0, search_start_offset: int | None = None):
    """The bit-synchronisation block of `package`'s NAVdecoding.m on the prompt stream `i_p` (one value per code period; Galileo
    E1: per 4-ms symbol).  `ms_to_process` = settings.msToProcess (default: len(i_p) - E1: 4 len(i_p)); `search_start_offset`
    replaces the file's own searchStartOffset (0; BDS: 1000).
    Returns a namespace: xcorr (float32, lags 0 .. n - 1 of the searched part), index (1-based, as the reference's `index`
    when its loop starts), candidates (those that pass the package's spacing rule), first (the verified start or None)."""
    spec = SYNC[package]
    x = np.asarray(i_p, dtype=np.float64).reshape(-1)
    off = spec.search_start_offset if search_start_offset is None else int(search_start_offset)
    pat = spec.pattern(prn)
    if x.shape[0] <= off:
        raise ValueError("find_sync: the stream ends before searchStartOffset")
    corr = engine.sync_xcorr(x[off:], pat, zero_is_plus=spec.zero_is_plus)
    a = np.abs(corr.astype(np.float64))
    if spec.rounded:
        a = _matlab_round(a)
    c = spec.per_bit(prn) if spec.per_bit else 1
    thr = spec.threshold * c
    index = np.flatnonzero(a > thr if spec.strict else a >= thr) + 1 + off + spec.index_shift
    n = x.shape[0]
    ms = (4 * n if package == "GAL_E1C" else n) if ms_to_process is None else int(ms_to_process)
    first = None
    if package == "GPS_L1CA":
        index = index[(index > 40) & (index < ms - (20 * 60 - 1))]                                   # :100
        cand = np.array([i for i in index if np.any(index - i == 6000)], dtype=np.int64)             # :111-113
        for i in cand:                                                                              # :125-141
            bits = x[i - 40 - 1:i + 20 * 60 - 1].reshape(-1, 20).sum(axis=1)
            bits = np.where(bits > 0, 1, -1)
            if navPartyChk(bits[0:32]) != 0 and navPartyChk(bits[30:62]) != 0:
                first = int(i)
                break
    elif package == "GAL_E1C":
        cand = np.array([i for i in index if np.any(index - i == 250) and np.any(index - i == 500) and i < ms / 4 - 7500], dtype=np.int64)   # :108-110
    elif package == "GAL_E5a":
        cand = np.array([i for i in index if np.any(np.abs(index - i) == 10000)], dtype=np.int64)    # :102-108: index = newIndex
        index = cand
    elif package == "GAL_E5b":
        cand = np.array([i for i in index if np.any(index - i == 1000) and (n - off) - (i - off) + 1 > 30000], dtype=np.int64)   # :118-119
    elif package in ("BDS_B1I", "BDS_B3I"):
        if package == "BDS_B3I":
            index = index[index < ms - 1500 * 20 + 300 * c]                                          # B3I :113
        cand = np.array([i for i in index if np.any(index - i == 300 * c)], dtype=np.int64)          # B1I :131
        for i in cand:                                                                              # :143-166
            if i + 30 * c - 1 > n:
                break                       # the reference would stop with an index error here
            bits = x[i - 1:i + 30 * c - 1].reshape(-1, c).sum(axis=1)
            if bch_15_11_ok((bits > 0).astype(np.int64)[15:30]):
                first = int(i)
                break
    else:                                                                                            # GLONASS :105
        cand = np.array([i for i in index if np.any(index - i == 2000)], dtype=np.int64)
    return SimpleNamespace(xcorr=corr, index=index.astype(np.int64), candidates=cand, first=first, pattern=pat)


def find_subframe_start(engine, i_p, ms_to_process: int, search_start_offset: int = 0):
    """GPS L1 C/A: GPS/GPS_L1CA/include/NAVdecoding.m:66-145.  Returns the 1-based index of the first sub-frame start, or None."""
    return find_sync(engine, "GPS_L1CA", i_p, ms_to_process, search_start_offset=search_start_offset).first
