"""Bit synchronisation ahead of navigation decoding (GPS/GPS_L1CA/include/NAVdecoding.m:55-100): find the first
sub-frame start in a channel's prompt in-phase stream.  The 160-tap preamble cross-correlation over the whole stream
runs on the GPU (gc_preamble_xcorr); the candidate filtering and the two-word parity check are restated here."""
from __future__ import annotations

import numpy as np

PREAMBLE_BITS = np.array([1, -1, -1, -1, 1, -1, 1, 1], dtype=np.int8)           # NAVdecoding.m:58
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
    Returns -1 or +1 (word polarity) when the six parity bits check, 0 otherwise (navPartyChk.m)."""
    b = np.asarray(ndat, dtype=np.int64).copy()
    if b[1] != 1:                       # D30* set: the data bits arrive inverted
        b[2:26] = -b[2:26]
    parity = [int(np.prod(b[[t - 1 for t in taps]])) for taps in _PARITY_TAPS]
    return int(-b[1]) if parity == [int(v) for v in b[26:32]] else 0


def find_subframe_start(engine, i_p, ms_to_process: int, search_start_offset: int = 0):
    """NAVdecoding.m:55-100.  Returns the 1-based index of the first sub-frame start, or None."""
    i_p = np.asarray(i_p, dtype=np.float64)
    pattern = np.kron(PREAMBLE_BITS, np.ones(20, dtype=np.int8))                # :60
    corr = engine.preamble_xcorr(i_p[search_start_offset:], pattern)            # :62-68, lags 0..
    index = np.flatnonzero(np.abs(corr) > 153) + 1 + search_start_offset        # :76-78 (1-based)
    index = index[(index > 40) & (index < ms_to_process - (20 * 60 - 1))]       # :81
    for i in index:                                                            # :84-98
        if np.any(index - i == 6000):
            bits = i_p[i - 40 - 1:i + 20 * 60 - 1].reshape(-1, 20).sum(axis=1)
            bits = np.where(bits > 0, 1, -1)
            if navPartyChk(bits[0:32]) != 0 and navPartyChk(bits[30:62]) != 0:
                return int(i)
    return None
