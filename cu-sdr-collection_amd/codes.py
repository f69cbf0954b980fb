"""Host-side spreading-code generators and replica tables that feed the device.

Independent bit-level implementations (integer LFSRs) of what the reference computes with
+-1 product registers; outputs are int8 in {-1, 0, +1} ready for gc_set_code / gc_acquire_coarse.
"""
from __future__ import annotations

import math

import numpy as np

# G2 code delays in chips for PRN 1..32 (IS-GPS-200 Table 3-Ia) followed by the SBAS delays the
# reference also carries (generateCAcode.m:42-50).
_CA_G2_DELAY = (5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470,
                471, 472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862,
                145, 175, 52, 21, 237, 235, 886, 657, 634, 762, 355, 1012, 176, 603, 130, 359,
                595, 68, 386)


def _lfsr10(taps) -> np.ndarray:
    """1023 output bits of a 10-stage Fibonacci LFSR, all-ones start, output = stage 10."""
    reg = [1] * 10  # reg[0] = stage 1
    out = np.empty(1023, dtype=np.uint8)
    for i in range(1023):
        out[i] = reg[9]
        fb = 0
        for t in taps:
            fb ^= reg[t - 1]
        reg = [fb] + reg[:9]
    return out


_G1 = _lfsr10((3, 10))
_G2 = _lfsr10((2, 3, 6, 8, 9, 10))


def generateCAcode(PRN: int) -> np.ndarray:
    """GPS C/A code of `PRN` as int8 +-1, logic 1 -> +1 (same convention as
    GPS/GPS_L1CA/include/generateCAcode.m:90)."""
    if not 1 <= PRN <= len(_CA_G2_DELAY):
        raise ValueError(f"PRN {PRN} out of range")
    g2d = np.roll(_G2, _CA_G2_DELAY[PRN - 1])
    logic = _G1 ^ g2d
    return (2 * logic.astype(np.int8) - 1).astype(np.int8)


def padded_table(code: np.ndarray) -> np.ndarray:
    """[c(end) c c(1)] — the replica table of tracking.m:158."""
    code = np.asarray(code, dtype=np.int8)
    return np.concatenate([code[-1:], code, code[:1]])


def samplesPerCode(settings) -> int:
    x = settings.samplingFreq / (settings.codeFreqBasis / settings.codeLength)
    return int(math.floor(x + 0.5))  # MATLAB round() for positive x


def makeCaTable(PRN: int, settings) -> np.ndarray:
    """Sampled C/A code for acquisition (makeCaTable.m:43-67): index ceil(ts*(1:spc)/tc), last = 1023."""
    spc = samplesPerCode(settings)
    ts = 1.0 / settings.samplingFreq
    tc = 1.0 / settings.codeFreqBasis
    idx = np.ceil((ts * np.arange(1, spc + 1, dtype=np.float64)) / tc).astype(np.int64)
    idx[-1] = 1023
    return generateCAcode(PRN)[idx - 1]


# ---------------------------------------------------------------------------------------------
# Galileo E1-B / E1-C (GAL/GAL_E1C/include/generateE1Bcode.m, generateE1Ccode.m)
# ---------------------------------------------------------------------------------------------
_E1 = None


def _e1_primary(which: str, PRN: int) -> np.ndarray:
    """4092-chip memory code of Galileo OS SIS ICD Annex C as 0/1 bits (data/gal_e1_memory_codes.npz:
    the packed form of the tables the reference reads from E1b.dat / E1c.dat)."""
    global _E1
    if not 1 <= PRN <= 50:
        raise ValueError(f"Galileo PRN {PRN} out of range")
    if _E1 is None:
        import os
        _E1 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "gal_e1_memory_codes.npz"))
    return np.unpackbits(_E1[which][PRN - 1])[:4092]


def _boc11(bits: np.ndarray) -> np.ndarray:
    chips = (1 - 2 * bits.astype(np.int8)).astype(np.int8)      # logic 1 -> -1 (generateE1Bcode.m:56)
    return np.stack([chips, -chips], axis=1).reshape(-1)         # sub-carrier [+1, -1] (:59-65)


def generateE1Bcode(PRN: int) -> np.ndarray:
    """E1-B data code with BOC(1,1): 8184 half-chips, int8 +-1."""
    return _boc11(_e1_primary("E1b", PRN))


def generateE1Ccode(PRN: int) -> np.ndarray:
    """E1-C pilot primary code with BOC(1,1): 8184 half-chips, int8 +-1."""
    return _boc11(_e1_primary("E1c", PRN))


# ---------------------------------------------------------------------------------------------
# GPS L5 I5 / Q5 (GPS/GPS_L5C/include/generateL5Icode.m, generateL5Qcode.m)
# ---------------------------------------------------------------------------------------------
# XB code advance in chips for PRN 1..37, IS-GPS-705 Table 3-I
_L5I_ADVANCE = (266, 365, 804, 1138, 1509, 1559, 1756, 2084, 2170, 2303, 2527, 2687, 2930, 3471, 3940, 4132, 4332,
                4924, 5343, 5443, 5641, 5816, 5898, 5918, 5955, 6243, 6345, 6477, 6518, 6875, 7168, 7187, 7329, 7577,
                7720, 7777, 8057)
_L5Q_ADVANCE = (1701, 323, 5292, 2020, 5429, 7136, 1041, 5947, 4315, 148, 535, 1939, 5206, 5910, 3595, 5135, 6082,
                6990, 3546, 1523, 4548, 4484, 1893, 3961, 7106, 5299, 4660, 276, 4389, 3783, 1591, 1601, 749, 1387,
                1661, 3210, 708)


def _lfsr13(taps, n_out: int, skip: int = 0, short_cycle_at: int | None = None) -> np.ndarray:
    """13-stage Fibonacci LFSR in integer form (bit i = stage i+1), all-ones start, output = stage 13.
    `short_cycle_at`: reload all ones after emitting from this state (XA: 0b1111111111101 -> 8190 period)."""
    mask = sum(1 << (t - 1) for t in taps)
    reg = 0x1FFF
    for _ in range(skip):
        fb = bin(reg & mask).count("1") & 1
        reg = ((reg << 1) & 0x1FFF) | fb
    out = np.empty(n_out, dtype=np.uint8)
    for i in range(n_out):
        out[i] = (reg >> 12) & 1
        if short_cycle_at is not None and reg == short_cycle_at:
            reg = 0x1FFF
        else:
            fb = bin(reg & mask).count("1") & 1
            reg = ((reg << 1) & 0x1FFF) | fb
    return out


_XA = None


def _l5(PRN: int, adv) -> np.ndarray:
    global _XA
    if not 1 <= PRN <= len(adv):
        raise ValueError(f"GPS L5 PRN {PRN} out of range")
    if _XA is None:
        # stage 12 = logic 0, all others 1  <->  the reference's reset_state [-1 x11, +1, -1]
        _XA = _lfsr13((9, 10, 12, 13), 10230, short_cycle_at=0x1FFF & ~(1 << 11))
    xb = _lfsr13((1, 3, 4, 6, 7, 8, 12, 13), 10230, skip=adv[PRN - 1])
    # product of +-1 registers with logic 1 -> -1:  XOR = 1 -> -1  (generateL5Icode.m:123)
    return (1 - 2 * (_XA ^ xb).astype(np.int8)).astype(np.int8)


def generateL5Icode(PRN: int) -> np.ndarray:
    """GPS L5 I5 (data) code, 10230 chips int8 +-1."""
    return _l5(PRN, _L5I_ADVANCE)


def generateL5Qcode(PRN: int) -> np.ndarray:
    """GPS L5 Q5 (pilot) code, 10230 chips int8 +-1."""
    return _l5(PRN, _L5Q_ADVANCE)


# ---------------------------------------------------------------------------------------------
# GLONASS L1OF / L2OF ranging code (GLO/GLO_GL1/include/generateCAcode.m:93-104)
# ---------------------------------------------------------------------------------------------
def generateGLOcode() -> np.ndarray:
    """511-chip m-sequence x^9 + x^5 + 1, output of stage 7, all-ones start; int8 +-1 with logic 1 -> -1
    (the reference keeps the raw register value, no final negation)."""
    reg = [1] * 9
    out = np.empty(511, dtype=np.int8)
    for i in range(511):
        out[i] = 1 - 2 * reg[6]
        fb = reg[4] ^ reg[8]
        reg = [fb] + reg[:8]
    return out


# ---------------------------------------------------------------------------------------------
# BDS B1I (BDS/B1I/include/generateCAcode53.m), PRN 1..37
# ---------------------------------------------------------------------------------------------
_B1I_PHASE = ((1, 3), (1, 4), (1, 5), (1, 6), (1, 8), (1, 9), (1, 10), (1, 11), (2, 7), (3, 4), (3, 5), (3, 6), (3, 8),
              (3, 9), (3, 10), (3, 11), (4, 5), (4, 6), (4, 8), (4, 9), (4, 10), (4, 11), (5, 6), (5, 8), (5, 9), (5, 10),
              (5, 11), (6, 8), (6, 9), (6, 10), (6, 11), (8, 9), (8, 10), (8, 11), (9, 10), (9, 11), (10, 11))


def generateCAcode53(PRN: int) -> np.ndarray:
    """BDS B1I ranging code, 2046 chips int8 +-1 (logic 1 -> +1 after the reference's final negation).
    11-stage G1 (taps 1,7,8,9,10,11) and G2 (taps 1,2,3,4,5,8,9,11), initial state 01010101010,
    G2 tapped at the PRN's two phase-selector stages (BDS-SIS-ICD-B1I Table 4-1)."""
    if not 1 <= PRN <= len(_B1I_PHASE):
        raise ValueError(f"BDS B1I PRN {PRN} out of range")
    init = [0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0]
    s1, s2 = _B1I_PHASE[PRN - 1]
    r1, r2 = list(init), list(init)
    out = np.empty(2046, dtype=np.int8)
    for i in range(2046):
        g1 = r1[10]
        g2 = r2[s1 - 1] ^ r2[s2 - 1]
        out[i] = 2 * (g1 ^ g2) - 1
        f1 = r1[0] ^ r1[6] ^ r1[7] ^ r1[8] ^ r1[9] ^ r1[10]
        f2 = r2[0] ^ r2[1] ^ r2[2] ^ r2[3] ^ r2[4] ^ r2[7] ^ r2[8] ^ r2[10]
        r1 = [f1] + r1[:10]
        r2 = [f2] + r2[:10]
    return out
