"""Host-side spreading-code generators and replica tables that feed the device.

Independent bit-level implementations (integer LFSRs) of what the reference computes with
+-1 product registers; outputs are int8 in {-1, 0, +1} ready for gc_set_code / gc_acquire_coarse.
"""
from __future__ import annotations

import functools
import math

import numpy as np


def _kept(fn):
    """The codes are constants of the signal: each (generator, arguments) is computed once per process and handed out as a fresh
    copy (GPS L5 / Galileo E5 / BDS B2a / B3I registers cost 9-18 ms per PRN in Python, the 767 250-chip L2 CL code 285 ms - the
    default searches of those packages spent 0.2-0.5 s generating codes around ~10 ms of GPU work)."""
    store = {}

    @functools.wraps(fn)
    def wrapper(*args):
        key = tuple(int(a) if isinstance(a, (int, np.integer, float)) and float(a) == int(a) else a for a in args)
        code = store.get(key)
        if code is None:
            code = store[key] = fn(*args)
            code.setflags(write=False)
        return code.copy()
    wrapper.__wrapped__ = fn
    return wrapper

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


_CA_CODE = {}


def generateCAcode(PRN: int) -> np.ndarray:
    """GPS C/A code of `PRN` as int8 +-1, logic 1 -> +1 (same convention as
    GPS/GPS_L1CA/include/generateCAcode.m:90)."""
    if not 1 <= PRN <= len(_CA_G2_DELAY):
        raise ValueError(f"PRN {PRN} out of range")
    code = _CA_CODE.get(PRN)
    if code is None:
        g2d = np.roll(_G2, _CA_G2_DELAY[PRN - 1])
        logic = _G1 ^ g2d
        code = _CA_CODE[PRN] = (2 * logic.astype(np.int8) - 1).astype(np.int8)
    return code.copy()


def padded_table(code: np.ndarray) -> np.ndarray:
    """[c(end) c c(1)] — the replica table of tracking.m:158."""
    code = np.asarray(code, dtype=np.int8)
    return np.concatenate([code[-1:], code, code[:1]])


def samplesPerCode(settings) -> int:
    x = settings.samplingFreq / (settings.codeFreqBasis / settings.codeLength)
    return int(math.floor(x + 0.5))  # MATLAB round() for positive x


_CA_INDEX = {}
_CA_TABLE = {}


def _ca_table_index(settings) -> np.ndarray:
    """codeValueIndex of makeCaTable.m:52-58, the same for every PRN of a front end (kept per (fs, chip rate, spc))."""
    spc = samplesPerCode(settings)
    key = (float(settings.samplingFreq), float(settings.codeFreqBasis), spc)
    idx = _CA_INDEX.get(key)
    if idx is None:
        ts = 1.0 / settings.samplingFreq
        tc = 1.0 / settings.codeFreqBasis
        idx = np.ceil((ts * np.arange(1, spc + 1, dtype=np.float64)) / tc).astype(np.int64)
        idx[-1] = 1023
        idx -= 1
        if len(_CA_INDEX) > 16:
            _CA_INDEX.clear()
        _CA_INDEX[key] = idx
    return idx


def makeCaTable(PRN: int, settings) -> np.ndarray:
    """Sampled C/A code for acquisition (makeCaTable.m:43-67): index ceil(ts*(1:spc)/tc), last = 1023."""
    idx = _ca_table_index(settings)
    key = (int(PRN), float(settings.samplingFreq), float(settings.codeFreqBasis), idx.size)
    tab = _CA_TABLE.get(key)
    if tab is None:                      # 32 PRNs x 65 us per acquisition call otherwise: kept like the index (read-only)
        tab = generateCAcode(PRN)[idx]
        tab.setflags(write=False)
        if len(_CA_TABLE) > 256:
            _CA_TABLE.clear()
        _CA_TABLE[key] = tab
    return tab


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


@_kept
def generateE1Bcode(PRN: int) -> np.ndarray:
    """E1-B data code with BOC(1,1): 8184 half-chips, int8 +-1."""
    return _boc11(_e1_primary("E1b", PRN))


@_kept
def generateE1Ccode(PRN: int) -> np.ndarray:
    """E1-C pilot primary code with BOC(1,1): 8184 half-chips, int8 +-1."""
    return _boc11(_e1_primary("E1c", PRN))


@_kept
def generateE1C_BOC61(PRN: int) -> np.ndarray:
    """E1-C pilot primary code with the BOC(6,1) subcarrier of CBOC(6,1,1/11): 49104 entries (12 per chip), chip x
    [+1, -1] x 6 - same subcarrier phase convention as the BOC(1,1) table above (generateE1Bcode.m:59-65)."""
    c = (1 - 2 * _e1_primary("E1c", PRN).astype(np.int8)).astype(np.int8)   # logic 1 -> -1, as _boc11
    return (c[:, None] * np.tile(np.array([1, -1], dtype=np.int8), 6)[None, :]).reshape(-1)


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


@_kept
def generateL5Icode(PRN: int) -> np.ndarray:
    """GPS L5 I5 (data) code, 10230 chips int8 +-1."""
    return _l5(PRN, _L5I_ADVANCE)


@_kept
def generateL5Qcode(PRN: int) -> np.ndarray:
    """GPS L5 Q5 (pilot) code, 10230 chips int8 +-1."""
    return _l5(PRN, _L5Q_ADVANCE)


# ---------------------------------------------------------------------------------------------
# GLONASS L1OF / L2OF ranging code (GLO/GLO_GL1/include/generateCAcode.m:93-104)
# ---------------------------------------------------------------------------------------------
@_kept
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
# BDS B1I (BDS/B1I/include/generateCAcode53.m), PRN 1..58
# ---------------------------------------------------------------------------------------------
_B1I_PHASE = ((1, 3), (1, 4), (1, 5), (1, 6), (1, 8), (1, 9), (1, 10), (1, 11), (2, 7), (3, 4), (3, 5), (3, 6), (3, 8),
              (3, 9), (3, 10), (3, 11), (4, 5), (4, 6), (4, 8), (4, 9), (4, 10), (4, 11), (5, 6), (5, 8), (5, 9), (5, 10),
              (5, 11), (6, 8), (6, 9), (6, 10), (6, 11), (8, 9), (8, 10), (8, 11), (9, 10), (9, 11), (10, 11),
              # PRN 38..58 (BDS-3 satellites, BDS-SIS-ICD-B1I-3.0 Table 4-1): three phase-selector stages
              (1, 2, 7), (1, 3, 4), (1, 3, 6), (1, 3, 8), (1, 3, 10), (1, 3, 11), (1, 4, 5), (1, 4, 9), (1, 5, 6), (1, 5, 8),
              (1, 5, 10), (1, 5, 11), (1, 6, 9), (1, 8, 9), (1, 9, 10), (1, 9, 11), (2, 3, 7), (2, 5, 7), (2, 7, 9), (3, 4, 5),
              (3, 4, 9))


@_kept
def generateCAcode53(PRN: int) -> np.ndarray:
    """BDS B1I ranging code, 2046 chips int8 +-1 (logic 1 -> +1 after the reference's final negation).
    11-stage G1 (taps 1,7,8,9,10,11) and G2 (taps 1,2,3,4,5,8,9,11), initial state 01010101010,
    G2 tapped at the PRN's two (PRN <= 37) or three phase-selector stages (BDS-SIS-ICD-B1I Table 4-1)."""
    if not 1 <= PRN <= len(_B1I_PHASE):
        raise ValueError(f"BDS B1I PRN {PRN} out of range")
    init = [0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0]
    sel = _B1I_PHASE[PRN - 1]
    r1, r2 = list(init), list(init)
    out = np.empty(2046, dtype=np.int8)
    for i in range(2046):
        g1 = r1[10]
        g2 = 0
        for st in sel:
            g2 ^= r2[st - 1]
        out[i] = 2 * (g1 ^ g2) - 1
        f1 = r1[0] ^ r1[6] ^ r1[7] ^ r1[8] ^ r1[9] ^ r1[10]
        f2 = r2[0] ^ r2[1] ^ r2[2] ^ r2[3] ^ r2[4] ^ r2[7] ^ r2[8] ^ r2[10]
        r1 = [f1] + r1[:10]
        r2 = [f2] + r2[:10]
    return out


# ---------------------------------------------------------------------------------------------
# ICD constant tables (register initial states, advances, Weil parameters): data/icd_tables.npz,
# extracted once by tests/golden/make_icd_tables.py
# ---------------------------------------------------------------------------------------------
_ICD = None


def _icd(name: str) -> np.ndarray:
    global _ICD
    if _ICD is None:
        import os
        _ICD = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "icd_tables.npz")))
    return _ICD[name]


def _prn_row(table: np.ndarray, PRN: int, what: str):
    if not 1 <= PRN <= table.shape[0]:
        raise ValueError(f"{what} PRN {PRN} out of range 1..{table.shape[0]}")
    return table[PRN - 1]


def _fib_lfsr(nbits: int, taps, state: int, n_out: int, reset_after: int | None = None, reset_state: int | None = None,
              skip: int = 0) -> np.ndarray:
    """Fibonacci LFSR in integer form, bit i = stage i+1, feedback = XOR of `taps` (1-based stages) shifted into
    stage 1, output = stage nbits.  reset_after: reload all ones after that many output chips (BDS B2a register 1);
    reset_state: reload all ones instead of shifting when the register equals it (BDS B3I G1)."""
    full = (1 << nbits) - 1
    mask = sum(1 << (t - 1) for t in taps)
    reg = state
    for _ in range(skip):
        reg = ((reg << 1) & full) | (bin(reg & mask).count("1") & 1)
    out = np.empty(n_out, dtype=np.uint8)
    for i in range(n_out):
        out[i] = reg >> (nbits - 1)
        if reset_state is not None and reg == reset_state:
            reg = full
        else:
            reg = ((reg << 1) & full) | (bin(reg & mask).count("1") & 1)
        if reset_after is not None and i + 1 == reset_after:
            reg = full
    return out


def _stages_to_int(bits) -> int:
    """bits[j] = logic value of stage j+1  ->  integer with bit j = stage j+1."""
    return int(sum(int(b) << j for j, b in enumerate(bits)))


# ---------------------------------------------------------------------------------------------
# BDS B2a data / pilot (BDS/B2a/include/generateB2aDataCode.m:109-138, generateB2aPilotCode.m:104-138)
# ---------------------------------------------------------------------------------------------
def _b2a(PRN: int, taps1, taps2, g2_table: str) -> np.ndarray:
    g2 = _stages_to_int(_prn_row(_icd(g2_table), PRN, "BDS B2a"))
    r1 = _fib_lfsr(13, taps1, 0x1FFF, 10230, reset_after=8190)   # register 1 restarts from all ones after chip 8190
    r2 = _fib_lfsr(13, taps2, g2, 10230)
    return (1 - 2 * (r1 ^ r2).astype(np.int8)).astype(np.int8)


@_kept
def generateB2aDataCode(PRN: int) -> np.ndarray:
    """BDS B2a data-channel code, 10230 chips int8 +-1 (logic 1 -> -1)."""
    return _b2a(PRN, (1, 5, 11, 13), (3, 5, 9, 11, 12, 13), "b2a_data_g2")


@_kept
def generateB2aPilotCode(PRN: int) -> np.ndarray:
    """BDS B2a pilot-channel code, 10230 chips int8 +-1."""
    return _b2a(PRN, (3, 6, 7, 13), (1, 5, 7, 8, 12, 13), "b2a_pilot_g2")


# ---------------------------------------------------------------------------------------------
# BDS B3I (BDS/B3I/include/generateB3Icode.m:39-110)
# ---------------------------------------------------------------------------------------------
_B3I_G1 = None


@_kept
def generateB3Icode(PRN: int) -> np.ndarray:
    """BDS B3I ranging code, 10230 chips int8 +-1: G1 (taps 1,3,4,13; restarts when it reaches 1111111111100,
    period 8190) times G2 (taps 1,5,6,7,9,10,12,13) pre-advanced by the PRN's table entry."""
    global _B3I_G1
    adv = int(_prn_row(_icd("b3i_advance"), PRN, "BDS B3I"))
    if _B3I_G1 is None:
        # reset_state [-1 x11, +1, +1] in the reference's +-1 form = stages 1..11 logic 1, stages 12, 13 logic 0
        _B3I_G1 = _fib_lfsr(13, (1, 3, 4, 13), 0x1FFF, 10230, reset_state=0x07FF)
    g2 = _fib_lfsr(13, (1, 5, 6, 7, 9, 10, 12, 13), 0x1FFF, 10230, skip=adv)
    return (1 - 2 * (_B3I_G1 ^ g2).astype(np.int8)).astype(np.int8)


# ---------------------------------------------------------------------------------------------
# Galileo E5a-I/Q, E5b-I/Q primary codes and the tiered E5-I codes
# (GAL/GAL_E5a/include/generateE5aIcode.m:36-124, generateE5aQcode.m, generateE5aQ_secondary.m; GAL_E5b twins)
# ---------------------------------------------------------------------------------------------
def _e5_primary(sig: str, PRN: int) -> np.ndarray:
    """Two 14-stage registers written MSB-first: output = first element, feedback = XOR of the elements selected
    by the first 14 bits of the octal polynomial, shifted in at the last element.  Register 1 starts at all ones,
    register 2 at the PRN's start value (Galileo OS SIS ICD tables 15 / 17)."""
    start = int(_prn_row(_icd(sig + "_start_octal"), PRN, "Galileo " + sig))
    polys = _icd(sig + "_poly_octal")
    out = []
    for poly, reg in ((int(polys[0]), 0x3FFF), (int(polys[1]), start)):
        nb = poly.bit_length()                       # dec2bin drops leading zeros
        sel = (poly >> (nb - 14)) & 0x3FFF           # first 14 binary digits, digit 1 = element 1 = bit 13 here
        bits = np.empty(10230, dtype=np.uint8)
        for i in range(10230):
            bits[i] = (reg >> 13) & (sel >> 13)      # RegOut(1) = Register(1) * taps(1)
            fb = bin(reg & sel).count("1") & 1
            reg = ((reg << 1) & 0x3FFF) | fb
        out.append(bits)
    return (1 - 2 * (out[0] ^ out[1]).astype(np.int8)).astype(np.int8)


_E5I_SECONDARY = {"e5ai": (20, 0x842E9), "e5bi": (4, 0xE)}   # CS20_1 / CS4_1, Galileo OS SIS ICD table 18


def _e5_i(sig: str, PRN: int, flag: int) -> np.ndarray:
    prim = _e5_primary(sig, PRN)
    if flag == 1:
        return prim
    n, word = _E5I_SECONDARY[sig]
    sec = np.array([1 - 2 * ((word >> (n - 1 - k)) & 1) for k in range(n)], dtype=np.int8)
    return (sec[:, None] * prim[None, :]).reshape(-1)


@_kept
def generateE5aIcode(PRN: int, flag: int = 1) -> np.ndarray:
    """flag 1: 10230-chip primary code; flag 2: the 20-ms tiered code (primary x CS20_1 = 842E9)."""
    return _e5_i("e5ai", PRN, flag)


@_kept
def generateE5bIcode(PRN: int, flag: int = 1) -> np.ndarray:
    """flag 1: primary; flag 2: the 4-ms tiered code (primary x CS4_1 = E)."""
    return _e5_i("e5bi", PRN, flag)


def _e5_secondary100(sig: str, PRN: int) -> np.ndarray:
    hexstr = str(_prn_row(_icd(sig + "_secondary_hex"), PRN, "Galileo " + sig))
    bits = [(int(hexstr[:13], 16) >> (51 - k)) & 1 for k in range(52)] + [(int(hexstr[13:], 16) >> (47 - k)) & 1 for k in range(48)]
    return (1 - 2 * np.array(bits, dtype=np.int8)).astype(np.int8)


def _e5_q(sig: str, PRN: int, flag: int) -> np.ndarray:
    prim = _e5_primary(sig, PRN)
    if flag == 1:
        return prim
    return (_e5_secondary100(sig, PRN)[:, None] * prim[None, :]).reshape(-1)


@_kept
def generateE5aQcode(PRN: int, flag: int = 1) -> np.ndarray:
    return _e5_q("e5aq", PRN, flag)


@_kept
def generateE5bQcode(PRN: int, flag: int = 1) -> np.ndarray:
    return _e5_q("e5bq", PRN, flag)


@_kept
def generateE5aQ_secondary(PRN: int) -> np.ndarray:
    """100-chip secondary code CS100 of E5a-Q as +-1."""
    return _e5_secondary100("e5aq", PRN)


@_kept
def generateE5bQ_secondary(PRN: int) -> np.ndarray:
    return _e5_secondary100("e5bq", PRN)


# ---------------------------------------------------------------------------------------------
# GPS L2C CM / CL (GPS/GPS_L2C/include/generateCMcode.m:39-111, generateCLcode.m): 27-stage modular register,
# returned RZ-interleaved with zeros (CM in the even, CL in the odd positions of the doubled-rate code)
# ---------------------------------------------------------------------------------------------
_L2C_XOR = (4, 7, 9, 12, 15, 17, 19, 22, 23, 24, 25)


def _l2c_index(PRN: int) -> int:
    if 1 <= PRN <= 63:
        return PRN - 1
    if 159 <= PRN <= 210:
        return PRN - 96
    raise ValueError(f"GPS L2C PRN {PRN} does not exist")


def _l2c_chips(state: int, n: int) -> np.ndarray:
    """state: the octal initial state read as a 27-digit binary number, digit 1 = stage 1 ... digit 27 = output."""
    xor_mask = sum(1 << (27 - p) for p in _L2C_XOR)   # digit p <-> bit 27 - p
    out = np.empty(n, dtype=np.uint8)
    reg = state
    for i in range(n):
        o = reg & 1                                   # digit 27
        out[i] = o
        reg = (reg >> 1) | (o << 26)                  # rotate: the output re-enters at digit 1
        if o:
            reg ^= xor_mask
    return out


@_kept
def generateCMcode(PRN: int, codeLength: int = 10230) -> np.ndarray:
    """L2 CM code as the reference returns it: 2*codeLength entries [chip, 0, chip, 0, ...], chips +-1."""
    chips = 1 - 2 * _l2c_chips(int(_icd("l2cm_init_octal")[_l2c_index(PRN)]), codeLength).astype(np.int8)
    out = np.zeros(2 * codeLength, dtype=np.int8)
    out[0::2] = chips
    return out


@_kept
def generateCLcode(PRN: int, CLCodeLength: int = 767250) -> np.ndarray:
    """L2 CL code: 2*CLCodeLength entries [0, chip, 0, chip, ...]."""
    chips = 1 - 2 * _l2c_chips(int(_icd("l2cl_init_octal")[_l2c_index(PRN)]), CLCodeLength).astype(np.int8)
    out = np.zeros(2 * CLCodeLength, dtype=np.int8)
    out[1::2] = chips
    return out


# ---------------------------------------------------------------------------------------------
# BDS B1C Weil codes (BDS/B1C/include/generateDataBOC11.m:43-91, generatePilotBOC11.m, generatePilotBOC61.m:44-96,
# generate2ndCode.m, JacobiSymbol.m)
# ---------------------------------------------------------------------------------------------
_LEGENDRE = {}


def _legendre_bits(N: int) -> np.ndarray:
    """L[0] = 0, L[i] = 1 if i is a quadratic residue mod the prime N else 0 (Jacobi symbol -1 -> 0)."""
    if N not in _LEGENDRE:
        L = np.zeros(N, dtype=np.uint8)
        L[(np.arange(1, N, dtype=np.int64) ** 2) % N] = 1
        _LEGENDRE[N] = L
    return _LEGENDRE[N]


def _weil(N: int, w: int, p: int, n: int) -> np.ndarray:
    L = _legendre_bits(N)
    k = (np.arange(n, dtype=np.int64) + p - 1) % N
    return (1 - 2 * (L[k] ^ L[(k + w) % N]).astype(np.int8)).astype(np.int8)


@_kept
def generateB1Cprimary(PRN: int, component: str) -> np.ndarray:
    w, p = _prn_row(_icd("b1c_data_wp" if component == "data" else "b1c_pilot_wp"), PRN, "BDS B1C")
    return _weil(10243, int(w), int(p), 10230)


@_kept
def generateDataBOC11(PRN: int) -> np.ndarray:
    """B1C data component with the BOC(1,1) sub-carrier baked in: 20460 half-chips, chip x [-1, +1]."""
    c = generateB1Cprimary(PRN, "data")
    return (c[:, None] * np.array([-1, 1], dtype=np.int8)[None, :]).reshape(-1)


@_kept
def generatePilotBOC11(PRN: int) -> np.ndarray:
    c = generateB1Cprimary(PRN, "pilot")
    return (c[:, None] * np.array([-1, 1], dtype=np.int8)[None, :]).reshape(-1)


@_kept
def generatePilotBOC61(PRN: int) -> np.ndarray:
    """B1C pilot BOC(6,1) component: 122760 entries, chip x (-1)^ii, ii = 1..12."""
    c = generateB1Cprimary(PRN, "pilot")
    return (c[:, None] * np.tile(np.array([-1, 1], dtype=np.int8), 6)[None, :]).reshape(-1)


@_kept
def generatePilot2ndCodes(PRN: int) -> np.ndarray:
    """1800-chip B1C pilot secondary (overlay) code: Weil code of length 3607."""
    w, p = _prn_row(_icd("b1c_secondary_wp"), PRN, "BDS B1C")
    return _weil(3607, int(w), int(p), 1800)
