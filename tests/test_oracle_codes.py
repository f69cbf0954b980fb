"""Known-answer tests that pin the code generators: the reference ships no
vectors, so the pins are the public ICD values."""
import numpy as np
import pytest

from cu_sdr_collection_amd import codes as C
from oracle import c_oracle as CO
from oracle import gnss_oracle as O

# IS-GPS-200 Table 3-Ia, "first 10 chips, octal" for PRN 1..32
ICD_FIRST10_OCTAL = [0o1440, 0o1620, 0o1710, 0o1744, 0o1133, 0o1455, 0o1131, 0o1454, 0o1626, 0o1504,
                     0o1642, 0o1750, 0o1764, 0o1772, 0o1775, 0o1776, 0o1156, 0o1467, 0o1633, 0o1715,
                     0o1746, 0o1763, 0o1063, 0o1706, 0o1743, 0o1761, 0o1770, 0o1774, 0o1127, 0o1453,
                     0o1625, 0o1712]


def _first10(code):
    return int("".join("1" if v > 0 else "0" for v in code[:10]), 2)


def test_ca_code_first_ten_chips_match_icd():
    for prn, want in enumerate(ICD_FIRST10_OCTAL, start=1):
        assert _first10(O.generate_ca_code(prn)) == want, prn  # logic 1 -> +1 (generateCAcode.m:90)


def test_ca_code_three_implementations_agree():
    import cu_sdr_collection_amd as P
    for prn in range(1, 52):  # incl. the SBAS shifts of generateCAcode.m:47-50
        a = O.generate_ca_code(prn)
        assert np.array_equal(a, CO.generate_ca(prn))
        assert np.array_equal(a, P.codes.generateCAcode(prn).astype(np.float64))


def test_ca_code_gold_properties():
    for prn in (1, 7, 19, 32):
        c = O.generate_ca_code(prn)
        assert set(np.unique(c)) == {-1.0, 1.0} and c.shape == (1023,)
        assert c.sum() == 1.0  # balance: 512 ones, 511 zeros with logic 1 -> +1
        ac = np.array([np.dot(c, np.roll(c, k)) for k in range(1, 1023)])
        assert set(np.unique(ac)) <= {-1.0, 63.0, -65.0}
    x = np.dot(O.generate_ca_code(3), np.roll(O.generate_ca_code(11), 77))
    assert x in (-1.0, 63.0, -65.0)


def test_sampled_table_matches_product_side():
    import cu_sdr_collection_amd as P
    S = P.initSettings()
    for prn in (1, 17, 32):
        t = O.make_ca_table(prn, S)
        assert t.shape == (18000,)
        assert np.array_equal(t, P.codes.makeCaTable(prn, S).astype(np.float64))
        assert t[-1] == O.generate_ca_code(prn)[1022]  # makeCaTable.m:62 forces the last index
    assert np.array_equal(O.pad_code(O.generate_ca_code(5)),
                          P.codes.padded_table(P.codes.generateCAcode(5)).astype(np.float64))


def test_galileo_e1_memory_codes_match_icd_and_product():
    """Galileo OS SIS ICD Annex C: E1-B PRN 1 starts F5D710130573541B, E1-C PRN 1 B39340CA1C817D81
    (hex of the logic-level chips); the oracle and product read separately packed copies."""
    import os
    import cu_sdr_collection_amd as P
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = np.load(os.path.join(root, "tests", "golden", "gal_e1_memory_codes.npz"))
    b = np.load(os.path.join(root, "cu-sdr-collection_amd", "data", "gal_e1_memory_codes.npz"))
    assert bytes(a["E1b"][0, :8]).hex().upper() == "F5D710130573541B"
    assert bytes(a["E1c"][0, :8]).hex().upper() == "B39340CA1C817D81"
    assert np.array_equal(a["E1b"], b["E1b"]) and np.array_equal(a["E1c"], b["E1c"])
    for prn in (1, 11, 50):
        for comp, fn in (("B", P.codes.generateE1Bcode), ("C", P.codes.generateE1Ccode)):
            o = O.generate_e1_code(prn, comp)
            assert o.shape == (8184,) and np.array_equal(o, fn(prn).astype(np.float64))
            assert np.array_equal(o[0::2], -o[1::2])  # BOC(1,1): every chip is [+c, -c]
    first = O.generate_e1_code(1, "B")[0:16:2]  # logic 1 -> -1: 0xF5 = 11110101
    assert list(first) == [-1, -1, -1, -1, 1, -1, 1, -1]


def test_gps_l5_codes_two_implementations_agree():
    import cu_sdr_collection_amd as P
    for prn in (1, 2, 19, 37):
        i5, q5 = O.generate_l5_code(prn, "I"), O.generate_l5_code(prn, "Q")
        assert np.array_equal(i5, P.codes.generateL5Icode(prn).astype(np.float64))
        assert np.array_equal(q5, P.codes.generateL5Qcode(prn).astype(np.float64))
        assert i5.shape == (10230,) and abs(i5.sum()) < 200 and abs(np.dot(i5, q5)) < 400
    # XA is short-cycled to 8190 chips: chips 8190.. repeat chips 0.. of XA; I5 of different PRNs share XA,
    # so their product is a pure XB x XB sequence with the 8191-chip m-sequence period structure
    a, b = O.generate_l5_code(1, "I"), O.generate_l5_code(2, "I")
    assert not np.array_equal(a, b)


def test_glonass_and_b1i_codes_two_implementations_agree():
    import cu_sdr_collection_amd as P
    g = O.generate_glo_code()
    assert g.shape == (511,) and np.array_equal(g, P.codes.generateGLOcode().astype(np.float64))
    ac = np.array([np.dot(g, np.roll(g, k)) for k in range(1, 511)])
    assert np.all(ac == -1.0)  # maximal-length sequence
    for prn in (1, 8, 9, 22, 37):
        b = O.generate_b1i_code(prn)
        assert b.shape == (2046,) and np.array_equal(b, P.codes.generateCAcode53(prn).astype(np.float64))
        assert abs(b.sum()) <= 2


# ---- 10.23-Mcps family, L2C, B1C: product generators against the per-chip oracle restatement and ICD KATs -------
def _first24(code, base):
    v = int("".join(str(int(b)) for b in ((1 - np.asarray(code[:24])) // 2).astype(int)), 2)
    return ("%06X" if base == 16 else "%08o") % v


def test_e5_codes_match_icd_initial_sequences_and_oracle():
    """Galileo OS SIS ICD tables 15/17 list the first 24 chips of every primary code in hex (logic 1 <-> chip -1):
    E5a-I code 1 = 3CEA9D, E5a-Q code 1 = 515537."""
    assert _first24(C.generateE5aIcode(1), 16) == "3CEA9D"
    assert _first24(C.generateE5aQcode(1), 16) == "515537"
    for sig, fn in (("e5ai", C.generateE5aIcode), ("e5aq", C.generateE5aQcode), ("e5bi", C.generateE5bIcode), ("e5bq", C.generateE5bQcode)):
        for prn in (1, 17, 50):
            assert np.array_equal(fn(prn), O.generate_e5_primary(sig, prn)), (sig, prn)
        assert abs(int(fn(3).sum())) < 400
    assert np.array_equal(C.generateE5aIcode(2, 2), O.generate_e5_code("e5ai", 2, 2))
    assert C.generateE5aIcode(2, 2).shape == (204600,)
    assert np.array_equal(C.generateE5bIcode(5, 2), O.generate_e5_code("e5bi", 5, 2))
    assert np.array_equal(C.generateE5aQ_secondary(7), O.generate_e5_secondary100("e5aq", 7))
    assert np.array_equal(C.generateE5bQcode(9, 2)[:30690], O.generate_e5_code("e5bq", 9, 2)[:30690])


def test_b2a_b3i_codes():
    """BDS-SIS-ICD-B2a table 5-2: data code of PRN 1 starts 26771056 (octal, first 24 chips)."""
    assert _first24(C.generateB2aDataCode(1), 8) == "26771056"
    for prn in (1, 30, 63):
        assert np.array_equal(C.generateB2aDataCode(prn), O.generate_b2a_code(prn, "data"))
        assert np.array_equal(C.generateB2aPilotCode(prn), O.generate_b2a_code(prn, "pilot"))
        assert np.array_equal(C.generateB3Icode(prn), O.generate_b3i_code(prn))
    # register 1 has period 8190 in both B2a components and in B3I's G1: the code is NOT 8190-periodic, but the
    # product of two PRNs' codes is (register 1 cancels, register 2 runs free with period 8191) -- structure check
    a, b = C.generateB2aDataCode(1).astype(int), C.generateB2aDataCode(2).astype(int)
    assert not np.array_equal(a[:2040], a[8190:])
    assert abs(int((a * b).sum())) < 600


def test_l2c_codes():
    cm = C.generateCMcode(1)
    assert cm.shape == (20460,) and not cm[1::2].any() and set(np.unique(cm[0::2])) == {-1, 1}
    assert np.array_equal(cm, O.generate_l2c_code(1, "CM", 10230))
    assert np.array_equal(C.generateCMcode(63), O.generate_l2c_code(63, "CM", 10230))
    assert np.array_equal(C.generateCMcode(159), O.generate_l2c_code(159, "CM", 10230))
    cl = C.generateCLcode(5, 4000)
    assert not cl[0::2].any() and np.array_equal(cl, O.generate_l2c_code(5, "CL", 4000))
    with pytest.raises(ValueError):
        C.generateCMcode(100)


def test_b1c_weil_codes():
    assert O.jacobi_symbol(2, 10243) == (1 if pow(2, 5121, 10243) == 1 else -1)
    for prn in (1, 40, 63):
        for which, fn in (("data", C.generateDataBOC11), ("pilot11", C.generatePilotBOC11)):
            assert np.array_equal(fn(prn), O.generate_b1c_code(prn, which)), (prn, which)
    assert np.array_equal(C.generatePilotBOC61(2), O.generate_b1c_code(2, "pilot61"))
    assert C.generatePilotBOC61(2).shape == (122760,)
    assert np.array_equal(C.generatePilot2ndCodes(3), O.generate_b1c_code(3, "secondary"))
    d = C.generateDataBOC11(1).astype(int)
    assert np.array_equal(d[0::2], -d[1::2])              # BOC(1,1): chip x [-1, +1]
    assert abs(int(d[1::2].sum())) < 300                  # Weil codes are balanced
