"""Development-time extraction of ICD constant tables (register initial states, advances, secondary codes,
Weil-code parameters) that the reference embeds as literals in its code generators.  They are interface-document
DATA, not algorithm: BDS-SIS-ICD-B2a-1.0 tables 5-2/5-3 (register-2 initial states), BDS-SIS-ICD-B3I-1.0 table 4-1
(phase advances), Galileo OS SIS ICD tables 15/17 (E5 base-register-2 start values, octal) and 19 (E5a-Q/E5b-Q
secondary codes CS100, hex), IS-GPS-200 table 3-IIa (L2 CM / CL initial shift-register states, octal),
BDS-SIS-ICD-B1C-1.0 tables 5-2/5-3 (Weil-code phase difference w and truncation point p).

Run in the build container only (needs /root/reference):   python tests/golden/make_icd_tables.py
Output: cu-sdr-collection_amd/data/icd_tables.npz (committed; the GPU box and users never need the reference)."""
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(__file__), "..", "..", "cu-sdr-collection_amd", "data", "icd_tables.npz")


def text(path):
    lines = open(os.path.join(REF, path), encoding="latin-1").read().splitlines()
    return "\n".join(l for l in lines if not l.lstrip().startswith("%"))


def literal(path, name):
    """The bracketed literal assigned to `name`."""
    t = text(path)
    m = re.search(re.escape(name) + r"\s*=\s*\[(.*?)\];", t, re.S)
    assert m, (path, name)
    body = re.sub(r"\.\.\.[^\n]*", " ", m.group(1))   # continuation marks and their trailing comments
    return body


def bit_rows(path, name, width):
    nums = re.findall(r"\b[01]\b", literal(path, name))
    a = np.array(nums, dtype=np.uint8).reshape(-1, width)
    return a


def ints(path, name):
    return np.array([int(x) for x in re.findall(r"\d+", literal(path, name))], dtype=np.int64)


def quoted(path, name):
    return re.findall(r"'([0-9A-Fa-f]+)'", literal(path, name))


out = {}
out["b2a_data_g2"] = bit_rows("BDS/B2a/include/generateB2aDataCode.m", "B2aData_reg2_ini", 13)
out["b2a_pilot_g2"] = bit_rows("BDS/B2a/include/generateB2aPilotCode.m", "B2aData_reg2_ini", 13)
out["b3i_advance"] = ints("BDS/B3I/include/generateB3Icode.m", "B3I_init")
for sig, f, n in (("e5ai", "GAL/GAL_E5a/include/generateE5aIcode.m", "e5ai_init"), ("e5aq", "GAL/GAL_E5a/include/generateE5aQcode.m", "e5aq_init"),
                  ("e5bi", "GAL/GAL_E5b/include/generateE5bIcode.m", "e5bi_init"), ("e5bq", "GAL/GAL_E5b/include/generateE5bQcode.m", "e5bq_init")):
    t = text(f)
    n = re.search(r"(e5\w+_init\w*)\s*=\s*\[", t).group(1)   # the E5b files reuse other names
    out[sig + "_start_octal"] = np.array([int(s, 8) for s in quoted(f, n)], dtype=np.int64)
    out[sig + "_poly_octal"] = np.array([int(re.search(r"Feedback_Reg%d\s*=\s*'(\d+)'" % k, t).group(1), 8) for k in (1, 2)], dtype=np.int64)
for sig, f in (("e5aq", "GAL/GAL_E5a/include/generateE5aQ_secondary.m"), ("e5bq", "GAL/GAL_E5b/include/generateE5bQ_secondary.m")):
    out[sig + "_secondary_hex"] = np.array(quoted(f, "secondary_code"))
out["l2cm_init_octal"] = np.array([int(str(v), 8) for v in ints("GPS/GPS_L2C/include/generateCMcode.m", "l2cm_init")], dtype=np.int64)
out["l2cl_init_octal"] = np.array([int(str(v), 8) for v in ints("GPS/GPS_L2C/include/generateCLcode.m", "l2cl_init")], dtype=np.int64)
for sig, f, n in (("b1c_data", "BDS/B1C/include/generateDataBOC11.m", "wp_data"), ("b1c_pilot", "BDS/B1C/include/generatePilotBOC11.m", None)):
    t = text(f)
    if n is None:
        n = re.search(r"(wp_\w+)\s*=\s*\[", t).group(1)
    out[sig + "_wp"] = ints(f, n).reshape(-1, 2)
t = text("BDS/B1C/include/generate2ndCode.m")
n = re.search(r"(\w+)\s*=\s*\[", t).group(1)
out["b1c_secondary_wp"] = ints("BDS/B1C/include/generate2ndCode.m", n).reshape(-1, 2)
for k, v in out.items():
    print(k, v.shape, v.dtype, v.ravel()[:4])
os.makedirs(os.path.dirname(OUT), exist_ok=True)
np.savez_compressed(OUT, **out)
print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes")
