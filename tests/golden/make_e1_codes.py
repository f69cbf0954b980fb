"""Packs the Galileo E1-B / E1-C primary (memory) codes — public Galileo OS SIS ICD Annex C content
that the reference carries as ASCII tables GAL/GAL_E1C/include/E1b.dat / E1c.dat (one digit per line,
50 PRNs x 4092 chips) — into 51 kB of bits.  Memory codes have no generator, so the data itself has to
travel.  Run once in the build container:  python tests/golden/make_e1_codes.py"""
import os

import numpy as np

REF = "/root/reference/GAL/GAL_E1C/include"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
out = {}
for name in ("E1b", "E1c"):
    bits = np.loadtxt(os.path.join(REF, name + ".dat"), dtype=np.uint8)
    assert bits.shape == (50 * 4092,) and set(np.unique(bits)) == {0, 1}
    out[name] = np.packbits(bits.reshape(50, 4092), axis=1)
for dst in (os.path.join(HERE, "gal_e1_memory_codes.npz"),
            os.path.join(ROOT, "cu-sdr-collection_amd", "data", "gal_e1_memory_codes.npz")):
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst))
