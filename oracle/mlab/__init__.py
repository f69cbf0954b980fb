"""oracle/mlab — a minimal interpreter for the MATLAB subset the reference's hot-path files are written in.

WHY.  The reference (gnsscusdr/CU-SDR-Collection) is 100 % MATLAB, ships no golden vectors, and neither MATLAB nor
Octave exists in the build container or on the GPU box.  Round 1's oracle was therefore a hand
restatement that nothing in the reference pinned ("parity unpinned").  This package closes that gap as far as
the container allows: it EXECUTES THE REFERENCE'S OWN SOURCE TEXT - tracking.m, acquisition.m, preRun.m, initSettings.m,
the code generators and their helpers, read in place from /root/reference at generation time - statement by statement, and
`tests/golden/make_ref_vectors.py` stores the results as fixtures (`tests/golden/ref_*.npz`).  The oracle and the HIP path are
then both checked against what the reference's statements computed, not against one reading of them.

WHAT IT IS NOT.  It is not MATLAB: the language core (parser, value semantics, indexing, struct arrays) and ~200 built-ins
(colon, ceil, rem, sum, fft, var, max, sort, fread ...) are restated here from MATLAB's documented behaviour, in NumPy
float64.  A misreading of a BUILT-IN would still be common to fixtures and oracle; a misreading of the REFERENCE (loop order,
index offsets, seek arithmetic, which variable is recorded when) no longer can be.  DESIGN.md §5 says the same.

TEST INFRASTRUCTURE ONLY.  Nothing under cu-sdr-collection_amd/ imports this package; it needs /root/reference and so runs
in the build container only (the fixtures travel, the reference does not).  No reference source is copied: files are
opened where they lie, parsed, executed, and only numbers are kept.

Scope of the language subset: functions / scripts / local functions, struct arrays with nested fields and auto-growth,
cell arrays, char rows, logical / linear / 2-D indexing with `end`, colon ranges, matrix literals with MATLAB's whitespace
rules, if / for / while / switch / try, anonymous functions, multiple return values, value semantics (copies wherever
MATLAB's copy-on-write would be observable).  N-D arrays, classes, integer storage classes and graphics are out (graphics
calls are no-ops).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from . import builtins as _B
from .interp import Interpreter  # noqa: F401
from .values import M, MCell, MError, MFunc, MStr, MStruct  # noqa: F401

register_file = _B.register_file


def to_matlab(x):
    """Python -> interpreter value: SimpleNamespace / dict -> struct, list of those -> struct array, str -> char,
    numbers / numpy arrays -> double arrays."""
    if isinstance(x, SimpleNamespace):
        x = vars(x)
    if isinstance(x, dict):
        return MStruct([{k: to_matlab(v) for k, v in x.items()}], list(x.keys()))
    if isinstance(x, (list, tuple)) and x and isinstance(x[0], (SimpleNamespace, dict)):
        elems = [to_matlab(e).elems[0] for e in x]
        return MStruct(elems, list(elems[0].keys()))
    if isinstance(x, str):
        return MStr(x)
    if isinstance(x, range):
        x = list(x)
    if isinstance(x, (list, tuple)):
        return np.array(x, dtype=np.float64).reshape(1, -1)
    return M(x)


def from_matlab(v):
    """Interpreter value -> Python: struct -> SimpleNamespace (struct array -> list), char -> str, 1x1 -> float,
    vectors -> 1-D numpy arrays, matrices -> 2-D."""
    if isinstance(v, MStr):
        return v.s
    if isinstance(v, MStruct):
        out = [SimpleNamespace(**{k: from_matlab(e[k]) for k in v.fields}) for e in v.elems]
        return out[0] if len(out) == 1 else out
    if isinstance(v, MCell):
        return [from_matlab(x) for x in v.a.reshape(-1, order="F")]
    if isinstance(v, MFunc):
        return v
    a = np.asarray(v)
    if a.size == 1:
        s = a.flat[0]
        if np.iscomplexobj(a):
            return complex(s)
        return bool(s) if a.dtype == np.bool_ else float(s)
    if 1 in a.shape:
        return a.reshape(-1).copy()
    return a.copy()
