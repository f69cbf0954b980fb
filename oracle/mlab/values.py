"""Value model of the mini-MATLAB interpreter: every numeric value is a 2-D numpy array (float64 / complex128 / bool),
char rows are MStr, struct arrays MStruct (1 x n list of ordered dicts), cell arrays MCell, function handles MFunc."""
from __future__ import annotations

import copy
import math

import numpy as np


class MError(Exception):
    """A MATLAB run-time error (error(), bad index, unknown function ...)."""


class MStr:
    __slots__ = ("s",)

    def __init__(self, s: str = ""):
        self.s = s

    def __repr__(self):
        return f"MStr({self.s!r})"

    def codes(self) -> np.ndarray:
        return np.array([[float(ord(c)) for c in self.s]], dtype=np.float64).reshape(1, len(self.s))


class MStruct:
    """1 x n struct array; elems[k] maps field name -> value.  All elements share the field order of `fields`."""
    __slots__ = ("elems", "fields")

    def __init__(self, elems=None, fields=None):
        self.elems = [dict()] if elems is None else elems
        self.fields = list(fields) if fields is not None else (list(self.elems[0].keys()) if self.elems else [])

    def add_field(self, name):
        if name not in self.fields:
            self.fields.append(name)
            for e in self.elems:
                e.setdefault(name, empty())

    def __repr__(self):
        return f"MStruct(1x{len(self.elems)}, fields={self.fields})"


class MCell:
    __slots__ = ("a",)

    def __init__(self, a=None):
        self.a = np.empty((0, 0), dtype=object) if a is None else a


class MFunc:
    __slots__ = ("fn", "name")

    def __init__(self, fn, name="@"):
        self.fn = fn
        self.name = name


def empty():
    return np.zeros((0, 0))


def M(x):
    """Anything -> interpreter value."""
    if isinstance(x, np.ndarray):
        if x.ndim == 2:
            return x
        if x.ndim == 0:
            return x.reshape(1, 1)
        if x.ndim == 1:
            return x.reshape(1, -1)
        raise MError("arrays of more than two dimensions are not supported")
    if isinstance(x, (MStr, MStruct, MCell, MFunc)):
        return x
    if isinstance(x, str):
        return MStr(x)
    if isinstance(x, (bool, np.bool_)):
        return np.array([[bool(x)]])
    if isinstance(x, (int, float, np.integer, np.floating)):
        return np.array([[float(x)]])
    if isinstance(x, (complex, np.complexfloating)):
        return np.array([[complex(x)]])
    if isinstance(x, (list, tuple)):
        return np.array(x, dtype=np.float64).reshape(1, -1)
    raise MError(f"cannot convert {type(x).__name__} to a MATLAB value")


def num(x) -> np.ndarray:
    """Numeric view of a value (char -> codes, logical stays bool)."""
    if isinstance(x, np.ndarray):
        return x
    if isinstance(x, MStr):
        return x.codes()
    if isinstance(x, (int, float, complex, bool, np.number, np.bool_)):
        return M(x)
    raise MError(f"numeric value expected, got {type(x).__name__}")


def fnum(x) -> np.ndarray:
    """Numeric view, logical promoted to double (arithmetic operands)."""
    a = num(x)
    return a.astype(np.float64) if a.dtype == np.bool_ else a


def scalar(x) -> float:
    a = num(x)
    if a.size != 1:
        raise MError(f"scalar expected, got {a.shape[0]}x{a.shape[1]}")
    v = a.flat[0]
    if isinstance(v, (complex, np.complexfloating)):
        if v.imag == 0:
            return float(v.real)
        return complex(v)
    return float(v)


def iscalar(x) -> int:
    v = scalar(x)
    if isinstance(v, complex) or v != math.floor(v):
        raise MError(f"integer expected, got {v}")
    return int(v)


def is_true(x) -> bool:
    """MATLAB's `if` rule: non-empty and all elements non-zero."""
    if isinstance(x, MStr):
        return len(x.s) > 0 and all(ord(c) != 0 for c in x.s)
    a = num(x)
    if a.size == 0:
        return False
    return bool(np.all(a != 0))


def mcopy(x):
    """Value-semantics copy (MATLAB is copy-on-write; the interpreter copies wherever an alias would be created)."""
    if isinstance(x, np.ndarray):
        return x.copy() if x.dtype != object else copy.deepcopy(x)
    if isinstance(x, MStruct):
        return MStruct([{k: mcopy(v) for k, v in e.items()} for e in x.elems], x.fields)
    if isinstance(x, MCell):
        c = np.empty(x.a.shape, dtype=object)
        for idx, v in np.ndenumerate(x.a):
            c[idx] = mcopy(v)
        return MCell(c)
    return x


def simplify_complex(a: np.ndarray) -> np.ndarray:
    """MATLAB drops an all-zero imaginary part only in specific functions; arithmetic keeps complex storage.  We keep
    complex results complex, except for exact-zero imaginary parts produced by real-valued functions (real, abs ...)."""
    return a


def shape_str(x) -> str:
    if isinstance(x, np.ndarray):
        return f"{x.shape[0]}x{x.shape[1]} {x.dtype}"
    return type(x).__name__
