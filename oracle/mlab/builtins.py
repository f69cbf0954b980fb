"""Operators, indexing and the built-in function table of the mini-MATLAB interpreter.

Every function here restates a MATLAB built-in from its documented behaviour (the ones with numerics that matter to the
reference's hot path: colon, ceil/round/rem, sum, fft/ifft, var, max/sort tie rules, fread); none of it is reference code."""
from __future__ import annotations

import math
import os
import time

import numpy as np

from .values import (M, MCell, MError, MFunc, MStr, MStruct, empty, fnum, is_true, iscalar, mcopy, num, scalar, shape_str)

COLON = object()   # the bare ':' index


# ======================================================================================================
# operators
# ======================================================================================================
def _arith_operands(a, b):
    return fnum(a), fnum(b)


def _bcast_check(x, y, op):
    if x.size == 1 or y.size == 1:
        return
    (r1, c1), (r2, c2) = x.shape, y.shape
    if (r1 == r2 or r1 == 1 or r2 == 1) and (c1 == c2 or c1 == 1 or c2 == 1):
        return
    raise MError(f"Matrix dimensions must agree ({r1}x{c1} {op} {r2}x{c2})")


def _power(x, y):
    with np.errstate(all="ignore"):
        if np.iscomplexobj(x) or np.iscomplexobj(y):
            return np.power(x.astype(np.complex128), y)
        r = np.power(x, y)
        bad = np.isnan(r) & ~np.isnan(x) & ~np.isnan(y)
        if np.any(bad):
            return np.power(x.astype(np.complex128), y)
        return r


def binop(op, a, b):
    if op in ("==", "~=", "<", "<=", ">", ">="):
        x, y = num(a), num(b)
        _bcast_check(x, y, op)
        if op == "==":
            return x == y
        if op == "~=":
            return x != y
        xr = x.real if np.iscomplexobj(x) else x
        yr = y.real if np.iscomplexobj(y) else y
        return {"<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal}[op](xr, yr)
    if op in ("&", "|"):
        x, y = num(a) != 0, num(b) != 0
        _bcast_check(x, y, op)
        return (x & y) if op == "&" else (x | y)
    x, y = _arith_operands(a, b)
    with np.errstate(all="ignore"):
        if op == "+":
            _bcast_check(x, y, op)
            return x + y
        if op == "-":
            _bcast_check(x, y, op)
            return x - y
        if op == ".*":
            _bcast_check(x, y, op)
            return x * y
        if op == "./":
            _bcast_check(x, y, op)
            return _divide(x, y)
        if op == ".\\":
            _bcast_check(x, y, op)
            return _divide(y, x)
        if op == ".^":
            _bcast_check(x, y, op)
            return _power(x, y)
        if op == "*":
            if x.size == 1 or y.size == 1:
                return x * y
            if x.shape[1] != y.shape[0]:
                raise MError(f"Inner matrix dimensions must agree ({x.shape[0]}x{x.shape[1]} * {y.shape[0]}x{y.shape[1]})")
            return x @ y
        if op == "/":
            if y.size == 1:
                return _divide(x, y)
            # x / y = x * inv(y): least squares on the transposed system
            return np.linalg.lstsq(y.T, x.T, rcond=None)[0].T
        if op == "\\":
            if x.size == 1:
                return _divide(y, x)
            if x.shape[0] == x.shape[1]:
                return np.linalg.solve(x, y)
            return np.linalg.lstsq(x, y, rcond=None)[0]
        if op == "^":
            if x.size == 1 and y.size == 1:
                return _power(x, y)
            if y.size == 1 and x.shape[0] == x.shape[1] and float(y.flat[0]).is_integer():
                return np.linalg.matrix_power(x, int(y.flat[0]))
            raise MError("matrix power of this form is not supported")
    raise MError(f"unknown operator {op}")


def _divide(x, y):
    with np.errstate(all="ignore"):
        if np.iscomplexobj(x) or np.iscomplexobj(y):
            return x / y
        return np.true_divide(x, y)


def unop(op, a):
    if op == "-":
        return -fnum(a)
    if op == "+":
        return fnum(a)
    if op == "~":
        return num(a) == 0
    raise MError(f"unknown unary operator {op}")


def transpose(a, conj):
    if isinstance(a, MStr):
        if len(a.s) <= 1:
            return a
        return a.codes().T     # a char column: kept as codes (the hot path never prints one)
    if isinstance(a, MStruct):
        return a
    if isinstance(a, MCell):
        return MCell(a.a.T.copy())
    x = num(a)
    if conj and np.iscomplexobj(x):
        return np.conj(x.T)
    return x.T.copy()


def colon(a, st, b):
    """a:b and a:st:b for doubles, MathWorks' documented algorithm: integer operands count exactly; otherwise the element count
    comes from a tolerance test, the first half of the vector is built up from a, the second half down from the (snapped) end
    point, and the middle element of an odd-length vector is their mean."""
    a = scalar(num(a).reshape(-1)[:1].reshape(1, 1)) if num(a).size else None
    b = scalar(num(b).reshape(-1)[:1].reshape(1, 1)) if num(b).size else None
    d = 1.0 if st is None else (scalar(num(st).reshape(-1)[:1].reshape(1, 1)) if num(st).size else None)
    if a is None or b is None or d is None:
        return np.zeros((1, 0))
    a, b, d = (float(v.real) if isinstance(v, complex) else float(v) for v in (a, b, d))
    if d == 0 or (d > 0 and a > b) or (d < 0 and a < b) or math.isnan(a) or math.isnan(b) or math.isnan(d):
        return np.zeros((1, 0))
    if a == math.floor(a) and d == math.floor(d) and abs(a) < 2 ** 53 and abs(b) < 2 ** 53:
        n = int(math.floor((b - a) / d))
        return (a + d * np.arange(0, n + 1, dtype=np.float64)).reshape(1, -1)
    tol = 2.0 * np.finfo(np.float64).eps * max(abs(a), abs(b))
    sig = 1.0 if d > 0 else -1.0
    q = (b - a) / d
    n = int(math.floor(q + 0.5))
    if sig * (a + n * d - b) > tol:
        n -= 1
    c = a + n * d
    if sig * (c - b) > -tol:
        c = b
    out = np.empty(n + 1, dtype=np.float64)
    h = n // 2
    k = np.arange(0, h + 1, dtype=np.float64)
    ki = np.arange(0, h + 1)
    out[ki] = a + k * d
    out[n - ki] = c - k * d
    if n % 2 == 0:
        out[h] = (a + c) / 2.0
    return out.reshape(1, -1)


# ======================================================================================================
# indexing
# ======================================================================================================
def _to_positions(ix, n, what="index"):
    """One subscript -> (0-based int positions, shape of the subscript or None for ':' / masks)."""
    if ix is COLON or (isinstance(ix, MStr) and ix.s == ":"):
        return np.arange(n), None
    a = num(ix)
    if a.dtype == np.bool_:
        pos = np.flatnonzero(a.reshape(-1, order="F"))
        if pos.size and pos[-1] >= n and n >= 0:
            pass
        return pos, ("mask", a.shape)
    if np.iscomplexobj(a):
        raise MError("Subscript indices must be real positive integers")
    f = a.reshape(-1, order="F")
    p = np.rint(f).astype(np.int64)
    if f.size and (np.any(p != f) or np.any(p < 1)):
        raise MError(f"Subscript indices must either be real positive integers or logicals ({what})")
    return p - 1, a.shape


def get_index(base, idx):
    if isinstance(base, MFunc):
        raise MError("function handle indexing is handled by the caller")
    if isinstance(base, MStruct):
        if len(idx) == 1:
            pos, _ = _to_positions(idx[0], len(base.elems))
        elif len(idx) == 2:
            r, _ = _to_positions(idx[0], 1)
            if r.size != 1 or r[0] != 0:
                raise MError("struct arrays are 1 x n")
            pos, _ = _to_positions(idx[1], len(base.elems))
        else:
            raise MError("too many subscripts for a struct array")
        if pos.size and pos.max() >= len(base.elems):
            raise MError("Index exceeds the number of array elements (struct array)")
        return MStruct([base.elems[k] for k in pos], base.fields)
    if isinstance(base, MCell):
        sub = get_index(base.a, idx)
        return MCell(sub)
    if isinstance(base, MStr):
        r = get_index(base.codes(), idx)
        if r.shape[0] <= 1:
            return MStr("".join(chr(int(c)) for c in r.reshape(-1)))
        return r
    a = base if (isinstance(base, np.ndarray) and base.dtype == object) else num(base)
    if len(idx) == 0:
        return a
    if len(idx) == 1:
        n = a.size
        pos, shp = _to_positions(idx[0], n)
        if pos.size and pos.max() >= n:
            raise MError(f"Index exceeds the number of array elements ({int(pos.max()) + 1} > {n})")
        flat = a.reshape(-1, order="F")[pos]
        if shp is None:                       # A(:)
            return flat.reshape(-1, 1)
        if isinstance(shp, tuple) and shp and shp[0] == "mask":
            mshape = shp[1]
            if a.shape[0] == 1 and (mshape[0] == 1 or a.ndim == 2):
                return flat.reshape(1, -1) if a.shape[0] == 1 else flat.reshape(-1, 1)
            return flat.reshape(-1, 1)
        if (a.shape[0] == 1 or a.shape[1] == 1) and (shp[0] == 1 or shp[1] == 1) and a.size != 1:
            return flat.reshape(1, -1) if a.shape[0] == 1 else flat.reshape(-1, 1)
        return flat.reshape(shp, order="F")
    if len(idx) == 2:
        r, _ = _to_positions(idx[0], a.shape[0], "row")
        c, _ = _to_positions(idx[1], a.shape[1], "column")
        if (r.size and r.max() >= a.shape[0]) or (c.size and c.max() >= a.shape[1]):
            raise MError(f"Index in position exceeds array bounds ({a.shape[0]}x{a.shape[1]})")
        return a[np.ix_(r, c)]
    # trailing singleton subscripts
    if all(scalar(x) == 1 for x in idx[2:]):
        return get_index(base, idx[:2])
    raise MError("arrays of more than two dimensions are not supported")


def struct_linear_index(idx, n):
    if len(idx) == 2:
        if iscalar(idx[0]) != 1:
            raise MError("struct arrays are 1 x n")
        idx = idx[1:]
    if len(idx) != 1:
        raise MError("bad struct array subscript")
    return iscalar(idx[0]) - 1


def _common_dtype(cur, val):
    if cur.dtype == object or val.dtype == object:
        return object
    if np.iscomplexobj(cur) or np.iscomplexobj(val):
        return np.complex128
    if cur.dtype == np.bool_ and val.dtype == np.bool_:
        return np.bool_
    return np.float64


def set_index(cur, idx, value):
    """cur(idx) = value; returns the (possibly re-allocated) container."""
    if isinstance(cur, MStruct) or isinstance(value, MStruct):
        if cur is None or (isinstance(cur, np.ndarray) and cur.size == 0):
            cur = MStruct([], list(value.fields))
        if not isinstance(cur, MStruct) or not isinstance(value, MStruct) or len(value.elems) != 1:
            raise MError("struct array element assignment needs a scalar struct on the right")
        k = struct_linear_index(idx, len(cur.elems))
        for f in value.fields:
            cur.add_field(f)
        while len(cur.elems) <= k:
            cur.elems.append({f: empty() for f in cur.fields})
        el = {f: empty() for f in cur.fields}
        el.update(mcopy(value).elems[0])
        cur.elems[k] = el
        return cur
    if isinstance(cur, MCell):
        v = value if isinstance(value, MCell) else None
        if v is None:
            raise MError("()-assignment into a cell needs a cell on the right")
        cur.a = set_index(cur.a, idx, v.a)
        return cur
    was_str = isinstance(cur, MStr) or (cur is None and isinstance(value, MStr)) or \
        (isinstance(cur, np.ndarray) and cur.size == 0 and isinstance(value, MStr))
    a = empty() if cur is None else (cur.codes() if isinstance(cur, MStr) else cur)
    if not isinstance(a, np.ndarray):
        raise MError(f"cannot index-assign into {shape_str(a)}")
    v = value.codes() if isinstance(value, MStr) else value
    if isinstance(v, (MCell, MFunc)):
        raise MError("cannot store a cell / function handle in a numeric array")
    v = num(v) if a.dtype != object else v
    dt = _common_dtype(a, v)
    if a.dtype != dt:
        a = a.astype(dt)
    if len(idx) == 1:
        n = a.size
        pos, shp = _to_positions(idx[0], n)
        if shp is None and v.size != 1 and v.size != n:
            raise MError("A(:) = B: the number of elements must match")
        top = int(pos.max()) + 1 if pos.size else 0
        if top > n:
            if a.size == 0:
                a = np.zeros((1, top), dtype=dt)
            elif a.shape[0] == 1:
                g = np.zeros((1, top), dtype=dt)
                g[:, :n] = a
                a = g
            elif a.shape[1] == 1:
                g = np.zeros((top, 1), dtype=dt)
                g[:n, :] = a
                a = g
            else:
                raise MError("linear index beyond the end of a matrix cannot grow it")
        if v.size != 1 and v.size != pos.size:
            raise MError(f"In an assignment A(I) = B, the number of elements in B and I must be the same ({v.size} vs {pos.size})")
        nr = max(a.shape[0], 1)
        if a.dtype == object:                  # cell storage: element-wise, the elements may be arrays themselves
            src = v.reshape(-1, order="F")
            for k, q in enumerate(pos):
                a[q % nr, q // nr] = src[k if src.size != 1 else 0]
        else:
            vals = v.reshape(-1, order="F") if v.size != 1 else v.flat[0]
            a[pos % nr, pos // nr] = vals      # column-major linear index -> (row, column)
    elif len(idx) == 2:
        r, _ = _to_positions(idx[0], a.shape[0])
        c, _ = _to_positions(idx[1], a.shape[1])
        if idx[0] is COLON and a.shape[0] == 0 and v.ndim == 2:
            r = np.arange(v.shape[0] if v.size != 1 else 1)
        if idx[1] is COLON and a.shape[1] == 0 and v.ndim == 2:
            c = np.arange(v.shape[1] if v.size != 1 else 1)
        nr = max(a.shape[0], int(r.max()) + 1 if r.size else 0)
        nc = max(a.shape[1], int(c.max()) + 1 if c.size else 0)
        if (nr, nc) != a.shape:
            g = np.zeros((nr, nc), dtype=dt)
            g[:a.shape[0], :a.shape[1]] = a
            a = g
        if a.dtype == object:
            src = v.reshape(-1, order="F")
            k = 0
            for cc in c:
                for rr in r:
                    a[rr, cc] = src[k if src.size != 1 else 0]
                    k += 1
        elif v.size == 1:
            a[np.ix_(r, c)] = v.flat[0]
        else:
            if v.shape != (r.size, c.size):
                if v.size == r.size * c.size and (1 in (r.size, c.size)):
                    v = v.reshape(r.size, c.size)
                else:
                    raise MError(f"Subscripted assignment dimension mismatch ({v.shape[0]}x{v.shape[1]} into {r.size}x{c.size})")
            a[np.ix_(r, c)] = v
    else:
        raise MError("arrays of more than two dimensions are not supported")
    if was_str and a.shape[0] <= 1:
        return MStr("".join(chr(int(x)) for x in a.reshape(-1)))
    return a


def delete_index(cur, idx):
    a = num(cur)
    if len(idx) == 1:
        pos, _ = _to_positions(idx[0], a.size)
        keep = np.ones(a.size, dtype=bool)
        keep[pos] = False
        flat = a.reshape(-1, order="F")[keep]
        return flat.reshape(-1, 1) if (a.shape[1] == 1 and a.shape[0] != 1) else flat.reshape(1, -1)
    r, c = idx
    if r is COLON:
        pos, _ = _to_positions(c, a.shape[1])
        return np.delete(a, pos, axis=1)
    if c is COLON:
        pos, _ = _to_positions(r, a.shape[0])
        return np.delete(a, pos, axis=0)
    raise MError("A(i,j) = [] needs ':' in one position")


def end_value(obj, pos, n):
    if isinstance(obj, MStruct):
        return len(obj.elems) if (n == 1 or pos == 1) else 1
    if isinstance(obj, MStr):
        return len(obj.s) if (n == 1 or pos == 1) else 1
    if isinstance(obj, MCell):
        a = obj.a
    elif obj is None:
        return 0
    else:
        a = num(obj)
    if n == 1:
        return a.size
    if pos == 0:
        return a.shape[0]
    if pos == 1:
        return a.shape[1]
    return 1


def get_field(base, name):
    if not isinstance(base, MStruct):
        raise MError(f"Dot indexing is not supported for values of this type ({shape_str(base)}, field '{name}')")
    out = []
    for e in base.elems:
        if name not in e:
            raise MError(f"Reference to non-existent field '{name}'")
        out.append(e[name])
    return out


def cell_get(c, idx):
    if not isinstance(c, MCell):
        raise MError("{}-indexing of a non-cell")
    r = get_index(c.a, idx)
    if r.size != 1:
        raise MError("{}-indexing must select one cell here")
    return r.flat[0]


def cell_set(c, idx, v):
    box = np.empty((1, 1), dtype=object)
    box[0, 0] = v
    a = c.a
    if a.size == 0:
        a = np.empty((0, 0), dtype=object)
    new = set_index(a, idx, box)
    for i, x in np.ndenumerate(new):
        if x is None or (isinstance(x, (int, float)) and x == 0):
            new[i] = empty()
    c.a = new
    return c


def make_cell(rows):
    if not rows:
        return MCell()
    nc = len(rows[0])
    a = np.empty((len(rows), nc), dtype=object)
    for i, row in enumerate(rows):
        if len(row) != nc:
            raise MError("cell rows of different lengths")
        for j, v in enumerate(row):
            a[i, j] = v
    return MCell(a)


def concat(rows):
    """[a b; c d]"""
    any_str = any(isinstance(v, MStr) for row in rows for v in row)
    any_struct = any(isinstance(v, MStruct) for row in rows for v in row)
    any_cell = any(isinstance(v, MCell) for row in rows for v in row)
    if any_struct:
        elems, fields = [], None
        for row in rows:
            for v in row:
                if isinstance(v, np.ndarray) and v.size == 0:
                    continue
                if not isinstance(v, MStruct):
                    raise MError("cannot concatenate structs with non-structs")
                fields = fields or v.fields
                elems.extend(mcopy(v).elems)
        return MStruct(elems, fields or [])
    if any_cell:
        hs = []
        for row in rows:
            parts = [v.a for v in row if isinstance(v, MCell) and v.a.size]
            if parts:
                hs.append(np.concatenate(parts, axis=1))
        return MCell(np.concatenate(hs, axis=0) if hs else None)
    hrows = []
    for row in rows:
        parts = []
        for v in row:
            a = v.codes() if isinstance(v, MStr) else num(v)
            if a.size == 0 and a.shape[0] in (0, 1) and a.shape != (1, 0):
                if a.shape == (0, 0):
                    continue
            if a.shape == (0, 0):
                continue
            parts.append(a)
        if not parts:
            continue
        h = parts[0].shape[0]
        parts = [p for p in parts if not (p.size == 0 and p.shape[0] != h)] or parts
        for p in parts:
            if p.shape[0] != h:
                raise MError(f"Dimensions of arrays being concatenated are not consistent (rows {h} vs {p.shape[0]})")
        if all(p.dtype == np.bool_ for p in parts):
            hrows.append(np.concatenate(parts, axis=1))
        else:
            dt = np.complex128 if any(np.iscomplexobj(p) for p in parts) else np.float64
            hrows.append(np.concatenate([p.astype(dt) for p in parts], axis=1))
    if not hrows:
        return MStr("") if any_str else empty()
    w = hrows[0].shape[1]
    hrows = [h for h in hrows if h.shape[1] == w or h.size]
    for h in hrows:
        if h.shape[1] != w:
            raise MError(f"Dimensions of arrays being concatenated are not consistent (columns {w} vs {h.shape[1]})")
    if all(h.dtype == np.bool_ for h in hrows):
        out = np.concatenate(hrows, axis=0)
    else:
        dt = np.complex128 if any(np.iscomplexobj(h) for h in hrows) else np.float64
        out = np.concatenate([h.astype(dt) for h in hrows], axis=0)
    if any_str and out.shape[0] == 1:
        return MStr("".join(chr(int(round(float(x.real)))) if x.real >= 0 else "?" for x in out.reshape(-1)))
    return out


# ======================================================================================================
# built-in functions:  fn(interp, args, nargout) -> value | tuple | None
# ======================================================================================================
TABLE = {}


def builtin(*names):
    def deco(f):
        for n in names:
            TABLE[n] = f
        return f
    return deco


def _dims(args):
    """zeros(n) / zeros(m, n) / zeros([m n]) / zeros(m, n, 'like' ...)"""
    args = [a for a in args if not isinstance(a, MStr)]
    if not args:
        return 1, 1
    if len(args) == 1:
        a = num(args[0]).reshape(-1)
        if a.size == 1:
            return _dim(a[0]), _dim(a[0])
        if a.size >= 2:
            if a.size > 2 and np.any(a[2:] != 1):
                raise MError("arrays of more than two dimensions are not supported")
            return _dim(a[0]), _dim(a[1])
    r, c = _dim(scalar(args[0])), _dim(scalar(args[1]))
    for x in args[2:]:
        if scalar(x) != 1:
            raise MError("arrays of more than two dimensions are not supported")
    return r, c


def _dim(v):
    v = float(v)
    if math.isnan(v):
        raise MError("NaN as a size")
    return max(0, int(v))


@builtin("zeros")
def _zeros(I, a, n):
    return np.zeros(_dims(a))


@builtin("ones")
def _ones(I, a, n):
    return np.ones(_dims(a))


@builtin("inf", "Inf")
def _inf(I, a, n):
    return np.full(_dims(a), np.inf)


@builtin("nan", "NaN")
def _nan(I, a, n):
    return np.full(_dims(a), np.nan)


@builtin("eye")
def _eye(I, a, n):
    r, c = _dims(a)
    return np.eye(r, c)


@builtin("rand")
def _rand(I, a, n):
    raise MError("rand is not available: the reference path must be deterministic")


@builtin("pi")
def _pi(I, a, n):
    return math.pi


@builtin("eps")
def _eps(I, a, n):
    if a:
        return np.spacing(np.abs(fnum(a[0])))
    return float(np.finfo(np.float64).eps)


@builtin("i", "j", "1i", "1j")
def _imag_unit(I, a, n):
    return 1j


@builtin("true")
def _true(I, a, n):
    return np.ones(_dims(a), dtype=bool)


@builtin("false")
def _false(I, a, n):
    return np.zeros(_dims(a), dtype=bool)


@builtin("newline")
def _newline(I, a, n):
    return MStr("\n")


def _elementwise(name, f, complex_ok=True):
    def fn(I, a, n, _f=f):
        with np.errstate(all="ignore"):
            return _f(fnum(a[0]))
    TABLE[name] = fn


def _round_half_away(x):
    if np.iscomplexobj(x):
        return _round_half_away(x.real) + 1j * _round_half_away(x.imag)
    return np.where(x >= 0, np.floor(x + 0.5), -np.floor(-x + 0.5))


def _cplx_or_real(freal, fcplx=None):
    def f(x):
        if np.iscomplexobj(x):
            return (fcplx or freal)(x)
        return freal(x)
    return f


def _sqrt(x):
    if np.iscomplexobj(x):
        return np.sqrt(x)
    if np.any(x < 0):
        return np.sqrt(x.astype(np.complex128))
    return np.sqrt(x)


def _log_like(f):
    def g(x):
        if not np.iscomplexobj(x) and np.any(x < 0):
            return f(x.astype(np.complex128))
        return f(x)
    return g


for _n, _f in (("ceil", np.ceil), ("floor", np.floor), ("fix", np.trunc), ("round", _round_half_away),
               ("exp", np.exp), ("sin", np.sin), ("cos", np.cos), ("tan", np.tan), ("atan", np.arctan), ("asin", np.arcsin),
               ("acos", np.arccos), ("sinh", np.sinh), ("cosh", np.cosh), ("tanh", np.tanh), ("sqrt", _sqrt),
               ("log", _log_like(np.log)), ("log2", _log_like(np.log2)), ("log10", _log_like(np.log10)),
               ("sign", np.sign), ("conj", np.conj), ("isnan", np.isnan), ("isinf", np.isinf),
               ("isfinite", np.isfinite)):
    if _n in ("ceil", "floor", "fix"):
        def _mk(f):
            def g(x):
                if np.iscomplexobj(x):
                    return f(x.real) + 1j * f(x.imag)
                return f(x)
            return g
        _elementwise(_n, _mk(_f))
    else:
        _elementwise(_n, _f)


@builtin("abs")
def _abs(I, a, n):
    return np.abs(fnum(a[0]))


@builtin("real")
def _real(I, a, n):
    x = fnum(a[0])
    return np.ascontiguousarray(x.real) if np.iscomplexobj(x) else x


@builtin("imag")
def _imag(I, a, n):
    x = fnum(a[0])
    return np.ascontiguousarray(x.imag) if np.iscomplexobj(x) else np.zeros(x.shape)


@builtin("angle")
def _angle(I, a, n):
    return np.angle(fnum(a[0]))


@builtin("atan2")
def _atan2(I, a, n):
    return np.arctan2(fnum(a[0]), fnum(a[1]))


@builtin("rem")
def _rem(I, a, n):
    x, y = fnum(a[0]), fnum(a[1])
    with np.errstate(all="ignore"):
        r = np.fmod(x, y)            # exact, sign of x: MATLAB's rem
        r = np.where(y == 0, np.nan, r)
    return r


@builtin("mod")
def _mod(I, a, n):
    x, y = fnum(a[0]), fnum(a[1])
    with np.errstate(all="ignore"):
        r = np.mod(x, y)             # sign of y: MATLAB's mod; mod(x, 0) = x
        r = np.where(y == 0, x, r)
    return r


@builtin("power")
def _powerf(I, a, n):
    return binop(".^", a[0], a[1])


@builtin("times")
def _timesf(I, a, n):
    return binop(".*", a[0], a[1])


@builtin("xor")
def _xor(I, a, n):
    return (num(a[0]) != 0) ^ (num(a[1]) != 0)


@builtin("not")
def _not(I, a, n):
    return num(a[0]) == 0


@builtin("and")
def _andf(I, a, n):
    return binop("&", a[0], a[1])


@builtin("or")
def _orf(I, a, n):
    return binop("|", a[0], a[1])


@builtin("bitxor")
def _bitxor(I, a, n):
    return np.bitwise_xor(fnum(a[0]).astype(np.int64), fnum(a[1]).astype(np.int64)).astype(np.float64)


@builtin("bitand")
def _bitand(I, a, n):
    return np.bitwise_and(fnum(a[0]).astype(np.int64), fnum(a[1]).astype(np.int64)).astype(np.float64)


@builtin("bitor")
def _bitor(I, a, n):
    return np.bitwise_or(fnum(a[0]).astype(np.int64), fnum(a[1]).astype(np.int64)).astype(np.float64)


@builtin("bitshift")
def _bitshift(I, a, n):
    x, k = fnum(a[0]).astype(np.int64), iscalar(a[1])
    return (np.left_shift(x, k) if k >= 0 else np.right_shift(x, -k)).astype(np.float64)


def _reduce_dim(x, args):
    """Dimension a reduction works along: given, else the first non-singleton one (1 for scalars / empties)."""
    for a in args:
        if not isinstance(a, MStr) and num(a).size == 1:
            return iscalar(a) - 1
    if x.shape[0] != 1:
        return 0
    return 1


@builtin("sum")
def _sum(I, a, n):
    x = fnum(a[0])
    d = _reduce_dim(x, a[1:])
    return np.sum(x, axis=d, keepdims=True)


@builtin("prod")
def _prod(I, a, n):
    x = fnum(a[0])
    return np.prod(x, axis=_reduce_dim(x, a[1:]), keepdims=True)


@builtin("cumsum")
def _cumsum(I, a, n):
    x = fnum(a[0])
    return np.cumsum(x, axis=_reduce_dim(x, a[1:]))


@builtin("cumprod")
def _cumprod(I, a, n):
    x = fnum(a[0])
    return np.cumprod(x, axis=_reduce_dim(x, a[1:]))


@builtin("mean")
def _mean(I, a, n):
    x = fnum(a[0])
    d = _reduce_dim(x, a[1:])
    return np.sum(x, axis=d, keepdims=True) / x.shape[d]


def _var(x, d, w=0):
    m = np.sum(x, axis=d, keepdims=True) / x.shape[d]
    dev = x - m
    s = np.sum((dev * np.conj(dev)).real if np.iscomplexobj(x) else dev * dev, axis=d, keepdims=True)
    nn = x.shape[d]
    den = nn if (w == 1 or nn == 1) else nn - 1
    return s / den


@builtin("var")
def _varf(I, a, n):
    x = fnum(a[0])
    w = 0
    if len(a) > 1 and num(a[1]).size == 1:
        w = int(scalar(a[1]))
    d = iscalar(a[2]) - 1 if len(a) > 2 else (0 if x.shape[0] != 1 else 1)
    return _var(x, d, w)


@builtin("std")
def _stdf(I, a, n):
    return np.sqrt(_varf(I, a, n))


def _minmax(I, a, nargout, is_max):
    x = fnum(a[0])
    if len(a) >= 2 and not (isinstance(a[1], np.ndarray) and a[1].size == 0):
        y = fnum(a[1])
        _bcast_check(x, y, "max")
        if np.iscomplexobj(x) or np.iscomplexobj(y):
            raise MError("max/min of two complex arrays is not supported")
        with np.errstate(all="ignore"):
            return np.fmax(x, y) if is_max else np.fmin(x, y)
    d = iscalar(a[2]) - 1 if len(a) >= 3 else (0 if x.shape[0] != 1 else 1)
    if x.size == 0:
        return (empty(), empty()) if nargout > 1 else empty()
    key = np.abs(x) if np.iscomplexobj(x) else x
    nanmask = np.isnan(key)
    if is_max:
        k2 = np.where(nanmask, -np.inf, key)
        pos = np.argmax(k2, axis=d)             # first occurrence, as MATLAB
    else:
        k2 = np.where(nanmask, np.inf, key)
        pos = np.argmin(k2, axis=d)
    vals = np.take_along_axis(x, np.expand_dims(pos, d), axis=d)
    allnan = np.all(nanmask, axis=d, keepdims=True)
    if np.any(allnan):
        vals = np.where(allnan, np.nan, vals)
    if nargout > 1:
        return vals, np.expand_dims(pos, d).astype(np.float64) + 1.0
    return vals


@builtin("max")
def _max(I, a, n):
    return _minmax(I, a, n, True)


@builtin("min")
def _min(I, a, n):
    return _minmax(I, a, n, False)


@builtin("sort")
def _sort(I, a, nargout):
    x = fnum(a[0])
    desc = any(isinstance(v, MStr) and v.s.lower() == "descend" for v in a[1:])
    dims = [v for v in a[1:] if not isinstance(v, MStr)]
    d = iscalar(dims[0]) - 1 if dims else (0 if x.shape[0] != 1 else 1)
    if np.iscomplexobj(x):
        raise MError("sort of complex values is not supported")
    # stable in both directions: equal elements keep their original order (MATLAB's documented behaviour); NaNs last / first
    if desc:
        key = np.where(np.isnan(x), np.inf, -x)
        idx = np.argsort(key, axis=d, kind="stable")
        nan_first = np.isnan(np.take_along_axis(x, idx, axis=d))
        if np.any(nan_first):
            idx = np.argsort(np.where(np.isnan(x), -np.inf, -x), axis=d, kind="stable")
    else:
        idx = np.argsort(x, axis=d, kind="stable")
    s = np.take_along_axis(x, idx, axis=d)
    if nargout > 1:
        return s, idx.astype(np.float64) + 1.0
    return s


@builtin("find")
def _find(I, a, nargout):
    x = num(a[0])
    nz = np.flatnonzero(x.reshape(-1, order="F") != 0)
    if len(a) > 1:
        k = iscalar(a[1])
        if len(a) > 2 and isinstance(a[2], MStr) and a[2].s == "last":
            nz = nz[-k:]
        else:
            nz = nz[:k]
    if nargout >= 2:
        r, c = nz % x.shape[0], nz // x.shape[0]
        shape = (1, -1) if x.shape[0] == 1 else (-1, 1)
        return (r + 1.0).reshape(shape), (c + 1.0).reshape(shape)
    out = nz.astype(np.float64) + 1.0
    return out.reshape(1, -1) if (x.shape[0] == 1 and x.shape[1] != 1) else out.reshape(-1, 1) if x.size else np.zeros((0, 0) if x.shape == (0, 0) else ((1, 0) if x.shape[0] == 1 else (0, 1)))


@builtin("any")
def _any(I, a, n):
    x = num(a[0]) != 0
    if x.size == 0:
        return False
    return np.any(x, axis=_reduce_dim(x, a[1:]), keepdims=True)


@builtin("all")
def _all(I, a, n):
    x = num(a[0]) != 0
    if x.size == 0:
        return True
    return np.all(x, axis=_reduce_dim(x, a[1:]), keepdims=True)


@builtin("nnz")
def _nnz(I, a, n):
    return float(np.count_nonzero(num(a[0])))


def _size_of(v):
    if isinstance(v, MStr):
        return (1 if v.s else 0, len(v.s)) if v.s else (0, 0)
    if isinstance(v, MStruct):
        return (1, len(v.elems))
    if isinstance(v, MCell):
        return v.a.shape
    if isinstance(v, MFunc):
        return (1, 1)
    return num(v).shape


@builtin("size")
def _size(I, a, nargout):
    r, c = _size_of(a[0])
    if len(a) > 1:
        d = iscalar(a[1])
        return float((r, c)[d - 1]) if d <= 2 else 1.0
    if nargout <= 1:
        return np.array([[float(r), float(c)]])
    return tuple([float(r), float(c)] + [1.0] * (nargout - 2))


@builtin("length")
def _length(I, a, n):
    r, c = _size_of(a[0])
    return float(0 if r == 0 or c == 0 else max(r, c))


@builtin("numel")
def _numel(I, a, n):
    r, c = _size_of(a[0])
    return float(r * c)


@builtin("ndims")
def _ndims(I, a, n):
    return 2.0


@builtin("height")
def _height(I, a, n):
    """height(A): number of rows (size(A, 1)), also for plain arrays since R2020b - NAVdecoding.m loops `1:height(index)`."""
    return float(_size_of(a[0])[0])


@builtin("width")
def _width(I, a, n):
    return float(_size_of(a[0])[1])


@builtin("isempty")
def _isempty(I, a, n):
    r, c = _size_of(a[0])
    return r == 0 or c == 0


@builtin("isreal")
def _isreal(I, a, n):
    return not np.iscomplexobj(num(a[0]))


@builtin("isnumeric")
def _isnumeric(I, a, n):
    return isinstance(a[0], np.ndarray) and a[0].dtype != np.bool_


@builtin("ischar")
def _ischar(I, a, n):
    return isinstance(a[0], MStr)


@builtin("isstruct")
def _isstruct(I, a, n):
    return isinstance(a[0], MStruct)


@builtin("iscell")
def _iscell(I, a, n):
    return isinstance(a[0], MCell)


@builtin("islogical")
def _islogical(I, a, n):
    return isinstance(a[0], np.ndarray) and a[0].dtype == np.bool_


@builtin("isscalar")
def _isscalar(I, a, n):
    return _size_of(a[0]) == (1, 1)


@builtin("isvector")
def _isvector(I, a, n):
    r, c = _size_of(a[0])
    return (r == 1 or c == 1) and r * c >= 1


@builtin("isfield")
def _isfield(I, a, n):
    return isinstance(a[0], MStruct) and isinstance(a[1], MStr) and a[1].s in a[0].fields


@builtin("isequal")
def _isequal(I, a, n):
    def eq(x, y):
        if isinstance(x, MStr) and isinstance(y, MStr):
            return x.s == y.s
        if isinstance(x, MStruct) or isinstance(y, MStruct):
            if not (isinstance(x, MStruct) and isinstance(y, MStruct)) or len(x.elems) != len(y.elems) or set(x.fields) != set(y.fields):
                return False
            return all(eq(e1[f], e2[f]) for e1, e2 in zip(x.elems, y.elems) for f in x.fields)
        if isinstance(x, (MCell, MFunc)) or isinstance(y, (MCell, MFunc)):
            return False
        p, q = num(x), num(y)
        return p.shape == q.shape and bool(np.all(p == q))
    return all(eq(a[0], b) for b in a[1:])


@builtin("double", "single")
def _double(I, a, n):
    return fnum(a[0]).astype(np.complex128 if np.iscomplexobj(num(a[0])) else np.float64)


def _int_cast(lo, hi):
    def f(I, a, n):
        x = fnum(a[0])
        return np.clip(_round_half_away(x), lo, hi)      # saturating, round-to-nearest; kept in double storage
    return f


for _n, _lo, _hi in (("int8", -128, 127), ("uint8", 0, 255), ("int16", -32768, 32767), ("uint16", 0, 65535),
                     ("int32", -2 ** 31, 2 ** 31 - 1), ("uint32", 0, 2 ** 32 - 1), ("int64", -2 ** 63, 2 ** 63 - 1),
                     ("uint64", 0, 2 ** 64 - 1)):
    TABLE[_n] = _int_cast(_lo, _hi)


@builtin("logical")
def _logical(I, a, n):
    return num(a[0]) != 0


@builtin("char")
def _char(I, a, n):
    if isinstance(a[0], MStr):
        return a[0]
    x = num(a[0])
    return MStr("".join(chr(int(v)) for v in x.reshape(-1)))


@builtin("fft")
def _fft(I, a, n):
    x = fnum(a[0])
    nn = None
    if len(a) > 1 and num(a[1]).size:
        nn = iscalar(a[1])
    d = iscalar(a[2]) - 1 if len(a) > 2 else (0 if x.shape[0] != 1 else 1)
    return np.fft.fft(x, n=nn, axis=d)


@builtin("ifft")
def _ifft(I, a, n):
    x = fnum(a[0])
    nn = None
    if len(a) > 1 and not isinstance(a[1], MStr) and num(a[1]).size:
        nn = iscalar(a[1])
    d = iscalar(a[2]) - 1 if len(a) > 2 and not isinstance(a[2], MStr) else (0 if x.shape[0] != 1 else 1)
    return np.fft.ifft(x, n=nn, axis=d)


@builtin("fftshift")
def _fftshift(I, a, n):
    return np.fft.fftshift(fnum(a[0]))


@builtin("circshift")
def _circshift(I, a, n):
    x = a[0].codes() if isinstance(a[0], MStr) else num(a[0])
    k = num(a[1]).reshape(-1)
    if len(a) > 2:
        r = np.roll(x, int(k[0]), axis=iscalar(a[2]) - 1)
    elif k.size >= 2:
        r = np.roll(np.roll(x, int(k[0]), axis=0), int(k[1]), axis=1)
    else:
        # scalar shift: along the first dimension whose size is not 1
        r = np.roll(x, int(k[0]), axis=0 if x.shape[0] != 1 else 1)
    return r


@builtin("repmat")
def _repmat(I, a, n):
    r, c = _dims(a[1:])
    v = a[0]
    if isinstance(v, MStruct):
        if r != 1:
            raise MError("struct arrays are 1 x n")
        out = []
        for _ in range(c):
            out.extend(mcopy(v).elems)
        return MStruct(out, v.fields)
    if isinstance(v, MStr):
        if r == 1:
            return MStr(v.s * c)
        return np.tile(v.codes(), (r, c))
    if isinstance(v, MCell):
        return MCell(np.tile(v.a, (r, c)))
    return np.tile(num(v), (r, c))


@builtin("reshape")
def _reshape(I, a, n):
    x = num(a[0])
    if len(a) == 2:
        dims = [int(v) for v in num(a[1]).reshape(-1)]
    else:
        dims = []
        for v in a[1:]:
            dims.append(None if (isinstance(v, np.ndarray) and v.size == 0) else iscalar(v))
        if None in dims:
            known = int(np.prod([d for d in dims if d is not None]))
            dims[dims.index(None)] = x.size // known
    while len(dims) > 2 and dims[-1] == 1:
        dims.pop()
    if len(dims) != 2:
        raise MError("arrays of more than two dimensions are not supported")
    if dims[0] * dims[1] != x.size:
        raise MError("reshape: the number of elements must not change")
    return x.reshape(dims, order="F").copy()


@builtin("fliplr")
def _fliplr(I, a, n):
    return num(a[0])[:, ::-1].copy()


@builtin("flipud")
def _flipud(I, a, n):
    return num(a[0])[::-1, :].copy()


@builtin("flip")
def _flip(I, a, n):
    x = num(a[0])
    d = iscalar(a[1]) - 1 if len(a) > 1 else (0 if x.shape[0] != 1 else 1)
    return np.flip(x, axis=d).copy()


@builtin("linspace")
def _linspace(I, a, n):
    k = iscalar(a[2]) if len(a) > 2 else 100
    return np.linspace(scalar(a[0]), scalar(a[1]), k).reshape(1, -1)


@builtin("kron")
def _kron(I, a, n):
    return np.kron(fnum(a[0]), fnum(a[1]))


@builtin("norm")
def _norm(I, a, n):
    x = fnum(a[0])
    if 1 in x.shape:
        return float(np.linalg.norm(x.reshape(-1)))
    return float(np.linalg.norm(x, 2))


@builtin("dot")
def _dot(I, a, n):
    return np.sum(np.conj(fnum(a[0]).reshape(-1)) * fnum(a[1]).reshape(-1))


@builtin("conv")
def _conv(I, a, n):
    x, y = fnum(a[0]), fnum(a[1])
    r = np.convolve(x.reshape(-1), y.reshape(-1))
    return r.reshape(-1, 1) if (x.shape[1] == 1 and x.shape[0] > 1) else r.reshape(1, -1)


@builtin("unique")
def _unique(I, a, n):
    x = num(a[0])
    u = np.unique(x.reshape(-1))
    return u.reshape(1, -1) if x.shape[0] == 1 else u.reshape(-1, 1)


@builtin("ismember")
def _ismember(I, a, n):
    return np.isin(num(a[0]), num(a[1]))


@builtin("numel")
def _numel2(I, a, n):
    r, c = _size_of(a[0])
    return float(r * c)


@builtin("factor")
def _factor(I, a, n):
    v = iscalar(a[0])
    out, p = [], 2
    while p * p <= v:
        while v % p == 0:
            out.append(float(p))
            v //= p
        p += 1
    if v > 1 or not out:
        out.append(float(v))
    return np.array([out])


@builtin("isprime")
def _isprime(I, a, n):
    def ip(v):
        v = int(v)
        if v < 2:
            return False
        return all(v % p for p in range(2, int(math.isqrt(v)) + 1))
    return np.vectorize(ip)(fnum(a[0])).astype(bool)


@builtin("gcd")
def _gcd(I, a, n):
    return np.gcd(fnum(a[0]).astype(np.int64), fnum(a[1]).astype(np.int64)).astype(np.float64)


# ---- strings ------------------------------------------------------------------------------------------
def _to_py(v):
    if isinstance(v, MStr):
        return v.s
    x = num(v)
    if x.size == 1:
        s = x.flat[0]
        if isinstance(s, (bool, np.bool_)):
            return int(s)
        if np.iscomplexobj(x):
            return complex(s)
        return float(s)
    return x


@builtin("int2str")
def _int2str(I, a, n):
    x = _round_half_away(fnum(a[0]))
    if x.size == 1:
        v = float(x.flat[0].real)
        if math.isnan(v) or math.isinf(v):
            return MStr("NaN" if math.isnan(v) else ("Inf" if v > 0 else "-Inf"))
        return MStr(str(int(v)))
    return MStr("  ".join(str(int(v)) for v in x.reshape(-1)))


def _fmt_g(v, prec=5):
    if v == int(v) and abs(v) < 1e15:
        return str(int(v))
    return ("%." + str(prec) + "g") % v


@builtin("num2str")
def _num2str(I, a, n):
    if isinstance(a[0], MStr):
        return a[0]
    x = fnum(a[0])
    if len(a) > 1 and isinstance(a[1], MStr):
        return MStr(_sprintf(a[1].s, [a[0]]))
    prec = iscalar(a[1]) if len(a) > 1 else None
    vals = []
    for v in x.reshape(-1):
        v = float(v.real)
        if prec is not None:
            vals.append(("%." + str(prec) + "g") % v)
        elif v == int(v):
            vals.append(str(int(v)))
        else:
            vals.append(("%.4f" % v).rstrip("0").rstrip(".") if abs(v) < 1e5 else "%.5g" % v)
    return MStr("  ".join(vals))


def _sprintf(fmt, args):
    """MATLAB sprintf: the format is recycled over the (flattened) arguments."""
    import re
    flat = []
    for v in args:
        if isinstance(v, MStr):
            flat.append(v.s)
        else:
            flat.extend([float(t.real) if not isinstance(t, (bool, np.bool_)) else float(t) for t in num(v).reshape(-1, order="F")])
    fmt = fmt.replace("\\n", "\n").replace("\\t", "\t").replace("\\\\", "\\")
    spec = re.compile(r"%(?:%|[-+ 0#]*\d*(?:\.\d+)?[diufeEgGxXcs])")
    pieces = spec.findall(fmt)
    nspec = len([p for p in pieces if p != "%%"])
    out = []
    pos = 0
    first = True
    while first or (pos < len(flat) and nspec):
        first = False

        def rep(m):
            nonlocal pos
            s = m.group(0)
            if s == "%%":
                return "%"
            if pos >= len(flat):
                return ""
            v = flat[pos]
            pos += 1
            conv = s[-1]
            if conv in "di":
                if isinstance(v, str):
                    return v
                if v == int(v):
                    return (s[:-1] + "d") % int(v)
                return (s[:-1] + "e") % v
            if conv == "u":
                return (s[:-1] + "d") % int(v) if not isinstance(v, str) else v
            if conv in "xX":
                return (s[:-1] + conv) % int(v)
            if conv == "c":
                return v if isinstance(v, str) else chr(int(v))
            if conv == "s":
                if isinstance(v, str):
                    return (s) % v
                return _fmt_g(v)
            if isinstance(v, str):
                return v
            return s % v
        out.append(spec.sub(rep, fmt))
        if not nspec:
            break
    return "".join(out)


@builtin("sprintf")
def _sprintff(I, a, n):
    return MStr(_sprintf(a[0].s, a[1:]))


@builtin("fprintf")
def _fprintf(I, a, n):
    if a and not isinstance(a[0], MStr):
        a = a[1:]      # fid
    if a:
        I.out.append(_sprintf(a[0].s, a[1:]))
    return None


@builtin("disp", "display")
def _disp(I, a, n):
    v = a[0]
    I.out.append((v.s if isinstance(v, MStr) else str(_to_py(v))) + "\n")
    return None


@builtin("error")
def _error(I, a, n):
    if a and isinstance(a[0], MStr):
        msg = _sprintf(a[0].s, a[1:]) if len(a) > 1 else a[0].s
    else:
        msg = "error"
    raise MError(msg)


@builtin("warning")
def _warning(I, a, n):
    return None


@builtin("strcmp")
def _strcmp(I, a, n):
    x, y = a[0], a[1]
    if isinstance(x, MCell) or isinstance(y, MCell):       # strcmp(str, cellOfStr) -> logical array of the cell's shape
        c, s_ = (x, y) if isinstance(x, MCell) else (y, x)
        out = np.zeros(c.a.shape, dtype=bool)
        for idx, v in np.ndenumerate(c.a):
            if isinstance(s_, MCell):
                w = s_.a[idx] if s_.a.shape == c.a.shape else None
                out[idx] = isinstance(v, MStr) and isinstance(w, MStr) and v.s == w.s
            else:
                out[idx] = isinstance(v, MStr) and isinstance(s_, MStr) and v.s == s_.s
        return out
    return isinstance(x, MStr) and isinstance(y, MStr) and x.s == y.s


@builtin("strcmpi")
def _strcmpi(I, a, n):
    return isinstance(a[0], MStr) and isinstance(a[1], MStr) and a[0].s.lower() == a[1].s.lower()


@builtin("strncmp")
def _strncmp(I, a, n):
    k = iscalar(a[2])
    return isinstance(a[0], MStr) and isinstance(a[1], MStr) and a[0].s[:k] == a[1].s[:k] and len(a[0].s) >= k and len(a[1].s) >= k


@builtin("upper")
def _upper(I, a, n):
    return MStr(a[0].s.upper())


@builtin("lower")
def _lower(I, a, n):
    return MStr(a[0].s.lower())


@builtin("strtrim")
def _strtrim(I, a, n):
    return MStr(a[0].s.strip())


@builtin("strcat")
def _strcat(I, a, n):
    return MStr("".join(v.s for v in a))


@builtin("str2num", "str2double")
def _str2num(I, a, n):
    try:
        return float(a[0].s)
    except ValueError:
        return np.nan


@builtin("dec2bin")
def _dec2bin(I, a, n):
    x = fnum(a[0]).reshape(-1)
    width = iscalar(a[1]) if len(a) > 1 else 0
    strs = [format(int(v), "b") for v in x]
    w = max([width] + [len(s) for s in strs])
    strs = [s.rjust(w, "0") for s in strs]
    if len(strs) == 1:
        return MStr(strs[0])
    return np.array([[float(ord(c)) for c in s] for s in strs])


def _char_rows(v):
    if isinstance(v, MStr):
        return [v.s]
    x = num(v)
    return ["".join(chr(int(c)) for c in row) for row in x]


@builtin("bin2dec")
def _bin2dec(I, a, n):
    r = [float(int(s.replace(" ", ""), 2)) for s in _char_rows(a[0])]
    return np.array(r).reshape(-1, 1)


@builtin("base2dec")
def _base2dec(I, a, n):
    b = iscalar(a[1])
    r = [float(int(s.replace(" ", ""), b)) for s in _char_rows(a[0])]
    return np.array(r).reshape(-1, 1)


@builtin("hex2dec")
def _hex2dec(I, a, n):
    r = [float(int(s.strip(), 16)) for s in _char_rows(a[0])]
    return np.array(r).reshape(-1, 1)


@builtin("oct2dec")
def _oct2dec(I, a, n):
    """Communications Toolbox oct2dec: the decimal DIGITS of each element are read as an octal number."""
    x = fnum(a[0])
    return np.vectorize(lambda v: float(int(str(int(v)), 8)))(x).astype(np.float64)


@builtin("fliplr")
def _fliplr2(I, a, n):
    if isinstance(a[0], MStr):
        return MStr(a[0].s[::-1])
    return num(a[0])[:, ::-1].copy()


# ---- structs / cells ---------------------------------------------------------------------------------------
@builtin("struct")
def _struct(I, a, n):
    if len(a) % 2:
        raise MError("struct: field / value pairs expected")
    e = {}
    for k in range(0, len(a), 2):
        e[a[k].s] = a[k + 1]
    return MStruct([e], list(e.keys()))


@builtin("fieldnames")
def _fieldnames(I, a, n):
    c = np.empty((len(a[0].fields), 1), dtype=object)
    for k, f in enumerate(a[0].fields):
        c[k, 0] = MStr(f)
    return MCell(c)


@builtin("rmfield")
def _rmfield(I, a, n):
    s = mcopy(a[0])
    s.fields = [f for f in s.fields if f != a[1].s]
    for e in s.elems:
        e.pop(a[1].s, None)
    return s


@builtin("cell")
def _cell(I, a, n):
    r, c = _dims(a)
    out = np.empty((r, c), dtype=object)
    for i in np.ndindex(r, c):
        out[i] = empty()
    return MCell(out)


@builtin("feval")
def _feval(I, a, n):
    f = a[0]
    if isinstance(f, MFunc):
        r = f.fn(a[1:], n)
        return tuple(r) if len(r) != 1 else r[0]
    raise MError("feval needs a function handle")


# ---- file I/O ----------------------------------------------------------------------------------------------
class MFile:
    def __init__(self, data: bytes, name="<memory>"):
        self.data = data
        self.pos = 0
        self.name = name


def register_file(I, data: bytes, name="<memory>") -> float:
    fid = I.next_fid
    I.next_fid += 1
    I.files[fid] = MFile(data, name)
    return float(fid)


def _file(I, v):
    fid = int(scalar(v))
    if fid not in I.files:
        raise MError(f"Invalid file identifier {fid}")
    return I.files[fid]


@builtin("fopen")
def _fopen(I, a, nargout):
    if not isinstance(a[0], MStr):
        if nargout > 1:
            return MStr(I.files[int(scalar(a[0]))].name), MStr("rb")
        return MStr(I.files[int(scalar(a[0]))].name)
    name = a[0].s
    cands = [name] + [os.path.join(d, name) for d in I.path]
    for c in cands:
        if os.path.isfile(c):
            fid = register_file(I, open(c, "rb").read(), c)
            return (fid, MStr("")) if nargout > 1 else fid
    return (-1.0, MStr(f"cannot open {name}")) if nargout > 1 else -1.0


@builtin("dir")
def _dir(I, a, n):
    """dir(name) for one file: struct with the documented fields name, folder, date, bytes, isdir, datenum (a file registered
    with register_file answers from its in-memory content)."""
    name = a[0].s
    for f in I.files.values():
        if f.name == name:
            if os.path.isfile(name):
                break
            return MStruct([{"name": MStr(os.path.basename(name)), "folder": MStr(os.path.dirname(name)), "date": MStr(""),
                             "bytes": M(float(len(f.data))), "isdir": M(0.0), "datenum": M(0.0)}])
    if not os.path.isfile(name):
        return MStruct([])
    st = os.stat(name)
    return MStruct([{"name": MStr(os.path.basename(name)), "folder": MStr(os.path.dirname(os.path.abspath(name))), "date": MStr(""),
                     "bytes": M(float(st.st_size)), "isdir": M(0.0), "datenum": M(st.st_mtime / 86400.0 + 719529.0)}])


@builtin("fclose")
def _fclose(I, a, n):
    if a and not isinstance(a[0], MStr):
        I.files.pop(int(scalar(a[0])), None)
    return 0.0


@builtin("fseek")
def _fseek(I, a, n):
    f = _file(I, a[0])
    off = scalar(a[1])
    if off != math.floor(off):
        raise MError(f"fseek: offset {off} is not a whole number of bytes")
    origin = a[2].s if (len(a) > 2 and isinstance(a[2], MStr)) else ("bof" if len(a) < 3 else {-1: "bof", 0: "cof", 1: "eof"}[int(scalar(a[2]))])
    base = {"bof": 0, "cof": f.pos, "eof": len(f.data)}[origin]
    p = base + int(off)
    if p < 0 or p > len(f.data):
        return -1.0
    f.pos = p
    return 0.0


@builtin("ftell")
def _ftell(I, a, n):
    return float(_file(I, a[0]).pos)


@builtin("feof")
def _feof(I, a, n):
    f = _file(I, a[0])
    return f.pos >= len(f.data)


_PREC = {"schar": np.int8, "int8": np.int8, "char": np.int8, "uchar": np.uint8, "uint8": np.uint8, "int16": np.dtype("<i2"),
         "uint16": np.dtype("<u2"), "int32": np.dtype("<i4"), "uint32": np.dtype("<u4"), "float32": np.dtype("<f4"),
         "single": np.dtype("<f4"), "double": np.dtype("<f8"), "float64": np.dtype("<f8"), "int64": np.dtype("<i8")}


@builtin("fread")
def _fread(I, a, nargout):
    f = _file(I, a[0])
    prec = "uint8"
    count = math.inf
    rest = a[1:]
    if rest and not isinstance(rest[0], MStr):
        cv = num(rest[0]).reshape(-1)
        count = float(np.prod(cv)) if cv.size else math.inf
        shape2 = cv if cv.size == 2 else None
        rest = rest[1:]
    else:
        shape2 = None
    if rest and isinstance(rest[0], MStr):
        prec = rest[0].s.split("=>")[0].strip().lstrip("*")
    if prec not in _PREC:
        raise MError(f"fread: precision '{prec}' is not supported")
    if isinstance(count, float) and count != math.inf and (math.isnan(count) or count < 0 or count != math.floor(count)):
        raise MError(f"fread: invalid size {count}")
    dt = np.dtype(_PREC[prec])
    avail = (len(f.data) - f.pos) // dt.itemsize
    k = int(avail if count == math.inf else min(avail, int(count)))
    vals = np.frombuffer(f.data, dtype=dt, count=k, offset=f.pos).astype(np.float64)
    f.pos += k * dt.itemsize
    if shape2 is not None and k:
        r = int(shape2[0])
        c = -(-k // r)
        buf = np.zeros(r * c)
        buf[:k] = vals
        out = buf.reshape((r, c), order="F")
    else:
        out = vals.reshape(-1, 1)
    return (out, float(k)) if nargout > 1 else out


@builtin("fscanf")
def _fscanf(I, a, nargout):
    f = _file(I, a[0])
    fmt = a[1].s
    text = f.data[f.pos:].decode("latin-1")
    count = math.inf
    if len(a) > 2:
        cv = num(a[2]).reshape(-1)
        count = float(np.prod(cv))
    import re
    if fmt.strip() in ("%d", "%f", "%g", "%i", "%u", "%e"):
        toks = re.findall(r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)", text)
        if count != math.inf:
            toks = toks[:int(count)]
        f.pos = len(f.data)
        out = np.array([float(t) for t in toks]).reshape(-1, 1)
    elif fmt.strip() in ("%c", "%s"):
        s = text if fmt.strip() == "%c" else "".join(text.split())
        if count != math.inf:
            s = s[:int(count)]
        f.pos += len(s)
        out = MStr(s)
        return (out, float(len(s))) if nargout > 1 else out
    elif fmt.strip() == "%1d":
        digs = [float(c) for c in text if c.isdigit()]
        if count != math.inf:
            digs = digs[:int(count)]
        f.pos = len(f.data)
        out = np.array(digs).reshape(-1, 1)
    else:
        raise MError(f"fscanf: format '{fmt}' is not supported")
    if len(a) > 2 and num(a[2]).size == 2 and out.size:
        r = int(num(a[2]).reshape(-1)[0])
        out = out.reshape(-1)[: (out.size // r) * r].reshape((r, -1), order="F")
    return (out, float(out.size)) if nargout > 1 else out


@builtin("fgetl")
def _fgetl(I, a, n):
    f = _file(I, a[0])
    if f.pos >= len(f.data):
        return -1.0
    e = f.data.find(b"\n", f.pos)
    e = len(f.data) if e < 0 else e
    s = f.data[f.pos:e].decode("latin-1").rstrip("\r")
    f.pos = min(len(f.data), e + 1)
    return MStr(s)


@builtin("load")
def _load(I, a, n):
    name = a[0].s
    for c in [name] + [os.path.join(d, name) for d in I.path]:
        if os.path.isfile(c):
            return np.loadtxt(c, ndmin=2)
    raise MError(f"load: cannot find {name}")


# ---- environment / GUI stubs ----------------------------------------------------------------------------------
@builtin("tic")
def _tic(I, a, n):
    I.tic = time.perf_counter()
    return float(I.tic) if n else None


@builtin("toc")
def _toc(I, a, n):
    return float(time.perf_counter() - (I.tic or time.perf_counter()))


@builtin("now", "clock", "cputime")
def _now(I, a, n):
    return 0.0


@builtin("datestr")
def _datestr(I, a, n):
    return MStr("00:00:00")


@builtin("waitbar")
def _waitbar(I, a, n):
    return 1.0


@builtin("get")
def _get(I, a, n):
    return np.array([[0.0, 0.0, 360.0, 75.0]])


for _n in ("set", "close", "delete", "figure", "plot", "subplot", "title", "xlabel", "ylabel", "zlabel", "legend", "grid",
           "axis", "hold", "drawnow", "bar", "text", "pause", "save", "clf", "colormap", "mesh", "surf", "view", "format",
           "addpath", "clc", "beep", "box", "xlim", "ylim", "shading", "annotation", "movegui", "print", "saveas", "clear",
           "rmpath", "more", "input", "keyboard", "uiwait", "msgbox", "colorbar", "stem", "semilogy", "polar", "line", "fill"):
    TABLE[_n] = (lambda I, a, n: None)


@builtin("lasterror")
def _lasterror(I, a, n):
    return MStruct([{"message": MStr(""), "identifier": MStr("")}])


# ---- numerical toolboxes ------------------------------------------------------------------------------------------
@builtin("integral", "quad", "quadl", "quadgk")
def _integral(I, a, n):
    from scipy.integrate import quad
    f = a[0]
    lo, hi = scalar(a[1]), scalar(a[2])

    def g(x):
        r = f.fn([M(float(x))], 1)[0]
        return float(num(r).flat[0].real)
    # MATLAB's integral() starts from ten equal subintervals and evaluates Gauss-Kronrod nodes strictly inside them, so an
    # integrand that is 0/0 at the centre of [lo, hi] (CalcWeighingFactor.m at f = 0) is never evaluated there; QUADPACK on the
    # whole interval would hit the centre with its first node.
    edges = np.linspace(lo, hi, 11)
    total = 0.0
    for k in range(10):
        total += quad(g, float(edges[k]), float(edges[k + 1]), limit=200, epsabs=1e-10, epsrel=1e-6)[0]   # AbsTol 1e-10, RelTol 1e-6
    return float(total)


@builtin("fir1")
def _fir1(I, a, n):
    from scipy.signal import firwin
    order = iscalar(a[0])
    wn = num(a[1]).reshape(-1)
    ftype = a[2].s.lower() if len(a) > 2 and isinstance(a[2], MStr) else None
    if wn.size == 1:
        pass_zero = ftype != "high"
        taps = firwin(order + 1, float(wn[0]), window="hamming", pass_zero=pass_zero, scale=True)
    else:
        pass_zero = ftype == "stop"
        taps = firwin(order + 1, [float(w) for w in wn], window="hamming", pass_zero=pass_zero, scale=True)
    return taps.reshape(1, -1)


@builtin("filtfilt")
def _filtfilt(I, a, n):
    from scipy.signal import filtfilt
    b, den, x = fnum(a[0]).reshape(-1), fnum(a[1]).reshape(-1), fnum(a[2])
    ax = 0 if x.shape[0] != 1 else 1
    nfilt = max(b.size, den.size)
    y = filtfilt(b, den, x, axis=ax, padtype="odd", padlen=3 * (nfilt - 1))
    return y


@builtin("filter")
def _filter(I, a, n):
    from scipy.signal import lfilter
    b, den, x = fnum(a[0]).reshape(-1), fnum(a[1]).reshape(-1), fnum(a[2])
    return lfilter(b, den, x, axis=0 if x.shape[0] != 1 else 1)


@builtin("xcorr")
def _xcorr(I, a, n):
    x = fnum(a[0]).reshape(-1)
    y = fnum(a[1]).reshape(-1) if len(a) > 1 and not isinstance(a[1], MStr) and num(a[1]).size > 1 else x
    m = max(x.size, y.size)
    xp = np.concatenate([x, np.zeros(m - x.size)])
    yp = np.concatenate([y, np.zeros(m - y.size)])
    r = np.correlate(xp, yp, mode="full")
    return r.reshape(1, -1) if num(a[0]).shape[0] == 1 else r.reshape(-1, 1)
