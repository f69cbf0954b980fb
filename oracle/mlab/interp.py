"""Tree-walking evaluator of the mini-MATLAB interpreter (oracle/mlab/__init__.py has the why and the scope)."""
from __future__ import annotations

import math
import os

import numpy as np

from . import builtins as B
from .parser import parse
from .values import (M, MCell, MError, MFunc, MStr, MStruct, empty, fnum, is_true, iscalar, mcopy, num, scalar, shape_str)


class _Break(Exception):
    pass


class _Continue(Exception):
    pass


class _Return(Exception):
    pass


class UserFunction:
    def __init__(self, node, fname, local_funcs):
        _, self.name, self.params, self.outs, self.body, self.line = node
        self.fname = fname
        self.local_funcs = local_funcs
        self.mutated = _assigned_names(self.body)   # parameters written inside the body are copied on entry


def _assigned_names(body) -> set:
    out = set()

    def base(e):
        while e[0] in ("index", "field", "dynfield"):
            e = e[1]
        return e[1] if e[0] == "id" else None

    def walk(stmts):
        for s in stmts:
            k = s[0]
            if k == "assign":
                for l in s[1]:
                    if l[0] != "tilde":
                        n = base(l)
                        if n:
                            out.add(n)
            elif k == "if":
                for _, b in s[1]:
                    walk(b)
                if s[2]:
                    walk(s[2])
            elif k in ("for",):
                out.add(s[1])
                walk(s[3])
            elif k == "while":
                walk(s[2])
            elif k == "switch":
                for _, b in s[2]:
                    walk(b)
                if s[3]:
                    walk(s[3])
            elif k == "try":
                walk(s[1])
                walk(s[3])
    walk(body)
    return out


class Interpreter:
    """One MATLAB "session": a search path (the package's include/ and Common/ directories), a file table, a global
    workspace.  call(name, *args, nargout=1) runs a function file of the path."""

    def __init__(self, path, stdout=None, quiet=True):
        self.path = list(path)
        self.func_cache = {}
        self.files = {}           # fid -> MFile
        self.next_fid = 3
        self.globals = {}
        self.quiet = quiet
        self.out = [] if stdout is None else stdout
        self.call_depth = 0
        self.stats = {"calls": {}}
        self.tic = None
        self.extra_builtins = {}  # name -> fn(interp, args, nargout): test doubles / host bindings of one session
        self.persist = {}         # (file, variable) -> value of `persistent` variables

    # ---- function lookup ------------------------------------------------------------------------
    def find_function(self, name, local_funcs=None):
        if local_funcs and name in local_funcs:
            return local_funcs[name]
        if name in self.func_cache:
            return self.func_cache[name]
        for d in self.path:
            f = os.path.join(d, name + ".m")
            if os.path.exists(f):
                funcs, script = parse(open(f, encoding="latin-1").read(), f)
                if not funcs:
                    fn = ("script", script, f)
                    self.func_cache[name] = fn
                    return fn
                local = {}
                ufs = [UserFunction(n, f, local) for n in funcs]
                for u in ufs:
                    local[u.name] = u
                self.func_cache[name] = ufs[0]
                return ufs[0]
        self.func_cache[name] = None
        return None

    def call(self, name, *args, nargout=1):
        fn = self.find_function(name)
        if fn is None:
            raise MError(f"Undefined function '{name}'")
        res = self.call_user(fn, [M(a) for a in args], nargout)
        return res[0] if nargout == 1 else tuple(res[:nargout])

    def run_script(self, fname, ws=None):
        funcs, script = parse(open(fname, encoding="latin-1").read(), fname)
        ws = {} if ws is None else ws
        frame = Frame(self, ws, {}, fname)
        try:
            frame.exec_block(script)
        except _Return:
            pass
        return ws

    def run_lines(self, fname, ranges, ws=None):
        """Executes the statements of `fname` that lie on the given 1-based inclusive line ranges, as a script in `ws` - a SECTION
        of a reference function whose other parts need toolboxes this interpreter does not restate (the bit-synchronisation
        block of NAVdecoding.m without its Viterbi / CRC / ephemeris tail).  Lines outside the ranges are blanked, so line
        numbers in error messages stay the file's own."""
        lines = open(fname, encoding="latin-1").read().splitlines()
        keep = [False] * len(lines)
        for a, b in ranges:
            for k in range(a - 1, min(b, len(lines))):
                keep[k] = True
        text = "\n".join(ln if k else "" for ln, k in zip(lines, keep)) + "\n"
        funcs, script = parse(text, fname)
        if funcs:
            raise MError("run_lines: the ranges must not contain a function header")
        ws = {} if ws is None else ws
        frame = Frame(self, ws, {}, fname)
        try:
            frame.exec_block(script)
        except _Return:
            pass
        return ws

    def call_user(self, fn, args, nargout):
        if isinstance(fn, tuple) and fn[0] == "script":
            raise MError("scripts cannot be called with arguments")
        self.stats["calls"][fn.name] = self.stats["calls"].get(fn.name, 0) + 1
        ws = {}
        fixed = fn.params
        if fn.params and fn.params[-1] == "varargin":
            fixed = fn.params[:-1]
            extra = list(args[len(fixed):])
            box = np.empty((1, len(extra)), dtype=object)
            for k, a in enumerate(extra):
                box[0, k] = a
            ws["varargin"] = MCell(box)
        elif len(args) > len(fn.params):
            raise MError(f"{fn.name}: too many input arguments")
        for p, a in zip(fixed, args):
            if p != "~":
                ws[p] = mcopy(a) if p in fn.mutated else a
        ws["nargin"] = M(float(len(args)))
        ws["nargout"] = M(float(nargout))
        frame = Frame(self, ws, fn.local_funcs, fn.fname)
        frame.func_name = fn.name
        self.call_depth += 1
        if self.call_depth > 200:
            raise MError("recursion limit")
        try:
            frame.exec_block(fn.body)
        except _Return:
            pass
        finally:
            self.call_depth -= 1
        outs = []
        for k, o in enumerate(fn.outs):
            if o not in ws:
                if k < max(nargout, 1):
                    raise MError(f"{fn.name}: output argument '{o}' not assigned")
                break
            v = ws[o]
            if any(v is a for a in args):
                v = mcopy(v)
            outs.append(v)
        return outs


class Frame:
    def __init__(self, interp: Interpreter, ws: dict, local_funcs, fname):
        self.I = interp
        self.ws = ws
        self.local_funcs = local_funcs
        self.fname = fname
        self.end_stack = []     # (value being indexed, position, number of index arguments)
        self.global_names = {}  # local name -> key in Interpreter.globals (global: the name; persistent: (file, function, name))
        self.func_name = ""

    # ---- statements -------------------------------------------------------------------------------
    def exec_block(self, stmts):
        for s in stmts:
            self.exec_stmt(s)

    def exec_stmt(self, s):
        k = s[0]
        try:
            if k == "assign":
                self.exec_assign(s)
            elif k == "expr":
                e = s[1]
                # a bare identifier / call evaluated for its side effect; `ans` is not kept
                self.eval_multi(e, 0)
            elif k == "if":
                for cond, body in s[1]:
                    if is_true(self.eval(cond)):
                        self.exec_block(body)
                        break
                else:
                    if s[2] is not None:
                        self.exec_block(s[2])
            elif k == "for":
                self.exec_for(s)
            elif k == "while":
                while is_true(self.eval(s[1])):
                    try:
                        self.exec_block(s[2])
                    except _Break:
                        break
                    except _Continue:
                        continue
            elif k == "switch":
                v = self.eval(s[1])
                for ce, body in s[2]:
                    c = self.eval(ce)
                    if self._switch_match(v, c):
                        self.exec_block(body)
                        break
                else:
                    if s[3] is not None:
                        self.exec_block(s[3])
            elif k == "try":
                try:
                    self.exec_block(s[1])
                except (_Break, _Continue, _Return):
                    raise
                except MError as ex:
                    if s[2]:
                        self.ws[s[2]] = MStruct([{"message": MStr(str(ex)), "identifier": MStr("")}])
                    self.exec_block(s[3])
            elif k == "break":
                raise _Break()
            elif k == "continue":
                raise _Continue()
            elif k == "return":
                raise _Return()
            elif k == "global":
                for n in s[1]:
                    key = n if s[2] == "global" else (self.fname, self.func_name, n)
                    self.global_names[n] = key
                    if key not in self.I.globals:
                        self.I.globals[key] = empty()
                    self.ws[n] = self.I.globals[key]
            elif k == "command":
                pass
            else:
                raise MError(f"unknown statement {k}")
        except MError as ex:
            if not getattr(ex, "located", False):
                ex.args = (f"{ex.args[0]}\n    at {self.fname}:{s[-1]}",)
                ex.located = True
            raise

    def _switch_match(self, v, c):
        if isinstance(c, MCell):
            return any(self._switch_match(v, x) for x in c.a.flat)
        if isinstance(v, MStr) or isinstance(c, MStr):
            return isinstance(v, MStr) and isinstance(c, MStr) and v.s == c.s
        return bool(np.all(num(v) == num(c)))

    def exec_for(self, s):
        _, var, e, body, _ = s
        if e[0] == "range":
            # iterate lazily over the colon vector (same element values as the materialised one)
            vec = self.eval(e)
        else:
            vec = self.eval(e)
        if isinstance(vec, MStruct):
            cols = [MStruct([el], vec.fields) for el in vec.elems]
        elif isinstance(vec, MCell):
            cols = [MCell(vec.a[:, j:j + 1]) for j in range(vec.a.shape[1])]
        else:
            a = num(vec)
            if a.shape[0] == 1:
                cols = None
                for j in range(a.shape[1]):
                    self.ws[var] = a[:, j:j + 1].copy()
                    try:
                        self.exec_block(body)
                    except _Break:
                        break
                    except _Continue:
                        continue
                return
            cols = [a[:, j:j + 1].copy() for j in range(a.shape[1])]
        for c in cols:
            self.ws[var] = c
            try:
                self.exec_block(body)
            except _Break:
                break
            except _Continue:
                continue

    # ---- assignment ---------------------------------------------------------------------------------
    def exec_assign(self, s):
        _, lhs, rhs, _ = s
        if len(lhs) == 1:
            if rhs[0] == "matrix" and not rhs[1] and lhs[0][0] == "index" and lhs[0][3] == "()":
                self.delete_elements(lhs[0])          # x(idx) = []
                return
            v = self.eval(rhs)
            if rhs[0] in ("id", "field", "dynfield", "paren") or (rhs[0] == "index" and rhs[3] == "{}"):
                v = mcopy(v)
            self.assign_to(lhs[0], v)
            return
        vals = self.eval_multi(rhs, len(lhs))
        if len(vals) < len([l for l in lhs if l[0] != "tilde"]) and len(vals) < len(lhs):
            raise MError("not enough output arguments")
        for l, v in zip(lhs, vals):
            if l[0] != "tilde":
                self.assign_to(l, v)

    def assign_to(self, target, value):
        accs = []
        e = target
        while e[0] in ("index", "field", "dynfield"):
            if e[0] == "index":
                accs.append((e[3], e[2]))
            elif e[0] == "field":
                accs.append((".", e[2]))
            else:
                accs.append((".", self._field_name(self.eval(e[2]))))
            e = e[1]
        if e[0] != "id":
            raise MError("invalid assignment target")
        name = e[1]
        accs.reverse()
        cur = self.ws.get(name)
        new = self.assign_into(cur, accs, value)
        self.ws[name] = new
        if name in self.global_names:
            self.I.globals[self.global_names[name]] = new

    @staticmethod
    def _field_name(v):
        if not isinstance(v, MStr):
            raise MError("dynamic field name must be a char vector")
        return v.s

    def assign_into(self, cur, accs, value):
        if not accs:
            return value
        kind, arg = accs[0]
        rest = accs[1:]
        if kind == ".":
            if cur is None or (isinstance(cur, np.ndarray) and cur.size == 0):
                cur = MStruct([dict()], [])
            if not isinstance(cur, MStruct):
                raise MError(f"field assignment to a non-struct ({shape_str(cur)})")
            if len(cur.elems) != 1:
                raise MError("field assignment needs a scalar struct (index the struct array first)")
            cur.add_field(arg)
            sub = cur.elems[0].get(arg)
            cur.elems[0][arg] = self.assign_into(sub if rest else None, rest, value)
            return cur
        if kind == "()":
            if not rest:
                idx = self.eval_index_args(cur, arg)
                return B.set_index(cur, idx, value)
            # s(k).field... : struct arrays (and cells of structs are not needed)
            if cur is None or (isinstance(cur, np.ndarray) and cur.size == 0):
                cur = MStruct([], [])
            if not isinstance(cur, MStruct):
                raise MError(f"()-indexing followed by more indexing on {shape_str(cur)}")
            idx = self.eval_index_args(cur, arg)
            k = B.struct_linear_index(idx, len(cur.elems)) if cur.elems else B.struct_linear_index(idx, 0)
            while len(cur.elems) <= k:
                cur.elems.append({f: empty() for f in cur.fields})
            elem = MStruct([cur.elems[k]], cur.fields)
            new = self.assign_into(elem, rest, value)
            cur.fields = new.fields
            for f in new.fields:
                for e in cur.elems:
                    e.setdefault(f, empty())
            cur.elems[k] = new.elems[0]
            return cur
        if kind == "{}":
            if cur is None or (isinstance(cur, np.ndarray) and cur.size == 0):
                cur = MCell()
            if not isinstance(cur, MCell):
                raise MError("{}-assignment to a non-cell")
            idx = self.eval_index_args(cur, arg)
            sub = None
            if rest:
                try:
                    sub = B.cell_get(cur, idx)
                except MError:
                    sub = None
            return B.cell_set(cur, idx, self.assign_into(sub, rest, value))
        raise MError("bad accessor")

    def delete_elements(self, target):
        name_e = target[1]
        if name_e[0] != "id":
            raise MError("element deletion is supported on plain variables only")
        cur = self.ws[name_e[1]]
        idx = self.eval_index_args(cur, target[2])
        self.ws[name_e[1]] = B.delete_index(cur, idx)

    # ---- expressions ----------------------------------------------------------------------------------
    def eval(self, e):
        r = self.eval_multi(e, 1)
        if not r:
            raise MError("expression produced no value")
        return r[0]

    def eval_index_args(self, obj, args):
        out = []
        n = len(args)
        for k, a in enumerate(args):
            if a[0] == "colon_all":
                out.append(B.COLON)
                continue
            self.end_stack.append((obj, k, n))
            try:
                v = self.eval(a)
            finally:
                self.end_stack.pop()
            out.append(v)
        return out

    def eval_multi(self, e, nargout):
        """Returns a list of values (function calls may return several)."""
        k = e[0]
        if k == "num":
            return [M(e[1])]
        if k == "str":
            return [MStr(e[1])]
        if k == "id":
            name = e[1]
            if name in self.ws:
                return [self.ws[name]]
            return self.call_function(name, [], nargout)
        if k == "paren":
            return [self.eval(e[1])]
        if k == "binop":
            op = e[1]
            if op == "&&":
                a = self.eval(e[2])
                if not is_true(a):
                    return [M(False)]
                return [M(is_true(self.eval(e[3])))]
            if op == "||":
                a = self.eval(e[2])
                if is_true(a):
                    return [M(True)]
                return [M(is_true(self.eval(e[3])))]
            return [B.binop(op, self.eval(e[2]), self.eval(e[3]))]
        if k == "unop":
            return [B.unop(e[1], self.eval(e[2]))]
        if k == "postfix":
            return [B.transpose(self.eval(e[2]), conj=(e[1] == "'"))]
        if k == "range":
            a = self.eval(e[1])
            b = self.eval(e[3])
            st = self.eval(e[2]) if e[2] is not None else None
            return [B.colon(a, st, b)]
        if k == "matrix":
            rows = [[v for x in row for v in self.eval_multi(x, 1)] for row in e[1]]     # [s.field] expands to a list
            return [B.concat(rows)]
        if k == "cell":
            rows = [[v for x in row for v in self.eval_multi(x, 1)] for row in e[1]]
            return [B.make_cell(rows)]
        if k == "end":
            if not self.end_stack:
                raise MError("'end' outside an index expression")
            obj, pos, n = self.end_stack[-1]
            return [M(float(B.end_value(obj, pos, n)))]
        if k == "colon_all":
            return [MStr(":")]
        if k == "index":
            return self.eval_index(e, nargout)
        if k == "field":
            base = self.eval(e[1])
            return B.get_field(base, e[2])
        if k == "dynfield":
            base = self.eval(e[1])
            return B.get_field(base, self._field_name(self.eval(e[2])))
        if k == "anon":
            params, body = e[1], e[2]
            captured = dict(self.ws)

            def fn(args, nargout_, _params=params, _body=body, _cap=captured, _self=self):
                ws = dict(_cap)
                for p, a in zip(_params, args):
                    ws[p] = a
                fr = Frame(_self.I, ws, _self.local_funcs, _self.fname)
                return fr.eval_multi(_body, max(nargout_, 1))
            return [MFunc(fn, "@anon")]
        if k == "fhandle":
            name = e[1]
            return [MFunc(lambda args, nargout_, _n=name: self.call_function(_n, args, max(nargout_, 1)), "@" + name)]
        raise MError(f"cannot evaluate node {k}")

    def eval_index(self, e, nargout):
        _, base_e, args, kind = e
        if base_e[0] == "id" and base_e[1] not in self.ws:
            if kind != "()":
                raise MError(f"Undefined variable '{base_e[1]}'")
            # function call; `end` is not allowed in its arguments, ':' is passed as the char ':'
            vals = []
            for a in args:
                if a[0] == "colon_all":
                    vals.append(MStr(":"))
                else:
                    vals.extend(self.eval_multi(a, 1)[:1])
            return self.call_function(base_e[1], vals, nargout)
        base = self.eval(base_e)
        if isinstance(base, MFunc):
            vals = [self.eval(a) for a in args]
            return base.fn(vals, nargout)
        idx = self.eval_index_args(base, args)
        if kind == "{}":
            return [B.cell_get(base, idx)]
        return [B.get_index(base, idx)]

    def call_function(self, name, args, nargout):
        fn = None if name in self.I.extra_builtins else self.I.find_function(name, self.local_funcs)
        if fn is not None:
            if isinstance(fn, tuple):   # a script: runs in the caller's workspace
                try:
                    self.exec_block(fn[1])
                except _Return:
                    pass
                return []
            return self.I.call_user(fn, args, nargout)
        if name == "exist":
            what = args[0].s if args and isinstance(args[0], MStr) else ""
            kind = args[1].s if len(args) > 1 and isinstance(args[1], MStr) else ""
            if what in self.ws and kind in ("", "var"):
                return [M(1.0)]
            if kind in ("", "file", "builtin") and (self.I.find_function(what, self.local_funcs) is not None or what in B.TABLE):
                return [M(2.0)]
            return [M(0.0)]
        bf = self.I.extra_builtins.get(name) or B.TABLE.get(name)
        if bf is None:
            raise MError(f"Undefined function or variable '{name}'")
        r = bf(self.I, args, nargout)
        if r is None:
            return []
        if isinstance(r, tuple):
            return [M(x) for x in r]
        return [M(r)]
