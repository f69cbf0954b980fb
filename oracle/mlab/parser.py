"""Recursive-descent parser for the MATLAB subset of the reference's hot-path files -> tuple AST (see interp.py)."""
from __future__ import annotations

from .lexer import Tok, tokenize

COMMAND_WORDS = {"format", "addpath", "close", "clc", "clear", "hold", "warning", "figure", "more", "rmpath", "drawnow"}


class ParseError(Exception):
    pass


class Parser:
    def __init__(self, src: str, fname: str = "<src>"):
        self.toks = tokenize(src, fname)
        self.i = 0
        self.fname = fname
        self.in_matrix = 0      # depth of [] / {} literals: whitespace separates elements there
        self.in_index = 0       # depth of () / {} index lists: `end` is a value there

    # ---- token helpers ------------------------------------------------------------------
    @property
    def t(self) -> Tok:
        return self.toks[self.i]

    def peek(self, k=1) -> Tok:
        return self.toks[min(self.i + k, len(self.toks) - 1)]

    def err(self, msg):
        raise ParseError(f"{self.fname}:{self.t.line}: {msg} (at {self.t.kind} {self.t.val!r})")

    def is_op(self, *ops):
        return self.t.kind == "OP" and self.t.val in ops

    def is_kw(self, *kws):
        return self.t.kind == "KW" and self.t.val in kws

    def eat_op(self, op):
        if not self.is_op(op):
            self.err(f"expected {op!r}")
        self.i += 1

    def skip_nl(self):
        while self.t.kind == "NL" or self.is_op(";", ","):
            self.i += 1

    # ---- file level -----------------------------------------------------------------------
    def parse_file(self):
        """Returns (functions: list of ('function', name, params, outs, body), script_body or None)."""
        self.skip_nl()
        funcs = []
        if self.is_kw("function"):
            while self.is_kw("function"):
                funcs.append(self.parse_function())
                self.skip_nl()
            if self.t.kind != "EOF":
                self.err("statements after the function definitions")
            return funcs, None
        body = self.parse_block(("EOF",))
        return funcs, body

    def parse_function(self):
        line = self.t.line
        self.i += 1  # function
        outs = []
        # forms: function name | function name(args) | function out = name(args) | function [o1, o2] = name(args)
        if self.is_op("["):
            self.i += 1
            while not self.is_op("]"):
                if self.is_op(","):
                    self.i += 1
                    continue
                if self.t.kind != "ID":
                    self.err("output name expected")
                outs.append(self.t.val)
                self.i += 1
            self.i += 1
            self.eat_op("=")
            name = self.t.val
            self.i += 1
        else:
            name = self.t.val
            self.i += 1
            if self.is_op("="):
                outs = [name]
                self.i += 1
                name = self.t.val
                self.i += 1
        params = []
        if self.is_op("("):
            self.i += 1
            while not self.is_op(")"):
                if self.is_op(","):
                    self.i += 1
                    continue
                if self.is_op("~"):
                    params.append("~")
                elif self.t.kind == "ID":
                    params.append(self.t.val)
                else:
                    self.err("parameter name expected")
                self.i += 1
            self.i += 1
        body = self.parse_block(("end", "function", "EOF"))
        if self.is_kw("end"):
            self.i += 1
        return ("function", name, params, outs, body, line)

    # ---- statements -------------------------------------------------------------------------
    def parse_block(self, stops):
        """Statements until one of the stop keywords (not consumed) or EOF."""
        body = []
        while True:
            self.skip_nl()
            if self.t.kind == "EOF":
                if "EOF" in stops:
                    return body
                self.err("unexpected end of file")
            if self.t.kind == "KW" and self.t.val in stops:
                return body
            body.append(self.parse_statement())

    def parse_statement(self):
        t = self.t
        line = t.line
        if t.kind == "KW":
            kw = t.val
            if kw == "if":
                self.i += 1
                clauses = []
                cond = self.parse_expr()
                body = self.parse_block(("elseif", "else", "end"))
                clauses.append((cond, body))
                els = None
                while True:
                    if self.is_kw("elseif"):
                        self.i += 1
                        c = self.parse_expr()
                        b = self.parse_block(("elseif", "else", "end"))
                        clauses.append((c, b))
                    elif self.is_kw("else"):
                        self.i += 1
                        els = self.parse_block(("end",))
                    else:
                        break
                self.i += 1  # end
                return ("if", clauses, els, line)
            if kw == "for":
                self.i += 1
                paren = self.is_op("(")
                if paren:
                    self.i += 1
                var = self.t.val
                self.i += 1
                self.eat_op("=")
                e = self.parse_expr()
                if paren:
                    self.eat_op(")")
                body = self.parse_block(("end",))
                self.i += 1
                return ("for", var, e, body, line)
            if kw == "while":
                self.i += 1
                c = self.parse_expr()
                body = self.parse_block(("end",))
                self.i += 1
                return ("while", c, body, line)
            if kw == "switch":
                self.i += 1
                e = self.parse_expr()
                self.skip_nl()
                cases, default = [], None
                while not self.is_kw("end"):
                    if self.is_kw("case"):
                        self.i += 1
                        ce = self.parse_expr()
                        b = self.parse_block(("case", "otherwise", "end"))
                        cases.append((ce, b))
                    elif self.is_kw("otherwise"):
                        self.i += 1
                        default = self.parse_block(("case", "otherwise", "end"))
                    else:
                        self.err("case expected")
                self.i += 1
                return ("switch", e, cases, default, line)
            if kw == "try":
                self.i += 1
                body = self.parse_block(("catch", "end"))
                cvar, cbody = None, []
                if self.is_kw("catch"):
                    self.i += 1
                    if self.t.kind == "ID" and self.peek().kind == "NL" and self.toks[self.i - 1].line == self.t.line:
                        cvar = self.t.val
                        self.i += 1
                    cbody = self.parse_block(("end",))
                self.i += 1
                return ("try", body, cvar, cbody, line)
            if kw in ("break", "continue", "return"):
                self.i += 1
                return (kw, line)
            if kw in ("global", "persistent"):
                self.i += 1
                names = []
                while self.t.kind == "ID":
                    names.append(self.t.val)
                    self.i += 1
                return ("global", names, kw, line)
            self.err(f"unexpected keyword {kw}")
        # command syntax: `format long`, `addpath include`, `close all`
        if t.kind == "ID" and t.val in COMMAND_WORDS:
            nx = self.peek()
            if nx.kind in ("ID", "NL", "EOF") or (nx.kind == "OP" and nx.val in (";", ",")):
                if not (nx.kind == "OP" and nx.val in ("=", "(")):
                    words = []
                    self.i += 1
                    while self.t.kind not in ("NL", "EOF") and not self.is_op(";", ","):
                        words.append(str(self.t.val))
                        self.i += 1
                    return ("command", t.val, words, line)
        # multi-assignment [a, b] = f(...)
        if self.is_op("["):
            save = self.i
            lhs = self.try_parse_multi_lhs()
            if lhs is not None:
                rhs = self.parse_expr()
                return ("assign", lhs, rhs, line)
            self.i = save
        e = self.parse_expr()
        if self.is_op("="):
            self.i += 1
            rhs = self.parse_expr()
            return ("assign", [e], rhs, line)
        return ("expr", e, line)

    def try_parse_multi_lhs(self):
        # find the matching ']' and check for '=' (not '==') right after it
        depth, j = 0, self.i
        while True:
            tk = self.toks[j]
            if tk.kind in ("NL", "EOF"):
                return None
            if tk.kind == "OP" and tk.val in "([{":
                depth += 1
            elif tk.kind == "OP" and tk.val in ")]}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        nx = self.toks[j + 1]
        if not (nx.kind == "OP" and nx.val == "="):
            return None
        self.i += 1  # [
        lhs = []
        self.in_matrix += 1
        while not self.is_op("]"):
            if self.is_op(","):
                self.i += 1
                continue
            if self.is_op("~"):
                lhs.append(("tilde",))
                self.i += 1
                continue
            lhs.append(self.parse_postfix_only())
        self.in_matrix -= 1
        self.i += 1  # ]
        self.eat_op("=")
        return lhs

    def parse_postfix_only(self):
        save = self.in_matrix
        self.in_matrix = 0
        e = self.parse_postfix()
        self.in_matrix = save
        return e

    # ---- expressions (precedence climbing, MATLAB's table) ------------------------------------
    def parse_expr(self):
        return self.parse_oror()

    def _binary_level(self, ops, sub):
        left = sub()
        while self.t.kind == "OP" and self.t.val in ops and not self._op_is_element_separator():
            op = self.t.val
            self.i += 1
            right = sub()
            left = ("binop", op, left, right)
        return left

    def _op_is_element_separator(self):
        """Inside [ ] a '+' or '-' with whitespace in front and none behind starts a new element: [a -b]."""
        if not self.in_matrix or self.in_index_inner():
            return False
        t = self.t
        return t.val in ("+", "-") and t.sp and not t.sp_after

    def in_index_inner(self):
        return self._paren_depth > 0

    _paren_depth = 0

    def parse_oror(self):
        return self._binary_level(("||",), self.parse_andand)

    def parse_andand(self):
        return self._binary_level(("&&",), self.parse_or)

    def parse_or(self):
        return self._binary_level(("|",), self.parse_and)

    def parse_and(self):
        return self._binary_level(("&",), self.parse_cmp)

    def parse_cmp(self):
        return self._binary_level(("==", "~=", "<", "<=", ">", ">="), self.parse_range)

    def parse_range(self):
        first = self.parse_additive()
        if self.is_op(":") and not self._colon_ends_here():
            self.i += 1
            second = self.parse_additive()
            if self.is_op(":") and not self._colon_ends_here():
                self.i += 1
                third = self.parse_additive()
                return ("range", first, second, third)
            return ("range", first, None, second)
        return first

    def _colon_ends_here(self):
        nx = self.peek()
        return nx.kind == "OP" and nx.val in (")", ",", "]", "}") or nx.kind in ("NL", "EOF")

    def parse_additive(self):
        return self._binary_level(("+", "-"), self.parse_mul)

    def parse_mul(self):
        return self._binary_level(("*", "/", "\\", ".*", "./", ".\\"), self.parse_unary)

    def parse_unary(self):
        if self.is_op("+", "-", "~", "!"):
            op = self.t.val
            self.i += 1
            operand = self.parse_unary()
            return ("unop", "~" if op == "!" else op, operand)
        return self.parse_power()

    def parse_power(self):
        base = self.parse_postfix()
        while self.is_op("^", ".^"):
            op = self.t.val
            self.i += 1
            # the exponent binds unary operators tighter than the power: 2^-1
            if self.is_op("+", "-", "~"):
                uop = self.t.val
                self.i += 1
                ex = ("unop", uop, self.parse_postfix())
            else:
                ex = self.parse_postfix()
            base = ("binop", op, base, ex)
        return base

    def parse_postfix(self):
        e = self.parse_primary()
        while True:
            t = self.t
            if t.kind == "OP" and t.val == "(" and not (self.in_matrix and not self._paren_depth and t.sp):
                self.i += 1
                args = self.parse_args(")")
                e = ("index", e, args, "()")
            elif t.kind == "OP" and t.val == "{" and not (self.in_matrix and not self._paren_depth and t.sp):
                self.i += 1
                args = self.parse_args("}")
                e = ("index", e, args, "{}")
            elif t.kind == "OP" and t.val == "." and not t.sp and self.peek().kind in ("ID", "KW") and not self.peek().sp:
                self.i += 1
                e = ("field", e, self.t.val)
                self.i += 1
            elif t.kind == "OP" and t.val == "." and self.peek().kind == "OP" and self.peek().val == "(":
                self.i += 2
                self._paren_depth += 1
                name = self.parse_expr()
                self._paren_depth -= 1
                self.eat_op(")")
                e = ("dynfield", e, name)
            elif t.kind == "OP" and t.val in ("'", ".'"):
                self.i += 1
                e = ("postfix", t.val, e)
            else:
                return e

    def parse_args(self, closer):
        args = []
        self._paren_depth += 1
        self.in_index += 1
        while True:
            while self.t.kind == "NL":
                self.i += 1
            if self.is_op(closer):
                self.i += 1
                break
            if self.is_op(","):
                self.i += 1
                continue
            if self.is_op(":") and (self.peek().kind == "OP" and self.peek().val in (",", closer)):
                args.append(("colon_all",))
                self.i += 1
                continue
            args.append(self.parse_expr())
        self.in_index -= 1
        self._paren_depth -= 1
        return args

    def parse_primary(self):
        t = self.t
        if t.kind == "NUM":
            self.i += 1
            return ("num", t.val)
        if t.kind == "STR":
            self.i += 1
            return ("str", t.val)
        if t.kind == "ID":
            self.i += 1
            return ("id", t.val)
        if t.kind == "KW" and t.val == "end" and self.in_index:
            self.i += 1
            return ("end",)
        if t.kind == "OP":
            if t.val == "(":
                self.i += 1
                self._paren_depth += 1
                save_m = self.in_matrix
                self.in_matrix = 0
                e = self.parse_expr()
                self.in_matrix = save_m
                self._paren_depth -= 1
                self.eat_op(")")
                return ("paren", e)
            if t.val == "[":
                return self.parse_matrix("[", "]", "matrix")
            if t.val == "{":
                return self.parse_matrix("{", "}", "cell")
            if t.val == "@":
                self.i += 1
                if self.is_op("("):
                    self.i += 1
                    params = []
                    while not self.is_op(")"):
                        if self.is_op(","):
                            self.i += 1
                            continue
                        params.append(self.t.val)
                        self.i += 1
                    self.i += 1
                    save_m, save_p = self.in_matrix, self._paren_depth
                    self.in_matrix, self._paren_depth = 0, 0
                    body = self.parse_expr()
                    self.in_matrix, self._paren_depth = save_m, save_p
                    return ("anon", params, body)
                name = self.t.val
                self.i += 1
                return ("fhandle", name)
            if t.val == ":":
                self.i += 1
                return ("colon_all",)
        self.err("expression expected")

    def parse_matrix(self, opener, closer, kind):
        self.i += 1
        save_p, save_i = self._paren_depth, self.in_index
        self._paren_depth = 0          # `end` stays a value inside x([1 end]): in_index is left alone
        self.in_matrix += 1
        rows, row = [], []
        while True:
            t = self.t
            if t.kind == "EOF":
                self.err("unterminated matrix")
            if self.is_op(closer):
                self.i += 1
                break
            if t.kind == "NL" or self.is_op(";"):
                self.i += 1
                if row:
                    rows.append(row)
                    row = []
                continue
            if self.is_op(","):
                self.i += 1
                continue
            row.append(self.parse_expr())
        if row:
            rows.append(row)
        self.in_matrix -= 1
        self._paren_depth, self.in_index = save_p, save_i
        return (kind, rows)


def parse(src: str, fname: str = "<src>"):
    return Parser(src, fname).parse_file()
