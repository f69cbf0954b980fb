"""Tokeniser for the MATLAB subset the reference's hot-path files use (oracle/mlab/__init__.py)."""
from __future__ import annotations

from dataclasses import dataclass

KEYWORDS = {"function", "end", "if", "elseif", "else", "for", "while", "switch", "case", "otherwise", "try", "catch",
            "break", "continue", "return", "global", "persistent"}

# longest first
OPERATORS = ["...", ".^'", ".*", "./", ".\\", ".^", ".'", "==", "~=", "<=", ">=", "&&", "||",
             "+", "-", "*", "/", "\\", "^", "<", ">", "=", "&", "|", "~", ":", ",", ";", "(", ")", "[", "]", "{", "}",
             "@", ".", "'", "!"]


@dataclass
class Tok:
    kind: str       # NUM STR ID KW OP NL EOF
    val: object
    line: int
    sp: bool        # whitespace in front of the token
    sp_after: bool = False


class LexError(Exception):
    pass


def _strip_block_comments(src: str) -> str:
    """%{ ... %} on lines of their own (nesting allowed); the lines are blanked so that line numbers stay."""
    out, lvl = [], 0
    for ln in src.split("\n"):
        t = ln.strip()
        if t == "%{":
            lvl += 1
            out.append("")
        elif t == "%}" and lvl:
            lvl -= 1
            out.append("")
        else:
            out.append("" if lvl else ln)
    return "\n".join(out)


def tokenize(src: str, fname: str = "<src>"):
    src = _strip_block_comments(src.replace("\r\n", "\n").replace("\r", "\n"))
    toks: list[Tok] = []
    i, n, line = 0, len(src), 1
    depth = []          # open brackets: '(' '[' '{'
    sp = False

    def prev_allows_transpose():
        if not toks:
            return False
        t = toks[-1]
        if t.kind in ("NUM", "ID"):
            return True
        if t.kind == "KW":
            return t.val == "end" and bool(depth)
        if t.kind == "OP":
            return t.val in (")", "]", "}", "'", ".'")
        return False

    while i < n:
        c = src[i]
        if c == "\n":
            toks.append(Tok("NL", "\n", line, sp))
            line += 1
            i += 1
            sp = False
            continue
        if c in " \t\r":
            i += 1
            sp = True
            continue
        if c == "%":
            le = src.find("\n", i)
            i = n if le < 0 else le
            continue
        if src.startswith("...", i):
            # continuation: the rest of the line is a comment
            le = src.find("\n", i)
            i = n if le < 0 else le + 1
            line += 1
            sp = True
            continue
        if c.isdigit() or (c == "." and i + 1 < n and src[i + 1].isdigit()):
            j = i
            while j < n and src[j].isdigit():
                j += 1
            if j < n and src[j] == "." and not (j + 1 < n and src[j + 1] in "*/\\^'"):
                j += 1
                while j < n and src[j].isdigit():
                    j += 1
            if j < n and src[j] in "eEdD" and (j + 1 < n and (src[j + 1].isdigit() or (src[j + 1] in "+-" and j + 2 < n and src[j + 2].isdigit()))):
                j += 2
                while j < n and src[j].isdigit():
                    j += 1
            txt = src[i:j].replace("d", "e").replace("D", "e")
            imag = False
            if j < n and src[j] in "ij" and not (j + 1 < n and (src[j + 1].isalnum() or src[j + 1] == "_")):
                imag = True
                j += 1
            v = float(txt)
            toks.append(Tok("NUM", complex(0, v) if imag else v, line, sp))
            i = j
            sp = False
            continue
        if c.isalpha() or c == "_":
            j = i
            while j < n and (src[j].isalnum() or src[j] == "_"):
                j += 1
            w = src[i:j]
            # 'end' inside an index expression is a value, not a block terminator; the parser decides with `depth`
            toks.append(Tok("KW" if w in KEYWORDS else "ID", w, line, sp))
            i = j
            sp = False
            continue
        if c == '"':
            j = i + 1
            buf = []
            while True:
                if j >= n or src[j] == "\n":
                    raise LexError(f"{fname}:{line}: unterminated string")
                if src[j] == '"':
                    if j + 1 < n and src[j + 1] == '"':
                        buf.append('"')
                        j += 2
                        continue
                    break
                buf.append(src[j])
                j += 1
            toks.append(Tok("STR", "".join(buf), line, sp))
            i = j + 1
            sp = False
            continue
        if c == "'":
            in_matrix = bool(depth) and depth[-1] in "[{"
            if prev_allows_transpose() and not (in_matrix and sp):
                toks.append(Tok("OP", "'", line, sp))
                i += 1
                sp = False
                continue
            j = i + 1
            buf = []
            while True:
                if j >= n or src[j] == "\n":
                    raise LexError(f"{fname}:{line}: unterminated string")
                if src[j] == "'":
                    if j + 1 < n and src[j + 1] == "'":
                        buf.append("'")
                        j += 2
                        continue
                    break
                buf.append(src[j])
                j += 1
            toks.append(Tok("STR", "".join(buf), line, sp))
            i = j + 1
            sp = False
            continue
        for op in OPERATORS:
            if src.startswith(op, i):
                if op in "([{":
                    depth.append(op)
                elif op in ")]}":
                    if depth:
                        depth.pop()
                if op == ";" or op == ",":
                    pass
                toks.append(Tok("OP", op, line, sp))
                i += len(op)
                sp = False
                break
        else:
            raise LexError(f"{fname}:{line}: unexpected character {c!r}")
    toks.append(Tok("NL", "\n", line, sp))
    toks.append(Tok("EOF", None, line, False))
    for k in range(len(toks) - 1):
        toks[k].sp_after = toks[k + 1].sp or toks[k + 1].kind in ("NL", "EOF")
    return toks
