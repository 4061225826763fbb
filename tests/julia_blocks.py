"""A block-structure check for the Julia sources (Julia is not in the build image, so the files never run here): strings, docstrings,
comments and character literals are blanked, then brackets and block keywords are matched with a stack — `end` inside `[...]` is an index,
`for` / `if` / `while` directly inside brackets are generators, `:end` / `x.end` are not closers.  Catches what a missing or stray `end`,
an unclosed string or an unbalanced bracket would do to the file; it does not replace a parser."""
import re

OPENERS = {"function", "if", "for", "while", "let", "begin", "struct", "module", "baremodule", "do", "try", "quote", "macro"}
GENERATORS = {"for", "if", "while"}
_CHAR = re.compile(r"'(\\.|[^\\'])'")
_TOKEN = re.compile(r"\n|[A-Za-z_¡-￿][A-Za-z_0-9!¡-￿]*|[()\[\]{}]|\S")


def blank(src):
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith('"""', i):
            j = src.index('"""', i + 3)
            out.append("\n" * src[i:j + 3].count("\n"))
            i = j + 3
        elif c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""' + "\n" * src[i:j].count("\n"))
            i = j + 1
        elif src.startswith("#=", i):
            j = src.index("=#", i + 2)
            out.append("\n" * src[i:j].count("\n"))
            i = j + 2
        elif c == "#":
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif c == "'" and _CHAR.match(src, i):
            out.append("' '")
            i = _CHAR.match(src, i).end()
        else:
            out.append(c)
            i += 1
    return "".join(out)


def check(path):
    """list of problems (empty: balanced)"""
    src = blank(open(path, encoding="utf-8").read())
    stack, line, errs, prev = [], 1, [], ""
    for m in _TOKEN.finditer(src):
        t = m.group(0)
        if t == "\n":
            line += 1
            continue
        if t in "([{":
            stack.append((t, line))
        elif t in ")]}":
            want = {")": "(", "]": "[", "}": "{"}[t]
            if not stack or stack[-1][0] != want:
                errs.append("line %d: unmatched %s (innermost open: %s)" % (line, t, stack[-1] if stack else None))
            else:
                stack.pop()
        elif t == "end" and prev not in (":", "."):
            if stack and stack[-1][0] == "[":
                pass                                               # a[end]
            elif not stack or stack[-1][0] in "({":
                errs.append("line %d: `end` closes nothing (innermost open: %s)" % (line, stack[-1] if stack else None))
            else:
                stack.pop()
        elif t in OPENERS and prev not in (":", "."):
            if not (t in GENERATORS and stack and stack[-1][0] in "([{"):
                stack.append((t, line))
        elif t == "type" and prev in ("abstract", "primitive"):
            stack.append((t, line))
        prev = t
    errs += ["unclosed %s opened at line %d" % s for s in stack]
    return errs


def _standalone_strings(src):
    """(first line, last line) of every string literal that is a statement of its own: only blanks in front of it on its first line and
    only blanks or a comment behind it on its last — i.e. what Julia's parser takes for a docstring of the NEXT expression."""
    out, i, n, line = [], 0, len(src), 1
    while i < n:
        c = src[i]
        if c == "\n":
            line += 1
            i += 1
        elif src.startswith("#=", i):
            j = src.index("=#", i + 2) + 2
            line += src[i:j].count("\n")
            i = j
        elif c == "#":
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif c == "'" and _CHAR.match(src, i):
            i = _CHAR.match(src, i).end()
        elif c == '"':
            if src.startswith('"""', i):
                j = src.index('"""', i + 3) + 3
            else:
                j = i + 1
                while src[j] != '"':
                    j += 2 if src[j] == "\\" else 1
                j += 1
            bol = src.rfind("\n", 0, i) + 1
            eol = src.find("\n", j)
            eol = n if eol < 0 else eol
            behind = src[j:eol].strip()
            l0, l1 = line, line + src[i:j].count("\n")
            if src[bol:i].strip() == "" and (behind == "" or behind.startswith("#")):
                out.append((l0, l1))
            line = l1
            i = j
        else:
            i += 1
    return out


def doc_problems(path):
    """Docstrings that document nothing.  A string literal on lines of its own is attached by Julia to the expression that follows; when
    that is another such literal the file does not even LOAD (`Core.@doc "a" "b"`: "cannot document the following expression"), and when
    it is an `end` or the end of the file the literal is dead.  (What the block matcher cannot see, because it blanks strings.)"""
    raw = open(path, encoding="utf-8").read()
    lines = raw.split("\n")
    code = blank(raw).split("\n")
    lits = _standalone_strings(raw)
    starts = {a for a, _ in lits}
    errs = []
    for a, b in lits:
        k = b                                     # 0-based index of the line after the literal
        while k < len(lines) and (lines[k].strip() == "" or (code[k].strip() == "" and (k + 1) not in starts)):
            k += 1
        if k >= len(lines):
            errs.append("line %d: string literal in front of the end of the file" % a)
        elif (k + 1) in starts:
            errs.append("line %d: string literal directly followed by another one at line %d (Base.Docs cannot document a string)" % (a, k + 1))
        elif re.match(r"\s*end\b", code[k]):
            errs.append("line %d: string literal in front of `end`" % a)
    return errs
