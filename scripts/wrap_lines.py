"""Wrap over-long C++ / HIP source lines at argument / operator boundaries (development aid; run, then rebuild and re-run the tests).
Only lines longer than LIMIT are touched; preprocessor lines, macro continuation lines and lines whose break points all sit inside string
literals are left alone.  A trailing // comment moves onto its own line above the statement first.
    python scripts/wrap_lines.py file ..."""
import re, sys
LIMIT = 136


def split_trailing_comment(line):
    """(code, comment) where comment starts at a // that is outside string / char literals; comment == '' if none"""
    in_s = in_c = False
    i = 0
    while i < len(line) - 1:
        ch = line[i]
        if in_s:
            if ch == "\\": i += 1
            elif ch == '"': in_s = False
        elif in_c:
            if ch == "\\": i += 1
            elif ch == "'": in_c = False
        else:
            if ch == '"': in_s = True
            elif ch == "'" and not (i > 0 and (line[i - 1].isalnum())): in_c = True
            elif ch == "/" and line[i + 1] == "/":
                return line[:i].rstrip(), line[i:]
        i += 1
    return line, ""


def break_points(code):
    """indices AFTER which the line may be broken, outside literals: after '; ' between the statements of a one-line { body } (preferred:
    returned with priority 0), after ', ' inside a parenthesis whose span is long (>= 40 characters: an argument list, not x(j, i)), before
    ' && ' / ' || ' inside parentheses, before ' << ' at top level (priority 1)."""
    pts = []
    in_s = in_c = False
    stack = []   # (char, index)
    spans = {}
    # first pass: span of every bracket
    i = 0
    while i < len(code):
        ch = code[i]
        if in_s:
            if ch == "\\": i += 1
            elif ch == '"': in_s = False
        elif in_c:
            if ch == "\\": i += 1
            elif ch == "'": in_c = False
        else:
            if ch == '"': in_s = True
            elif ch == "'" and not (i > 0 and code[i - 1].isalnum()): in_c = True
            elif ch in "([{": stack.append((ch, i))
            elif ch in ")]}" and stack:
                o, j = stack.pop(); spans[j] = i
        i += 1
    for o, j in stack:
        spans[j] = len(code)
    in_s = in_c = False
    stack = []
    i = 0
    while i < len(code):
        ch = code[i]
        if in_s:
            if ch == "\\": i += 1
            elif ch == '"': in_s = False
        elif in_c:
            if ch == "\\": i += 1
            elif ch == "'": in_c = False
        else:
            if ch == '"': in_s = True
            elif ch == "'" and not (i > 0 and code[i - 1].isalnum()): in_c = True
            elif ch in "([{": stack.append((ch, i))
            elif ch in ")]}" and stack: stack.pop()
            elif stack:
                o, j = stack[-1]
                if o == "{" and ch == ";" and i + 1 < len(code) and code[i + 1] == " ":
                    pts.append((0, i + 2))
                elif o == "(" and ch == "," and i + 1 < len(code) and code[i + 1] == " " and spans.get(j, len(code)) - j >= 40:
                    pts.append((1, i + 2))
                elif o == "(" and code.startswith((" && ", " || "), i):
                    pts.append((1, i + 1))
            elif code.startswith(" << ", i):
                pts.append((1, i + 1))
        i += 1
    return pts


def wrap(line):
    if len(line) <= LIMIT or line.lstrip().startswith("#") or line.rstrip().endswith("\\"):
        return [line]
    indent = len(line) - len(line.lstrip())
    code, comment = split_trailing_comment(line)
    out = []
    if comment and code.strip():
        out.append(" " * indent + comment)
        line = code
        if len(line) <= LIMIT:
            return out + [line]
    elif comment and not code.strip():   # a pure comment line: split at words
        words = comment[2:].split()
        cur = " " * indent + "//"
        for w in words:
            if len(cur) + 1 + len(w) > LIMIT:
                out.append(cur); cur = " " * indent + "// " + w
            else:
                cur += " " + w
        return out + [cur]
    cont = " " * (indent + 4)
    rest = line
    first = True
    while len(rest) > LIMIT:
        cand = [(pr, p) for pr, p in break_points(rest) if p <= LIMIT and p > (indent + 20)]
        if not cand:
            break
        stmt = [p for pr, p in cand if pr == 0 and p > LIMIT // 2]   # a statement boundary in the right half of the line wins
        p = stmt[-1] if stmt else max(p for pr, p in cand)
        out.append(rest[:p].rstrip())
        rest = cont + rest[p:].lstrip()
        first = False
    out.append(rest)
    return out


for path in sys.argv[1:]:
    src = open(path).read().split("\n")
    dst = []
    in_macro = False
    for ln in src:
        if in_macro or ln.lstrip().startswith("#define") and ln.rstrip().endswith("\\"):
            dst.append(ln)
            in_macro = ln.rstrip().endswith("\\")
            continue
        dst.extend(wrap(ln))
    if dst != src:
        open(path, "w").write("\n".join(dst))
        print(path, sum(1 for l in src if len(l) > 140), "->", sum(1 for l in dst if len(l) > 140))
