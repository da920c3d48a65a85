"""Remove preprocessor switches from the kernel sources, keeping the branch a build WITHOUT the macro compiles:
    python tools/strip_switches.py MACRO[,MACRO...] FILE...
Handles `#ifdef M`, `#ifndef M`, `#if defined(M)` / `#if !defined(M) && ...` (every listed macro is taken as undefined; a condition
that still names another macro is kept with the listed ones folded out), with their `#else` / `#elif` / `#endif`.  Used once per round to
delete experiment variants whose verdict is recorded in DESIGN.md; the product binary must come out unchanged (compare the .so)."""
import re
import sys


def fold(cond: str, macros) -> str:
    """cond of an #if with defined(M) -> 0 for the listed macros; returns '0', '1' or the simplified text."""
    c = cond
    for m in macros:
        c = re.sub(r"!\s*defined\s*\(\s*%s\s*\)" % m, "1", c)
        c = re.sub(r"defined\s*\(\s*%s\s*\)" % m, "0", c)
    terms = [t.strip() for t in c.split("&&")]
    if "||" in c or "(" in c.replace("defined(", "").replace("defined (", ""):
        return c if not re.fullmatch(r"[01\s&|!()]+", c) else str(int(bool(eval(c.replace("&&", " and ").replace("||", " or ").replace("!", " not ")))))
    if any(t == "0" for t in terms):
        return "0"
    terms = [t for t in terms if t != "1"]
    return " && ".join(terms) if terms else "1"


def strip(text: str, macros):
    out = []
    # stack entries: (kind, emitting_before, taken_branch_active, passthrough)
    stack = []
    emitting = True
    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting:
                out.append(line)
            continue
        kw, rest = m.group(1), m.group(2).strip()
        rest_nc = re.sub(r"//.*", "", rest).strip()
        if kw in ("ifdef", "ifndef", "if"):
            if kw == "ifdef":
                val = "0" if rest_nc in macros else None
            elif kw == "ifndef":
                val = "1" if rest_nc in macros else None
            else:
                val = fold(rest_nc, macros) if any(mm in rest_nc for mm in macros) else None
            if val in ("0", "1"):
                stack.append(("fold", emitting, val == "1"))
                emitting = emitting and val == "1"
            else:
                stack.append(("keep", emitting, True))
                if emitting:
                    out.append(line if val is None else re.sub(r"(#\s*if\b).*", r"\1 " + val, line))
        elif kw in ("else", "elif"):
            kind, before, taken = stack[-1]
            if kind == "fold":
                if kw == "elif":
                    raise SystemExit("elif after a folded #if: edit by hand: " + line)
                emitting = before and not taken
                stack[-1] = (kind, before, True)
            elif emitting or before:
                if before:
                    out.append(line)
        else:  # endif
            kind, before, _ = stack.pop()
            emitting = before
            if kind == "keep" and before:
                out.append(line)
    assert not stack
    return "\n".join(out)


if __name__ == "__main__":
    macros = sys.argv[1].split(",")
    for f in sys.argv[2:]:
        src = open(f).read()
        new = strip(src, macros)
        if new != src:
            open(f, "w").write(new)
            print("stripped", f)
