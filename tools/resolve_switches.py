"""Resolve decided compile-time switches in a source file (a small unifdef).

    python tools/resolve_switches.py FILE --undef NAME ... --define NAME=VALUE ... [--in-place]

Handles `#ifdef N`, `#ifndef N`, `#if defined(N)`, `#elif defined(N)`, `#else`, `#endif` for the names given, and
`#if <expr>` / `#elif <expr>` whose every identifier is a name given with --define (evaluated as C integers).
Conditionals on any other name are left alone.  A default-definition block `#ifndef N / #define N v / #endif` of a
--define'd name is dropped (the name's uses are NOT substituted: keep a `#define` or `constexpr` for it).
Used once per round to delete ablation / A-B switches whose verdict is recorded (DESIGN / HISTORY); same-box A/B
re-runs go through tools/ab_kernel.sh "SRC=<patched copy>" variants instead.
"""
import argparse
import re
import sys


def evaluate(expr: str, defs: dict, undef: set):
    """True / False when the expression is decided by the given names, None otherwise."""
    e = expr.split("//")[0].strip()
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if m.group(1) in defs else ("0" if m.group(1) in undef else f"defined_{m.group(1)}"), e)
    names = set(re.findall(r"[A-Za-z_]\w*", e))
    if any(n.startswith("defined_") or (n not in defs and n not in undef) for n in names):
        return None
    for n in names:
        e = re.sub(rf"\b{n}\b", str(defs.get(n, 0)), e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", " !=")
    return bool(eval(e, {"__builtins__": {}}))


def resolve(text: str, defs: dict, undef: set) -> str:
    out = []
    # stack entries: [kind, emitting_before, taken, live] — kind "ours" (resolved) or "other" (kept)
    stack = []
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        emitting = all(f[3] for f in stack)
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting:
                out.append(ln)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2).strip()
        if kw in ("ifdef", "ifndef", "if"):
            if kw == "if":
                val = evaluate(rest, defs, undef)
            else:
                name = rest.split()[0]
                val = None if (name not in defs and name not in undef) else ((name in defs) == (kw == "ifdef"))
            # default-definition block of a defined name: drop it whole
            if kw == "ifndef" and val is False and i + 2 < len(lines) and re.match(r"#\s*define\s+" + re.escape(rest.split()[0]) + r"\b", lines[i + 1].strip()) \
                    and re.match(r"#\s*endif", lines[i + 2].strip()):
                i += 3
                continue
            if val is None:
                stack.append(["other", emitting, False, True])
                if emitting:
                    out.append(ln)
            else:
                stack.append(["ours", emitting, val, val])
        elif kw in ("elif", "else"):
            f = stack[-1]
            if f[0] == "other":
                if all(g[3] for g in stack[:-1]):
                    out.append(ln)
            else:
                if f[2]:
                    f[3] = False
                elif kw == "else":
                    f[2] = f[3] = True
                else:
                    val = evaluate(rest, defs, undef)
                    if val is None:
                        raise SystemExit(f"line {i + 1}: #elif mixes resolved and unresolved names: {s}")
                    f[2] = f[3] = val
        else:
            f = stack.pop()
            if f[0] == "other" and all(g[3] for g in stack):
                out.append(ln)
        i += 1
    if stack:
        raise SystemExit("unbalanced conditionals")
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--undef", nargs="*", default=[])
    ap.add_argument("--define", nargs="*", default=[])
    ap.add_argument("--in-place", action="store_true")
    a = ap.parse_args()
    defs = {}
    for d in a.define:
        k, _, v = d.partition("=")
        defs[k] = int(v or 1)
    text = open(a.file).read()
    res = resolve(text, defs, set(a.undef))
    if a.in_place:
        open(a.file, "w").write(res)
    else:
        sys.stdout.write(res)


if __name__ == "__main__":
    main()
