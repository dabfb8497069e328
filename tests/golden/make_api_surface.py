"""tests/golden/api_surface.json: the public names of the reference package and the argument lists of its
functions / methods, read from the reference's files with `ast` (nothing is imported or executed; the file
holds names and defaults — an interface description, no code).  "constants" = every public module-level name
that is assigned (numbers, tables, the named trellis instances).  "attributes" (round 6) = per class, the public names its
instances carry: every `self.<name> = ...` of `__init__` and every annotated field of the class body (dataclass fields).

    python3 tests/golden/make_api_surface.py
"""
import ast
import json
from pathlib import Path

REF = Path("/root/reference/waveforms")
OUT = Path(__file__).resolve().parent / "api_surface.json"


def args_of(fn: ast.FunctionDef):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    ndef = len(a.defaults)
    out = []
    for i, name in enumerate(pos):
        d = a.defaults[i - (len(pos) - ndef)] if i >= len(pos) - ndef else None
        out.append([name, ast.unparse(d) if d is not None else None])
    if a.vararg:
        out.append(["*" + a.vararg.arg, None])
    for k, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append([k.arg, ast.unparse(d) if d is not None else None])
    if a.kwarg:
        out.append(["**" + a.kwarg.arg, None])
    return out


surface = {}
for path in sorted(REF.rglob("*.py")):
    mod = ".".join(("waveforms",) + path.relative_to(REF).with_suffix("").parts)
    if mod.endswith(".__init__"):
        mod = mod[: -len(".__init__")]
    tree = ast.parse(path.read_text())
    entry = {"functions": {}, "classes": {}, "constants": [], "imports": []}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and not node.name.startswith("_"):
            entry["functions"][node.name] = args_of(node)
        elif isinstance(node, ast.ClassDef) and not node.name.startswith("_"):
            methods = {}
            attrs = []
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and (not sub.name.startswith("_") or sub.name in ("__init__", "__call__")):
                    methods[sub.name] = args_of(sub)
                if isinstance(sub, ast.FunctionDef) and sub.name == "__init__":
                    for st in ast.walk(sub):
                        if isinstance(st, (ast.Assign, ast.AnnAssign, ast.AugAssign)):
                            for t in (st.targets if isinstance(st, ast.Assign) else [st.target]):
                                if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == "self" and not t.attr.startswith("_"):
                                    attrs.append(t.attr)
                if isinstance(sub, ast.AnnAssign) and isinstance(sub.target, ast.Name) and not sub.target.id.startswith("_"):
                    attrs.append(sub.target.id)
            entry["classes"][node.name] = methods
            entry.setdefault("attributes", {})[node.name] = sorted(set(attrs))
        elif isinstance(node, (ast.Assign, ast.AnnAssign)):
            targets = node.targets if isinstance(node, ast.Assign) else [node.target]
            for t in targets:
                if isinstance(t, ast.Name) and not t.id.startswith("_") and t.id not in ("rng",):
                    entry["constants"].append(t.id)
        elif isinstance(node, ast.ImportFrom) and node.level >= 1:      # names a package re-exports
            entry["imports"] += [a.asname or a.name for a in node.names]
    surface[mod] = entry
OUT.write_text(json.dumps(surface, indent=1, sort_keys=True) + "\n")
print("wrote", OUT, len(surface), "modules")
