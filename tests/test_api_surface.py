"""Drop-in check of the Python boundary: every module of the reference package imports under the same dotted
name, and every public function, class, method and module-level name it defines (or a package re-exports) exists
here with the same argument names, order and defaults.  The list was read from the reference's files with `ast`
(tests/golden/make_api_surface.py -> tests/golden/api_surface.json: names and defaults only)."""
import ast
import importlib
import inspect
import json
from pathlib import Path

import pytest

SURFACE = json.loads((Path(__file__).resolve().parent / "golden" / "api_surface.json").read_text())


def _same_default(want_src, got):
    if want_src is None:
        return got is inspect.Parameter.empty
    if got is inspect.Parameter.empty:
        return False
    try:
        want = ast.literal_eval(want_src)
    except (ValueError, SyntaxError):
        return True            # an expression (DEFAULT_RNG, 1 / 4, a constant's name): presence is what is checked
    try:
        return bool(want == got) or (isinstance(want, float) and abs(want - got) < 1e-15)
    except Exception:          # noqa: BLE001 — array-valued defaults
        return True


def _check_args(where, want, fn, skip_self=False):
    params = list(inspect.signature(fn).parameters.values())
    if skip_self and params and params[0].name == "self":
        params = params[1:]
        want = [w for w in want if w[0] != "self"]
    names = [("*" if p.kind is p.VAR_POSITIONAL else "**" if p.kind is p.VAR_KEYWORD else "") + p.name for p in params]
    assert names[:len(want)] == [w[0] for w in want], f"{where}: arguments {names} != reference {[w[0] for w in want]}"
    for p in params[len(want):]:       # extra trailing arguments must be optional
        assert p.default is not inspect.Parameter.empty or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD), f"{where}: extra required argument {p.name}"
    for (name, dflt), p in zip(want, params):
        assert _same_default(dflt, p.default), f"{where}: default of {name}: {p.default!r} != reference {dflt}"


@pytest.mark.parametrize("modname", sorted(SURFACE))
def test_module_surface_matches_reference(modname):
    entry = SURFACE[modname]
    mod = importlib.import_module(modname)
    for name in entry["constants"] + entry["imports"]:
        assert hasattr(mod, name), f"{modname}.{name} missing"
    for name, want in entry["functions"].items():
        fn = getattr(mod, name, None)
        assert fn is not None, f"{modname}.{name} missing"
        target = getattr(fn, "func", fn)           # (a matplotlib FuncFormatter wraps its function)
        assert callable(target)
        _check_args(f"{modname}.{name}", want, target)
    for cname, methods in entry["classes"].items():
        cls = getattr(mod, cname, None)
        assert inspect.isclass(cls), f"{modname}.{cname} missing"
        for mname, want in methods.items():
            assert hasattr(cls, mname), f"{modname}.{cname}.{mname} missing"
            member = inspect.getattr_static(cls, mname)
            if isinstance(member, property):
                continue
            fn = member.__func__ if isinstance(member, (staticmethod, classmethod)) else member
            if not inspect.isfunction(fn):
                continue                             # generated (dataclass / NamedTuple) members
            _check_args(f"{modname}.{cname}.{mname}", want, fn, skip_self=True)
        # the names an instance of the reference's class carries (self.<name> of __init__, dataclass fields): here as an
        # attribute assigned by one of the class's own methods, a property, or a class-level name
        want_attrs = entry.get("attributes", {}).get(cname, [])
        if want_attrs:
            have = set(dir(cls)) | set(getattr(cls, "__annotations__", {}))
            try:
                tree = ast.parse(inspect.getsource(cls).lstrip() if not inspect.getsource(cls).startswith("class") else inspect.getsource(cls))
            except (OSError, TypeError, SyntaxError, IndentationError):
                tree = None
            if tree is not None:
                for st in ast.walk(tree):
                    if isinstance(st, ast.Attribute) and isinstance(st.value, ast.Name) and st.value.id == "self" and isinstance(st.ctx, ast.Store):
                        have.add(st.attr)
            missing = [a for a in want_attrs if a not in have]
            assert not missing, f"{modname}.{cname}: instance attributes {missing} of the reference are missing"
