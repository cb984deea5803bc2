"""Fall-through for the drop-in shims: a name the MI355X implementation does not provide is looked up in the module of the same
name that comes NEXT on sys.path -- the reference's own (e.g. evaluate.evaluate_image, metrics.qwk: scalar host-side metrics
outside the hot path).  The shadowed module is loaded under a private alias, once."""
import importlib.util
import os
import sys

_cache = {}


def next_module(name, shim_file):
    if name in _cache:
        return _cache[name]
    here = os.path.dirname(os.path.abspath(shim_file))
    if os.path.basename(shim_file) == "__init__.py":
        here = os.path.dirname(here)                       # a package shim: its parent is the path entry
    for entry in sys.path:
        entry = os.path.abspath(entry or ".")
        if entry == here:
            continue
        for cand, is_pkg in ((os.path.join(entry, name, "__init__.py"), True), (os.path.join(entry, name + ".py"), False)):
            if os.path.isfile(cand):
                alias = "_shadowed_" + name
                spec = importlib.util.spec_from_file_location(alias, cand, submodule_search_locations=[os.path.dirname(cand)] if is_pkg else None)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[alias] = mod
                spec.loader.exec_module(mod)
                _cache[name] = mod
                return mod
    raise ImportError(f"drop-in {name!r}: no shadowed module of that name further down sys.path")


def fallthrough(name, shim_file):
    def __getattr__(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        try:
            return getattr(next_module(name, shim_file), attr)
        except ImportError as e:
            raise AttributeError(f"module {name!r} (MI355X drop-in) has no attribute {attr!r} and {e}") from None
    return __getattr__
