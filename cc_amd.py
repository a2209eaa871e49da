"""Loader for the `contour-context_amd/` package (hyphenated directory -> loaded by path)."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    name = "contour_context_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(_ROOT, "contour-context_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
