"""Loader of the native host helper (``lib/_bde_host*.so``, built from ``csrc/host.cpp`` by
``__graft_entry__.build()``): the per-tensor re-pointing loops of the optimizer shells in C++.
Pure plumbing -- no arithmetic lives here; without the helper the same loops run in Python."""
from __future__ import annotations

import glob
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_mod = None
_tried = False


def load():
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    if os.environ.get("BDE_NO_HOST_HELPER"):
        return None
    for path in sorted(glob.glob(os.path.join(_HERE, "lib", "_bde_host*.so"))):
        try:
            import torch  # noqa: F401  (libtorch must be loaded first)
            spec = importlib.util.spec_from_file_location("_bde_host", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _mod = mod
            break
        except Exception:          # an unloadable helper only costs speed
            _mod = None
    return _mod
