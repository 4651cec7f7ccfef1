"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/bde_hip.h declares, with the arity the ctypes table binds (no compute
calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "bde_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(bde_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_header_symbols_exported_and_bound():
    from beyond_deep_ensembles_amd import _lib
    if not _lib.is_built():
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    decl = declared()
    assert len(decl) >= 20
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decl.items():
        assert hasattr(raw, name), f"{name} not exported"
        assert len(_lib.SIGNATURES[name][1]) == nargs, name
    assert lib.bde_version() >= 100 and lib.bde_arch() == b"gfx950"
    assert lib.bde_svgd_ws_bytes(8) > 0 and lib.bde_svgd_ws_bytes(65) == 0      # M <= 64
    assert lib.bde_svgd_kstat_floats(8) == 4 * 64 + 8 + 4


def test_no_product_import_of_the_oracle():
    """Nothing under beyond_deep_ensembles_amd/ may import oracle/ (DESIGN.md section 1)."""
    pkg = os.path.join(ROOT, "beyond_deep_ensembles_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("test seam", ""), os.path.join(dirpath, f)
