"""The C-ABI shared library builds, loads, and exports every symbol include/setok_hip.h declares;
the ctypes table mirrors the header.  No compute calls (no GPU needed)."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, ROOT)
    from setok_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "setok_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = re.findall(r"\b(?:int64_t|int|void|const char\*)\s+(setok_\w+)\s*\(([^;]*)\)\s*;", text, flags=re.S)
    return {name: [a for a in args.split(",") if a.strip() and a.strip() != "void"] for name, args in decls}


def test_header_symbols_exported(lib):
    decl = _declared()
    assert len(decl) >= 13
    cdll = ctypes.CDLL(lib.LIB_PATH)
    for name in decl:
        assert hasattr(cdll, name), f"{name} declared in include/setok_hip.h but not exported"


def test_ctypes_table_matches_header(lib):
    decl = _declared()
    assert set(decl) == set(lib.SIGNATURES), set(decl) ^ set(lib.SIGNATURES)
    for name, args in decl.items():
        assert len(args) == len(lib.SIGNATURES[name]), f"{name}: header has {len(args)} args, ctypes {len(lib.SIGNATURES[name])}"


def test_abi_version_and_error_plumbing(lib):
    l = lib.load()
    assert l.setok_abi_version() == 8
    # argument validation happens on the host before any launch: usable without a GPU
    rc = l.setok_linear(None, 0, 0, None, 0, None, None, None, None, 0, 1, 1, 16, 0, 1, 0, 0, 0)
    assert rc == -1 and b"null operand" in l.setok_last_error()
    with pytest.raises(lib.SetokHipError):
        lib.call("setok_layernorm", None, 0, 1, 1, 1, 1, 4, 12, 1e-5)   # C not a multiple of 8


def test_no_fallback_when_library_missing(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libsetok_hip.so")
    with pytest.raises(lib.SetokHipError):
        lib.load()


def test_driver_build_entry_point():
    """`__graft_entry__.build()` — what the driver runs as its build check: make + import + ABI version against the header."""
    import importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    importlib.import_module("__graft_entry__").build()
