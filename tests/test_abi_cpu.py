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
    if not os.path.isfile(_lib.LIB_PATH) or not os.path.isfile(_lib.LIB_PATH_F16):
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
    for path in (lib.LIB_PATH, lib.LIB_PATH_F16):                 # the bf16 build and the fp16 build of the same sources export the same ABI
        cdll = ctypes.CDLL(path)
        for name in decl:
            assert hasattr(cdll, name), f"{name} declared in include/setok_hip.h but not exported by {os.path.basename(path)}"


def test_ctypes_table_matches_header(lib):
    decl = _declared()
    assert set(decl) == set(lib.SIGNATURES), set(decl) ^ set(lib.SIGNATURES)
    for name, args in decl.items():
        assert len(args) == len(lib.SIGNATURES[name]), f"{name}: header has {len(args)} args, ctypes {len(lib.SIGNATURES[name])}"


def test_abi_version_and_error_plumbing(lib):
    l = lib.load()
    assert l.setok_abi_version() == 9 and lib.load(half=True).setok_abi_version() == 9
    # argument validation happens on the host before any launch: usable without a GPU
    rc = l.setok_linear(None, 0, 0, None, 0, None, None, None, None, 0, 1, 1, 16, 0, 1, 0, 0, 0)
    assert rc == -1 and b"null operand" in l.setok_last_error()
    with pytest.raises(lib.SetokHipError):
        lib.call("setok_layernorm", None, 0, 1, 1, 1, 1, 4, 12, 1e-5)   # C not a multiple of 8


def test_each_build_refuses_the_other_builds_16_bit_type(lib):
    """libsetok_hip.so serves float32 + bfloat16, libsetok_hip_f16.so float32 + float16 (include/setok_hip.h, `dtype`): a buffer of the other
    16-bit type must be refused on the host, never read as something else.  Routing: a call carrying the F16 code goes to the fp16 build."""
    import ctypes as C
    cfg = lib.SetokConfig(image_size=112, patch_size=14, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                          layer_norm_eps=1e-5, select_layer=-2, select_cls_patch=0, token_feat_dim=64, nheads=2, dim_feedforward=128,
                          inner_cluster_layers=1, intra_cluster_layers=1, min_cluster_num=8, threshold=0.5, dtype=2, fold_layernorm=1)
    h = C.c_void_p()
    assert lib.load().setok_create(C.byref(cfg), C.byref(h)) == -1 and b"dtype" in lib.load().setok_last_error()
    cfg.dtype = 1
    assert lib.load(half=True).setok_create(C.byref(cfg), C.byref(h)) == -1 and b"dtype" in lib.load(half=True).setok_last_error()
    assert lib.is_half(0, lib.F16, None) and not lib.is_half(0, 1, 2) and int(lib.F16) == 2
    for half, bad in ((False, 2), (True, 1)):
        l = lib.load(half)
        assert l.setok_linear(None, bad, bad, 1, 64, 1, None, None, 1, 64, 64, 64, 64, 0, 1, 0, 0, 0) != 0, (half, bad)
        assert l.setok_layernorm(None, bad, 1, 1, 1, 1, 4, 64, 1e-5) != 0, (half, bad)


def test_no_fallback_when_library_missing(lib, monkeypatch):
    monkeypatch.setattr(lib, "_libs", {})
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libsetok_hip.so")
    monkeypatch.setattr(lib, "LIB_PATH_F16", "/nonexistent/libsetok_hip_f16.so")
    with pytest.raises(lib.SetokHipError):
        lib.load()
    with pytest.raises(lib.SetokHipError):
        lib.load(half=True)


def test_driver_build_entry_point():
    """`__graft_entry__.build()` — what the driver runs as its build check: make + import + ABI version against the header."""
    import importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    importlib.import_module("__graft_entry__").build()
