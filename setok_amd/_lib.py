"""ctypes binding of libsetok_hip.so and libsetok_hip_f16.so (C ABI: include/setok_hip.h).

There is NO fallback: if the library is missing or a call fails this raises — the product path never
routes through the CPU oracle or plain torch ops.

Two builds of the same sources export the same ABI (include/setok_hip.h, `dtype`): libsetok_hip.so serves float32 + bfloat16,
libsetok_hip_f16.so float32 + float16 (the reference's inference loader casts the tower to torch.float16: src/model/builder.py:43,135-136).
`call` routes by the dtype code among the arguments — ops._code(torch.float16) returns the marked integer F16 — or by `half=True` for the
entry points whose element type is implied (setok_linear_ln, setok_ln_fold, the context calls)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SETOK_HIP_LIB") or os.path.join(_HERE, "libsetok_hip.so")   # SETOK_HIP_LIB: another build of the same ABI (A/B runs)
LIB_PATH_F16 = os.environ.get("SETOK_HIP_LIB_F16") or os.path.join(_HERE, "libsetok_hip_f16.so")


class _HalfCode(int):
    """SETOK_F16 as an int that also says WHICH library serves it: `call` sends every call carrying one to libsetok_hip_f16.so."""
    __slots__ = ()


F32, BF16 = 0, 1
F16 = _HalfCode(2)
_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
ACT_NONE, ACT_QUICK_GELU, ACT_GELU_ERF = 0, 1, 2

class SetokConfig(C.Structure):
    """`setok_config` of include/setok_hip.h, field for field."""
    _fields_ = [("image_size", _i), ("patch_size", _i), ("hidden_size", _i), ("intermediate_size", _i), ("num_hidden_layers", _i),
                ("num_attention_heads", _i), ("layer_norm_eps", _f), ("select_layer", _i), ("select_cls_patch", _i),
                ("token_feat_dim", _i), ("nheads", _i), ("dim_feedforward", _i), ("inner_cluster_layers", _i), ("intra_cluster_layers", _i),
                ("min_cluster_num", _i), ("threshold", _f), ("dtype", _i), ("fold_layernorm", _i)]


RESTYPES = {"setok_last_error": C.c_char_p, "setok_ctx_error": C.c_char_p, "setok_encode_workspace_bytes": _i64, "setok_destroy": None}

# name -> argtypes; mirrors include/setok_hip.h declaration by declaration
SIGNATURES = {
    "setok_profile_start": [],
    "setok_profile_stop": [_vp, _vp, _vp, _vp, _vp, _i],
    "setok_profile_pause": [_i],
    "setok_create": [C.POINTER(SetokConfig), C.POINTER(_vp)],
    "setok_destroy": [_vp],
    "setok_ctx_error": [_vp],
    "setok_load_weight": [_vp, _vp, C.c_char_p, _vp, _i, C.POINTER(_i64), _i],
    "setok_weights_ready": [_vp, _vp],
    "setok_encode_workspace_bytes": [_vp, _i],
    "setok_encode": [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(_i64),
                     C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    "setok_abi_version": [],
    "setok_last_error": [],
    "setok_device_info": [C.c_char_p, _i, C.POINTER(_i)],
    "setok_linear": [_vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i64, _i64, _i64],
    "setok_row_stats": [_vp, _i, _vp, _vp, _i, _i, _f],
    "setok_ln_fold": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i],
    "setok_linear_ln": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i],
    "setok_layernorm": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _f],
    "setok_activation": [_vp, _i, _vp, _vp, _i64, _i],
    "setok_dropout": [_vp, _i, _vp, _vp, _vp, _i64, _f, C.c_uint64, C.c_uint64],
    "setok_activation_dropout": [_vp, _i, _vp, _vp, _i64, _i, _f, C.c_uint64, C.c_uint64],
    "setok_attention": [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _f],
    "setok_cross_attention": [_vp, _i, _vp, _i64, _vp, _vp, _i64, _vp, _i, _i, _i, _vp, _i64, _i, _i, _f],
    "setok_patchify": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i],
    "setok_vit_assemble": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i],
    "setok_select_add_pos": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i],
    "setok_cluster_dpc_knn": [_vp, _i, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "setok_cluster_workspace": [_i, _i, _i, _i, C.POINTER(_i64), C.POINTER(_i64)],
    "setok_cluster_sort": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "setok_gather_rows": [_vp, _i, _vp, _vp, _vp, _i, _i],
    "setok_segment_mean": [_vp, _i, _vp, _vp, _vp, _i, _vp, _i],
    "setok_splice_lengths": [_vp, _vp, _vp, _i, _i, _i64, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "setok_splice_plan": [_vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _i64, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "setok_unpatchify": [_vp, _i, _vp, _i64, _vp, _i, _i, _i, _i],
    "setok_pixel_loss": [_vp, _i, _vp, _vp, _i64, _i, _vp, _vp],
    "setok_transpose": [_vp, _i, _vp, _i64, _i, _i, _vp, _i64, _i, _vp],
    "setok_colsum": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i],
    "setok_layernorm_bwd": [_vp, _i, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i],
    "setok_gelu_bwd": [_vp, _i, _vp, _vp, _vp, _i64],
    "setok_gelu_bwd_dropout": [_vp, _i, _vp, _vp, _vp, _i64, _f, C.c_uint64, C.c_uint64],
    "setok_attention_bwd": [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "setok_segment_mean_bwd": [_vp, _i, _vp, _vp, _vp, _i, _vp, _i],
    "setok_adamw": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _f],
    "setok_rmsnorm": [_vp, _i, _vp, _vp, _vp, _i, _i, _f],
    "setok_rope": [_vp, _i, _vp, _vp, _i, _i, _i, _f],
    "setok_rope_gqa": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f],
    "setok_swiglu": [_vp, _i, _vp, _vp, _i64, _i],
    "setok_swiglu_pairs": [_vp, _i, _vp, _vp, _i64, _i],
    "setok_linear_swiglu": [_vp, _i, _vp, _i64, _vp, _vp, _i64, _i, _i, _i],
    "setok_attention_causal": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f],
    "setok_attention_causal_gqa": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f],
    "setok_lm_loss": [_vp, _i, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "setok_splice_rows": [_vp, _i, _vp, _vp, _i, _vp, _i64, _vp, _i64, _i, _vp],
    "setok_splice_rows_bwd": [_vp, _i, _vp, _vp, _i64, _i, _vp, _i64, _vp, _i],
}

_libs = {}


class SetokHipError(RuntimeError):
    pass


def load(half: bool = False):
    """Load the HIP library (half=True: its fp16 build) once; raise loudly if it is not built."""
    lib = _libs.get(bool(half))
    if lib is not None:
        return lib
    path = LIB_PATH_F16 if half else LIB_PATH
    if not os.path.isfile(path):
        raise SetokHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C setok_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(path)                    # RTLD_LOCAL: the two builds export the same names and must not see each other's
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, _i)
    _libs[bool(half)] = lib
    return lib


def is_half(*args) -> bool:
    return any(type(a) is _HalfCode for a in args)


def call(name: str, *args, half: bool = False) -> None:
    lib = load(half or is_half(*args))
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SetokHipError(f"{name} failed (code {rc}): {lib.setok_last_error().decode()}")


def device_info():
    lib = load()
    buf = C.create_string_buffer(256)
    cu = _i(0)
    rc = lib.setok_device_info(buf, 256, C.byref(cu))
    if rc != 0:
        raise SetokHipError(lib.setok_last_error().decode())
    return buf.value.decode(), cu.value
