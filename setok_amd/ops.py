"""Torch-tensor front end of the C ABI: pointer extraction, shape/dtype checks, output allocation.

PyTorch is plumbing here (device memory + streams); every operator below is one call into
libsetok_hip.so on torch's current HIP stream."""
from __future__ import annotations

import os

import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_GELU_ERF, ACT_NONE, ACT_QUICK_GELU, BF16, F16, F32  # noqa: F401

Tensor = torch.Tensor


def _code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16                               # (a marked int: _lib.call sends the call to libsetok_hip_f16.so)
    raise TypeError(f"setok_amd supports float32, bfloat16 and float16, got {dt}")


LOW = (torch.bfloat16, torch.float16)            # the 16-bit element types: MFMA throughput mode, fp32 accumulation


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The HIP stream torch is currently recording on, as an integer handle.  (`torch.cuda.current_stream()` builds a Python Stream object
    per call — 8 us, 1.5 ms of host time per encode at ~200 launches; the raw getter is 0.3 us.)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-contiguous tensor required"
    return t.data_ptr()


def _f32(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is not None:
        assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()
    return t


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def k_align(dt: torch.dtype) -> int:
    return 64 if dt in LOW else 16


# ---- optional launch profiler (bench.py): HIP events recorded INSIDE the library around every GEMM / clustering call ---------------------
_ACT_NAMES = ("plain", "quick_gelu", "gelu_erf")


def _profiled_libs():
    """(half, lib) of the library builds this process has loaded: the profiler's state is per library."""
    _lib.load()
    return sorted(_lib._libs.items())


def profile_start() -> None:
    for half, _ in _profiled_libs():
        _lib.call("setok_profile_start", half=half)


def profile_pause(pause: bool = True) -> None:
    """Between profile_start() and profile_stop(): stop (True) / resume (False) attaching events to launches; what was recorded stays."""
    for half, _ in _profiled_libs():
        _lib.call("setok_profile_pause", 1 if pause else 0, half=half)


def profile_stop():
    """Synchronise and return [{kernel, flops, ms, bytes}] for every launch the library recorded since profile_start()."""
    import numpy as np
    cap = 1 << 16
    out = []
    for half, lib in _profiled_libs():
        kind, cls = np.empty(cap, np.int32), np.empty(cap, np.int32)
        work, nbytes, ms = np.empty(cap, np.float64), np.empty(cap, np.float64), np.empty(cap, np.float32)
        n = lib.setok_profile_stop(kind.ctypes.data, cls.ctypes.data, work.ctypes.data, nbytes.ctypes.data, ms.ctypes.data, cap)
        if n < 0:
            raise _lib.SetokHipError("setok_profile_stop failed")
        low = "gemm_f16:" if half else "gemm_bf16:"
        for i in range(n):
            if kind[i] == 2:
                name, w = "cluster_dpc_knn", float(nbytes[i])            # the clustering record's headline quantity is algorithmic BYTES
            else:
                c = int(cls[i])
                name = (low if kind[i] == 0 else "gemm_f32:") + _ACT_NAMES[c & 3] + ("+residual" if c & 4 else "") + ("+layernorm" if c & 8 else "")
                w = float(work[i])
            out.append(dict(kernel=name, flops=w, ms=float(ms[i]), bytes=float(nbytes[i])))
    return out


def linear(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
           act: int = ACT_NONE, out: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """out = act(a @ w.T + bias) + residual;  a: (M, K), w: (N, K), bias fp32 (N,)."""
    M, K = a.shape
    N, K2 = w.shape
    assert K == K2 and a.dtype == w.dtype
    od = out_dtype or a.dtype
    if out is None:
        out = torch.empty((M, N), dtype=od, device=a.device)
    assert out.shape == (M, N) and out.dtype == od
    if residual is not None:
        assert residual.shape == (M, N) and residual.dtype == od
    _lib.call("setok_linear", _stream(), _code(a.dtype), _code(od), _p(a), K, _p(w), _p(_f32(bias)), _p(residual),
              _p(out), N, M, N, K, act, 1, 0, 0, 0)
    return out


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5, out: Optional[Tensor] = None) -> Tensor:
    rows, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.call("setok_layernorm", _stream(), _code(x.dtype), _p(x), _p(_f32(gamma)), _p(_f32(beta)), _p(out), rows, Cc, eps)
    return out


# ---- LayerNorm folded into the consuming Linear (bf16 throughput mode; include/setok_hip.h explains the algebra) -------------------------
def row_stats(x: Tensor, eps: float = 1e-5, out: Optional[Tensor] = None) -> Tensor:
    """(rows, 8) fp32 words per row: [0:2] the compact MFMA fragment of (-mean, 1 / rstd) (bf16 pairs), [4] rstd = 1 / sqrt(var + eps), [5] mean — the
    statistics setok_layernorm computes, without writing a normalised copy."""
    rows, Cc = x.shape
    if out is None:
        out = torch.empty((rows, 8), dtype=torch.float32, device=x.device)
    assert out.shape == (rows, 8) and out.dtype == torch.float32
    _lib.call("setok_row_stats", _stream(), _code(x.dtype), _p(x), _p(out), rows, Cc, eps)
    return out


def ln_fold(w: Tensor, gamma: Tensor, beta: Tensor, bias: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Once per weight load: (W' = bf16(gamma * W), c = W' 1, b' = b + W beta, the (N, 4)-word MFMA fragments of (c, b')) for linear_ln."""
    assert w.dtype in LOW and w.dim() == 2
    N, K = w.shape
    wg = torch.empty_like(w)
    cs = torch.empty((N,), dtype=torch.float32, device=w.device)
    bf = torch.empty((N,), dtype=torch.float32, device=w.device)
    fr = torch.empty((N, 4), dtype=torch.float32, device=w.device)
    _lib.call("setok_ln_fold", _stream(), _p(w.contiguous()), _p(_f32(gamma)), _p(_f32(beta)), _p(_f32(bias)), _p(wg), _p(cs), _p(bf), _p(fr), N, K,
              half=w.dtype == torch.float16)
    return wg, cs, bf, fr


def linear_ln(a: Tensor, folded: Tuple[Tensor, Tensor, Tensor, Tensor], stats: Tensor, act: int = ACT_NONE, out: Optional[Tensor] = None) -> Tensor:
    """out = act(LayerNorm(a) @ W.T + b) from the raw rows `a`, their statistics and the folded weights of ln_fold."""
    wg, _, _, fr = folded
    M, K = a.shape
    N = wg.shape[0]
    assert a.dtype in LOW and wg.dtype == a.dtype and wg.shape[1] == K and stats.shape == (M, 8)
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    _lib.call("setok_linear_ln", _stream(), _p(a), K, _p(wg), _p(fr), _p(stats), _p(out), N, M, N, K, act, half=a.dtype == torch.float16)
    return out


def attention(qkv: Tensor, H: int, Dh: int, scale: float, seg_len: int, seg_offsets: Optional[Tensor] = None,
              n_segs: int = 0, out: Optional[Tensor] = None) -> Tensor:
    """Block-diagonal attention.  Uniform segments of `seg_len` rows, or ragged ones given by the int32
    device tensor `seg_offsets` (n_segs + 1 entries; `seg_len` is then an upper bound on the length)."""
    rows = qkv.shape[0]
    assert qkv.shape[1] == 3 * H * Dh
    if out is None:
        out = torch.empty((rows, H * Dh), dtype=qkv.dtype, device=qkv.device)
    if seg_offsets is not None:
        assert seg_offsets.dtype == torch.int32 and seg_offsets.numel() >= n_segs + 1
    _lib.call("setok_attention", _stream(), _code(qkv.dtype), _p(qkv), _p(seg_offsets), n_segs, seg_len, _p(out),
              rows, H, Dh, scale)
    return out


def cross_attention(q: Tensor, k: Tensor, v: Tensor, H: int, Dh: int, scale: float, q_len: int, kv_offsets: Optional[Tensor],
                    n_segs: int, max_kv: int, out: Optional[Tensor] = None) -> Tensor:
    """Cross-attention of `n_segs` groups of `q_len` query rows to ragged key / value segments (`kv_offsets`: int32
    device tensor of n_segs + 1 row offsets into k / v; None = uniform segments of `max_kv` rows).  q, k, v are 2-D row views
    (column windows of wider buffers are fine: only the row stride is passed down)."""
    assert q.dim() == 2 and k.dim() == 2 and v.dim() == 2 and q.shape[1] == H * Dh and k.shape[1] == H * Dh and v.shape[1] == H * Dh
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1 and k.stride(0) == v.stride(0) and k.dtype == q.dtype == v.dtype
    assert q.shape[0] == n_segs * q_len
    if out is None:
        out = torch.empty((q.shape[0], H * Dh), dtype=q.dtype, device=q.device)
    if kv_offsets is not None:
        assert kv_offsets.dtype == torch.int32 and kv_offsets.numel() >= n_segs + 1
    assert q.is_cuda and k.is_cuda and v.is_cuda
    _lib.call("setok_cross_attention", _stream(), _code(q.dtype), q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(0), _p(kv_offsets), n_segs,
              q_len, max_kv, _p(out), out.stride(0), H, Dh, scale)
    return out


def patchify(images: Tensor, p: int, kpad: int) -> Tensor:
    B, Cin, H, W = images.shape
    assert Cin == 3
    out = torch.empty((B * (H // p) * (W // p), kpad), dtype=images.dtype, device=images.device)
    _lib.call("setok_patchify", _stream(), _code(images.dtype), _p(images), _p(out), B, H, W, p, kpad)
    return out


def vit_assemble(patch_embed: Tensor, cls: Tensor, pos: Tensor, B: int, N: int) -> Tensor:
    Cc = patch_embed.shape[1]
    out = torch.empty((B * (N + 1), Cc), dtype=patch_embed.dtype, device=patch_embed.device)
    _lib.call("setok_vit_assemble", _stream(), _code(patch_embed.dtype), _p(patch_embed), _p(cls), _p(pos), _p(out), B, N, Cc)
    return out


def select_add_pos(hidden: Tensor, pos2d: Tensor, B: int, N: int, skip: int) -> Tensor:
    Cc = hidden.shape[-1]
    assert hidden.numel() == B * (N + skip) * Cc and pos2d.shape == (N, Cc) and pos2d.dtype == hidden.dtype
    out = torch.empty((B * N, Cc), dtype=hidden.dtype, device=hidden.device)
    _lib.call("setok_select_add_pos", _stream(), _code(hidden.dtype), _p(hidden), _p(pos2d), _p(out), B, N, Cc, skip)
    return out


def cluster_dpc_knn(x: Tensor, B: int, N: int, k: int, threshold: float, min_cluster_num: int,
                    noise: Optional[Tensor] = None, token_mask: Optional[Tensor] = None
                    ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """x: (B*N, C).  Returns (idx_cluster int64 (B,N), score fp32 (B,N), index_down int64 (B,N) [-1 padded],
    counts int32 (B,)).  No host synchronisation."""
    Cc = x.shape[-1]
    assert x.numel() == B * N * Cc
    dev = x.device
    idx = torch.empty((B, N), dtype=torch.int64, device=dev)
    score = torch.empty((B, N), dtype=torch.float32, device=dev)
    index_down = torch.empty((B, N), dtype=torch.int64, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    nd, nv = _lib.C.c_int64(0), _lib.C.c_int64(0)
    _lib.call("setok_cluster_workspace", _code(x.dtype), B, N, Cc, _lib.C.byref(nd), _lib.C.byref(nv))
    dist_ws = torch.empty((nd.value,), dtype=torch.float32, device=dev) if nd.value else None       # 0: the fused single-launch form
    vec_ws = torch.empty((nv.value,), dtype=torch.float32, device=dev) if nv.value else None
    if noise is not None:
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        assert noise.numel() == B * N
    if token_mask is not None:
        token_mask = token_mask.to(device=dev, dtype=torch.float32).contiguous()
        assert token_mask.numel() == B * N
    _lib.call("setok_cluster_dpc_knn", _stream(), _code(x.dtype), _p(x), B, N, Cc, int(k), float(threshold),
              int(min_cluster_num), _p(noise), _p(token_mask), _p(idx), _p(score), _p(index_down), _p(counts),
              _p(dist_ws), _p(vec_ws))
    return idx, score, index_down, counts


def cluster_sort(idx_cluster: Tensor, counts: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    B, N = idx_cluster.shape
    dev = idx_cluster.device
    perm = torch.empty((B * N,), dtype=torch.int32, device=dev)
    seg_offsets = torch.empty((B * N + 1,), dtype=torch.int32, device=dev)
    img_offsets = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    _lib.call("setok_cluster_sort", _stream(), _p(idx_cluster), _p(counts), B, N, _p(perm), _p(seg_offsets), _p(img_offsets))
    return perm, seg_offsets, img_offsets


def gather_rows(x: Tensor, perm: Tensor) -> Tensor:
    rows, Cc = perm.numel(), x.shape[-1]
    out = torch.empty((rows, Cc), dtype=x.dtype, device=x.device)
    _lib.call("setok_gather_rows", _stream(), _code(x.dtype), _p(x), _p(perm), _p(out), rows, Cc)
    return out


def segment_mean(h: Tensor, seg_offsets: Tensor, n_segs_dev: Tensor, n_segs: int) -> Tensor:
    """n_segs_dev: 1-element int32 device tensor (e.g. img_offsets[B:]); n_segs: rows to allocate/launch."""
    Cc = h.shape[-1]
    out = torch.empty((n_segs, Cc), dtype=h.dtype, device=h.device)
    _lib.call("setok_segment_mean", _stream(), _code(h.dtype), _p(h), _p(seg_offsets), _p(n_segs_dev), n_segs, _p(out), Cc)
    return out


def activation(x: Tensor, act: int, out: Optional[Tensor] = None) -> Tensor:
    if out is None:
        out = torch.empty_like(x)
    _lib.call("setok_activation", _stream(), _code(x.dtype), _p(x), _p(out), x.numel(), act)
    return out


def activation_dropout(x: Tensor, act: int, p: float, seed: int, offset: int = 0, out: Optional[Tensor] = None) -> Tensor:
    """dropout(activation(x)) in one pass (bit-identical to the two calls)."""
    if out is None:
        out = torch.empty_like(x)
    _lib.call("setok_activation_dropout", _stream(), _code(x.dtype), _p(x), _p(out), x.numel(), act, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF,
              int(offset) & 0xFFFFFFFFFFFFFFFF)
    return out


def dropout(x: Tensor, p: float, seed: int, offset: int = 0, residual: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """out = residual + x * mask / (1 - p), mask_i a pure function of (seed, offset + i) (include/setok_hip.h: setok_dropout).  Applying the same
    call (no residual) to an incoming gradient is the backward pass."""
    if out is None:
        out = torch.empty_like(x)
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape and (residual is None or (residual.shape == x.shape and residual.is_contiguous()))
    _lib.call("setok_dropout", _stream(), _code(x.dtype), _p(x), _p(residual), _p(out), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF,
              int(offset) & 0xFFFFFFFFFFFFFFFF)
    return out


# ---- prepare_inputs_labels_for_multimodal (setokim_arch.py:213-355) --------------------------------------------------------
def splice_lengths(input_ids: Tensor, attention_mask: Optional[Tensor], img_offsets: Tensor, n_images: int, image_token_index: int,
                   max_length: int, vocab: int = 0) -> Tuple[Tensor, Tensor, Tensor]:
    """Returns (seq_len int32 (B,), img_start int32 (B,), status int32 (4,)) on the device; no host synchronisation."""
    B, T = input_ids.shape
    dev = input_ids.device
    assert input_ids.dtype == torch.int64 and img_offsets.dtype == torch.int32 and img_offsets.numel() >= n_images + 1
    if attention_mask is not None:
        assert attention_mask.dtype == torch.uint8 and attention_mask.shape == (B, T)
    seq_len = torch.empty((B,), dtype=torch.int32, device=dev)
    img_start = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.empty((4,), dtype=torch.int32, device=dev)
    ws = torch.empty((3 * B,), dtype=torch.int32, device=dev)
    _lib.call("setok_splice_lengths", _stream(), _p(input_ids), _p(attention_mask), B, T, image_token_index, int(vocab), _p(img_offsets), n_images,
              int(max_length), _p(seq_len), _p(img_start), _p(status), _p(ws))
    return seq_len, img_start, status


def splice_plan(input_ids: Tensor, attention_mask: Optional[Tensor], labels: Optional[Tensor], img_offsets: Tensor, seq_len: Tensor,
                img_start: Tensor, max_len: int, left_pad: bool, image_token_index: int, ignore_index: int, target_token_index: int,
                want_mask: bool, want_pos: bool):
    B, T = input_ids.shape
    dev = input_ids.device
    src = torch.empty((B, max_len), dtype=torch.int32, device=dev)
    new_labels = torch.empty((B, max_len), dtype=torch.int64, device=dev) if labels is not None else None
    new_mask = torch.empty((B, max_len), dtype=torch.uint8, device=dev) if want_mask else None
    new_pos = torch.empty((B, max_len), dtype=torch.int64, device=dev) if want_pos else None
    if labels is not None:
        assert labels.dtype == torch.int64 and labels.shape == (B, T)
    _lib.call("setok_splice_plan", _stream(), _p(input_ids), _p(attention_mask), _p(labels), B, T, image_token_index, ignore_index,
              target_token_index, _p(img_offsets), _p(seq_len), _p(img_start), max_len, 1 if left_pad else 0, _p(src), _p(new_labels),
              _p(new_mask), _p(new_pos))
    return src, new_labels, new_mask, new_pos


def splice_rows(src: Tensor, embed_table: Tensor, image_tokens: Optional[Tensor], status: Optional[Tensor] = None) -> Tensor:
    """status: optional int32[2] device tensor — [0] = 1 / [1] = first row whose src neither table can serve (such rows are zero-filled)."""
    B, max_len = src.shape
    V, D = embed_table.shape
    out = torch.empty((B, max_len, D), dtype=embed_table.dtype, device=embed_table.device)
    n_img_rows = 0
    if image_tokens is not None:
        assert image_tokens.dtype == embed_table.dtype and image_tokens.shape[-1] == D
        n_img_rows = image_tokens.shape[0]
    if status is not None:
        assert status.dtype == torch.int32 and status.numel() >= 2
    _lib.call("setok_splice_rows", _stream(), _code(embed_table.dtype), _p(src), _p(embed_table), V, _p(image_tokens), n_img_rows, _p(out),
              B * max_len, D, _p(status))
    return out


def splice_rows_bwd(src: Tensor, d_out: Tensor, n_image_rows: int, want_embed: int = 0) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """(d image_tokens (n_image_rows, D) in d_out's dtype, d embed_table (want_embed, D) fp32 or None)."""
    rows, D = src.numel(), d_out.shape[-1]
    d_out = d_out.reshape(rows, D)
    assert d_out.is_contiguous()
    dfe = torch.empty((n_image_rows, D), dtype=d_out.dtype, device=d_out.device) if n_image_rows > 0 else None
    dem = torch.zeros((want_embed, D), dtype=torch.float32, device=d_out.device) if want_embed > 0 else None
    _lib.call("setok_splice_rows_bwd", _stream(), _code(d_out.dtype), _p(src), _p(d_out), rows, D, _p(dfe), n_image_rows, _p(dem), want_embed)
    return dfe, dem


# ---- pixel head of the reconstruction decoder (include/setok_hip.h: the output the reference never defines) ---------------------------------
def unpatchify(patches: Tensor, B: int, gh: int, gw: int, p: int) -> Tensor:
    """patches (B*gh*gw, >= 3 p^2) -> image (B, 3, gh*p, gw*p); only the first 3 p^2 columns are read (a padded GEMM output is fine)."""
    assert patches.dim() == 2 and patches.shape[0] == B * gh * gw and patches.shape[1] >= 3 * p * p and patches.stride(1) == 1
    img = torch.empty((B, 3, gh * p, gw * p), dtype=patches.dtype, device=patches.device)
    assert patches.is_cuda
    _lib.call("setok_unpatchify", _stream(), _code(patches.dtype), patches.data_ptr(), patches.stride(0), _p(img), B, gh, gw, p)
    return img


def pixel_loss(pred: Tensor, target: Tensor, kind: str = "mse") -> Tensor:
    """0-d fp32 tensor: mean squared ("mse") or mean absolute ("l1") difference over all elements."""
    assert pred.shape == target.shape and pred.dtype == target.dtype and kind in ("mse", "l1")
    out = torch.empty((1,), dtype=torch.float32, device=pred.device)
    _lib.call("setok_pixel_loss", _stream(), _code(pred.dtype), _p(pred.contiguous()), _p(target.contiguous()), pred.numel(), 0 if kind == "mse" else 1,
              _p(_ws(pred.device, 1024)), _p(out))
    return out[0]


# ---- backward pass of the trainable head (csrc/backward.hip) ----------------------------------------------------------------
def transpose(x: Tensor, pad_to: int = 1, splits: int = 1, with_colsum: bool = False, colsum_out: Optional[Tensor] = None):
    """(rows, cols) -> (cols, P) with P = rows zero-padded to a multiple of pad_to: the A / W operand of a dW = dY^T X GEMM.
    splits > 1: P is padded to splits * chunk (chunk a multiple of pad_to) and the result is (splits, cols, chunk) — the split-K layout.
    with_colsum: also returns the fp32 column sums of x (the bias gradient when x is dY), computed inside the same pass."""
    rows, cols = x.shape
    assert x.is_contiguous()
    if splits <= 1:
        ldo, chunk = round_up(max(rows, 1), pad_to), 0
        out = torch.empty((cols, ldo), dtype=x.dtype, device=x.device)
    else:
        chunk = round_up((max(rows, 1) + splits - 1) // splits, pad_to)
        ldo = splits * chunk
        out = torch.empty((splits, cols, chunk), dtype=x.dtype, device=x.device)
    part = torch.empty(((ldo + 63) // 64, cols), dtype=torch.float32, device=x.device) if with_colsum else None
    _lib.call("setok_transpose", _stream(), _code(x.dtype), _p(x), cols, rows, cols, _p(out), ldo, chunk, _p(part))
    if not with_colsum:
        return out
    return out, colsum(part, out=colsum_out)


def linear_tn(aT: Tensor, bT: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """dW = sum_s aT[s] @ bT[s]^T in fp32 for split-K operands aT (S, N, chunk), bT (S, K, chunk) from `transpose(..., splits=S)`
    (or 2-D (N, P), (K, P) for S = 1): one batched GEMM for the partial products, then a fixed-order sum over the splits."""
    if aT.dim() == 2:
        return linear(aT, bT, out=out, out_dtype=torch.float32)
    S, N, chunk = aT.shape
    S2, K, chunk2 = bT.shape
    assert S == S2 and chunk == chunk2 and aT.dtype == bT.dtype
    part = torch.empty((S, N, K), dtype=torch.float32, device=aT.device)
    _lib.call("setok_linear", _stream(), _code(aT.dtype), F32, _p(aT), chunk, _p(bT), None, None, _p(part), K, N, K, chunk, ACT_NONE, S,
              N * chunk, K * chunk, N * K)
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=aT.device)
    assert out.shape == (N, K) and out.dtype == torch.float32 and out.is_contiguous()
    ws = _ws(aT.device, N * K)
    _lib.call("setok_colsum", _stream(), F32, _p(part), S, N * K, _p(out), 0, _p(ws), 1)
    return out


_WS: dict = {}


def _ws(device, n: int) -> Tensor:
    """fp32 scratch reused across calls on the same stream (kernels of one stream run in order)."""
    key = str(device)
    if key not in _WS or _WS[key].numel() < n:
        _WS[key] = torch.empty((max(n, 1 << 20),), dtype=torch.float32, device=device)
    return _WS[key]


def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    rows, cols = x.shape
    if out is None:
        out = torch.empty((cols,), dtype=torch.float32, device=x.device)
        assert not accumulate
    ws_rows = 128
    ws = _ws(x.device, ws_rows * cols)
    _lib.call("setok_colsum", _stream(), _code(x.dtype), _p(x), rows, cols, _p(out), 1 if accumulate else 0, _p(ws), ws_rows)
    return out


def layernorm_bwd(x: Tensor, dy: Tensor, gamma: Tensor, eps: float, dgamma: Tensor, dbeta: Tensor, accumulate: bool,
                  need_dx: bool = True, res: Optional[Tensor] = None) -> Optional[Tensor]:
    rows, Cc = x.shape
    assert dy.shape == x.shape and dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32
    dx = torch.empty_like(x) if need_dx else None
    ws_rows = 2048
    ws = _ws(x.device, ws_rows * Cc)
    _lib.call("setok_layernorm_bwd", _stream(), _code(x.dtype), _p(x), _p(dy), _p(_f32(gamma)), eps, rows, Cc, _p(dx), _p(res),
              _p(dgamma), _p(dbeta), 1 if accumulate else 0, _p(ws), ws_rows)
    return dx


def gelu_bwd(pre: Tensor, dy: Tensor) -> Tensor:
    dx = torch.empty_like(pre)
    _lib.call("setok_gelu_bwd", _stream(), _code(pre.dtype), _p(pre), _p(dy), _p(dx), pre.numel())
    return dx


def gelu_bwd_dropout(pre: Tensor, dy: Tensor, p: float, seed: int, offset: int = 0, out: Optional[Tensor] = None) -> Tensor:
    """gelu'(pre) * dropout(dy) in one pass (bit-identical to dropout(dy) then gelu_bwd); `out` may be `dy`."""
    dx = torch.empty_like(pre) if out is None else out
    _lib.call("setok_gelu_bwd_dropout", _stream(), _code(pre.dtype), _p(pre), _p(dy), _p(dx), pre.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF,
              int(offset) & 0xFFFFFFFFFFFFFFFF)
    return dx


def attention_bwd(qkv: Tensor, out: Tensor, dout: Tensor, H: int, Dh: int, scale: float, seg_len: int,
                  seg_offsets: Optional[Tensor] = None, n_segs: int = 0) -> Tensor:
    rows = qkv.shape[0]
    dqkv = torch.empty_like(qkv)
    ws = _ws(qkv.device, 2 * rows * H)
    _lib.call("setok_attention_bwd", _stream(), _code(qkv.dtype), _p(qkv), _p(seg_offsets), n_segs, seg_len, _p(out), _p(dout), _p(dqkv),
              rows, H, Dh, scale, _p(ws))
    return dqkv


def segment_mean_bwd(dseg: Tensor, seg_offsets: Tensor, n_segs_dev: Tensor, n_segs: int, rows: int) -> Tensor:
    Cc = dseg.shape[-1]
    out = torch.empty((rows, Cc), dtype=dseg.dtype, device=dseg.device)
    _lib.call("setok_segment_mean_bwd", _stream(), _code(dseg.dtype), _p(dseg), _p(seg_offsets), _p(n_segs_dev), n_segs, _p(out), Cc)
    return out


def adamw(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, param_lp: Optional[Tensor], lr: float, beta1: float,
          beta2: float, eps: float, weight_decay: float, step: int, grad_scale: float = 1.0) -> None:
    assert param.dtype == grad.dtype == exp_avg.dtype == exp_avg_sq.dtype == torch.float32
    n = param.numel()
    assert grad.numel() == n and (param_lp is None or param_lp.numel() == n)
    lp = _code(param_lp.dtype) if param_lp is not None else 0
    _lib.call("setok_adamw", _stream(), lp, _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(param_lp), n, lr, beta1, beta2, eps,
              weight_decay, step, grad_scale)


# ---- LLM prefill (csrc/llama.hip) ------------------------------------------------------------------------------------------------
def rmsnorm(x: Tensor, weight: Tensor, eps: float, out: Optional[Tensor] = None) -> Tensor:
    rows, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    w = weight if weight.dtype == torch.float32 else weight.float()       # a bf16 weight is exact in fp32; the kernel re-rounds it to the activation dtype
    _lib.call("setok_rmsnorm", _stream(), _code(x.dtype), _p(x), _p(w.contiguous()), _p(out), rows, Cc, eps)
    return out


def rope_(qkv: Tensor, position_ids: Tensor, H: int, Dh: int, theta: float, Hkv: Optional[int] = None) -> Tensor:
    """In place on the q and k parts of qkv (rows, (H + 2*Hkv)*Dh) laid out [q | k | v]; Hkv (grouped-query attention) defaults to H."""
    rows = qkv.shape[0]
    Hkv = H if Hkv is None else Hkv
    assert qkv.shape[1] == (H + 2 * Hkv) * Dh and position_ids.dtype == torch.int64 and position_ids.numel() == rows
    _lib.call("setok_rope_gqa", _stream(), _code(qkv.dtype), _p(qkv), _p(position_ids.contiguous()), rows, H, Hkv, Dh, theta)
    return qkv


def swiglu(gate_up: Tensor) -> Tensor:
    rows, F2 = gate_up.shape
    out = torch.empty((rows, F2 // 2), dtype=gate_up.dtype, device=gate_up.device)
    _lib.call("setok_swiglu", _stream(), _code(gate_up.dtype), _p(gate_up), _p(out), rows, F2 // 2)
    return out


def swiglu_pairs(gate_up_pairs: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """act_fn(gate) * up on INTERLEAVED (gate_j, up_j) columns — the output of `linear` with pair-interleaved weight rows (interleave_gate_up)."""
    rows, F2 = gate_up_pairs.shape
    if out is None:
        out = torch.empty((rows, F2 // 2), dtype=gate_up_pairs.dtype, device=gate_up_pairs.device)
    _lib.call("setok_swiglu_pairs", _stream(), _code(gate_up_pairs.dtype), _p(gate_up_pairs), _p(out), rows, F2 // 2)
    return out


def interleave_gate_up(gate_w: Tensor, up_w: Tensor) -> Tensor:
    """(F, K), (F, K) -> (2 F, K) with row 2 j = gate_w[j], row 2 j + 1 = up_w[j]: the weight layout of linear_swiglu."""
    assert gate_w.shape == up_w.shape
    return torch.stack([gate_w, up_w], dim=1).reshape(2 * gate_w.shape[0], gate_w.shape[1]).contiguous()


def linear_swiglu(a: Tensor, w_pairs: Tensor) -> Tensor:
    """act_fn(a @ Wg.T) * (a @ Wu.T) for pair-interleaved weights (interleave_gate_up): LlamaMLP's gate|up Linear + SwiGLU.  In the 16-bit modes the whole 256-row
    tiles go through ONE launch (SwiGLU in the GEMM's epilogue: the (M, 2 F) intermediate is never written); the rows behind them — and everything when the
    problem is too small for the persistent kernel, or in fp32 — through `linear` + `swiglu_pairs`.  The two forms give the same bits (the epilogue keeps torch's
    rounding points), so a row's result does not depend on which of them it took."""
    M, K = a.shape
    F2 = w_pairs.shape[0]
    out = torch.empty((M, F2 // 2), dtype=a.dtype, device=a.device)
    main = 0
    if os.environ.get("SETOK_SWIGLU_FUSED", "1") != "0" and a.dtype in LOW and F2 % 256 == 0 and K % 64 == 0 and K >= 128 and (M // 256) * (F2 // 256) >= 90:
        main = M // 256 * 256
        _lib.call("setok_linear_swiglu", _stream(), _code(a.dtype), _p(a), K, _p(w_pairs), _p(out), F2 // 2, main, F2 // 2, K)
    if main < M:
        swiglu_pairs(linear(a[main:], w_pairs), out=out[main:])
    return out


def lm_loss(logits: Tensor, labels: Tensor, attention_mask: Optional[Tensor], ignore_index: int = -100) -> Tensor:
    """The shifted, padding-masked cross entropy of setokim_llama.py:145-160.  logits (B, T, V) fp32 / bf16 (read as fp32), labels (B, T)
    int64, attention_mask (B, T) or None.  Returns a 2-element fp32 tensor: [mean loss, number of positions that counted]."""
    B, T, V = logits.shape
    lg = logits.reshape(B * T, V)
    assert lg.stride(1) == 1
    lab = labels.to(device=logits.device, dtype=torch.int64).reshape(B * T).contiguous()
    am = None if attention_mask is None else (attention_mask.to(logits.device) != 0).to(torch.uint8).reshape(B * T).contiguous()
    ws = torch.empty(2 * B * T, dtype=torch.float32, device=logits.device)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    _lib.call("setok_lm_loss", _stream(), _code(logits.dtype), lg.data_ptr(), lg.stride(0), _p(lab), _p(am), B, T, V, ignore_index, _p(ws), _p(out))
    return out


def attention_causal(qkv: Tensor, key_mask: Optional[Tensor], B: int, T: int, H: int, Dh: int, scale: float, Hkv: Optional[int] = None) -> Tensor:
    """Causal, key-padding-masked attention over B sequences of T rows [q: H heads | k: Hkv | v: Hkv]; Hkv < H: grouped-query attention."""
    Hkv = H if Hkv is None else Hkv
    assert qkv.shape == (B * T, (H + 2 * Hkv) * Dh) and H % Hkv == 0
    if key_mask is not None:
        assert key_mask.dtype == torch.uint8 and key_mask.numel() == B * T
    out = torch.empty((B * T, H * Dh), dtype=qkv.dtype, device=qkv.device)
    _lib.call("setok_attention_causal_gqa", _stream(), _code(qkv.dtype), _p(qkv), _p(key_mask), _p(out), B, T, H, Hkv, Dh, scale)
    return out
