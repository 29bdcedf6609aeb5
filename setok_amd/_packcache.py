"""Cache of compute-ready weight copies (fused q|k|v matrices, fp32 biases / LayerNorm affines) with an invalidation
that cannot be bypassed.

A module's own `load_state_dict` override is NOT called when the weights arrive through a parent
(`SetokTokenizer.load_state_dict`, `load_pretrained_tokenizer`): nn.Module recurses with `_load_from_state_dict`.  The
load post-hook below IS part of that recursion, so it fires for every way of loading; `_apply` covers `.to()` / `.half()`;
and the key carries the `_version` counters of the parameters for in-place updates (an optimiser step, `copy_`)."""
from __future__ import annotations

from typing import Any, Dict, Iterable

import torch


def _drop_pack_hook(module, incompatible) -> None:
    module._drop_pack()


class PackCacheMixin:
    def _init_pack_cache(self) -> None:
        self._packed: Dict[str, Any] = {}
        self.register_load_state_dict_post_hook(_drop_pack_hook)                # (a module-level function: lambdas do not pickle)

    def _drop_pack(self) -> None:
        self._packed = {}

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    @staticmethod
    def _versions(params: Iterable[torch.Tensor]) -> int:
        """Sum of the in-place modification counters: changes whenever any of the tensors is written in place."""
        return sum(p._version for p in params)


def f32_of(owner, name: str, t):
    """fp32 contiguous copy of parameter `t` (a bias, a LayerNorm affine), cached on `owner` under `name` and refreshed when the
    parameter is rewritten, moved or cast — so a forward pass does not re-materialise it on every call (a bf16 -> fp32 copy kernel per
    bias per call otherwise: 383 launches per encode step in round 1's profile)."""
    if t is None:
        return None
    if t.dtype == torch.float32 and t.is_contiguous():
        return t.detach()
    cache = owner.__dict__.setdefault("_f32_cache", {})
    key = (t.data_ptr(), t._version, t.dtype, str(t.device))
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = (key, t.detach().float().contiguous())
        cache[name] = hit
    return hit[1]
