"""Cache of compute-ready weight copies (fused q|k|v matrices, fp32 biases / LayerNorm affines) with an invalidation
that cannot be bypassed.

A module's own `load_state_dict` override is NOT called when the weights arrive through a parent
(`SetokTokenizer.load_state_dict`, `load_pretrained_tokenizer`): nn.Module recurses with `_load_from_state_dict`.  The
load post-hook below IS part of that recursion, so it fires for every way of loading; `_apply` covers `.to()` / `.half()`;
and the key carries the `_version` counters of the parameters for in-place updates (an optimiser step, `copy_`)."""
from __future__ import annotations

from typing import Any, Dict, Iterable

import torch


class PackCacheMixin:
    def _init_pack_cache(self) -> None:
        self._packed: Dict[str, Any] = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_pack())

    def _drop_pack(self) -> None:
        self._packed = {}

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    @staticmethod
    def _versions(params: Iterable[torch.Tensor]) -> int:
        """Sum of the in-place modification counters: changes whenever any of the tensors is written in place."""
        return sum(p._version for p in params)
