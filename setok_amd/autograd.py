"""Autograd through the drop-in boundary.

The reference's main caller of `encode_images` is a TRAINING forward: stage 2 trains `mm_in_projector` through the LLM loss with the
tokenizer frozen (scripts/pretrain_mm_proj.sh:40, src/train/train_setokim.py:336-339), the projected tokens travel through
`prepare_inputs_labels_for_multimodal` into `inputs_embeds` (src/model/setokim_arch.py:206-211,290-293) and torch autograd carries
dL/dinputs_embeds back.  Stage 1 trains the tokenizer's head itself.  The reference gets all of that for free from torch ops; the modules
here run on the HIP library, so the three places where a gradient has to cross the library are `torch.autograd.Function`s whose backward
is made of the same entry points the hand-written training step uses (setok_linear for dX, the split-K `linear_tn` for dW, setok_colsum
inside the transposes for db, setok_gelu_bwd, setok_layernorm_bwd, setok_splice_rows_bwd, training.head_backward):

    ProjectorFn   VisionProjector / LinearProjector              d loss / d projector parameters, d loss / d tokens
    SpliceRowsFn  splice_multimodal (the embedding + image rows)    d loss / d projected image tokens (d embed_tokens.weight on request)
    HeadFn        SetokTokenizer.encode_features                    d loss / d head parameters (inner_encoder, inter_encoder, out)

Everything else on the path is inference-only and says so (round 4): the forward always runs — reference-style inference callers
do not wrap their calls in no_grad and freshly built modules have trainable parameters — and, when torch would have recorded a graph, the
outputs carry a grad_fn whose backward RAISES (`no_backward`), so nothing trains silently.  The CLIP tower, whose reference forward is itself
`@torch.no_grad()`, warns once and proceeds.  `refuse_grad` (raise at the call) is left for inputs a caller explicitly marked as needing a
gradient that cannot be delivered.  The forward values are the inference
path's, bit for bit, for the projector and the splice; the head's training forward keeps its pre-activation (training.py)."""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import ops
from ._packcache import f32_of


def grad_needed(*tensors) -> bool:
    """Would torch autograd record a graph for an op over these tensors?"""
    if not torch.is_grad_enabled():
        return False
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.requires_grad:
            return True
    return False


def refuse_grad(what: str, tensors: Iterable[Any], hint: str = "") -> None:
    """The loud guard for an INPUT a caller explicitly asked a gradient for and cannot get (features handed to the head's training step with
    `requires_grad=True`, ...): raising at the call is the only honest answer there."""
    if not torch.is_grad_enabled():
        return
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.requires_grad:
            raise NotImplementedError(
                f"{what} runs on the HIP library without an autograd graph, but gradients are enabled and one of its inputs / parameters "
                f"requires a gradient: the result would silently train nothing.  Freeze it (`requires_grad_(False)`, what the reference's "
                f"stage-2 recipe does for everything but mm_in_projector) or call it under `torch.no_grad()`." + (" " + hint if hint else ""))


class _NoBackwardFn(torch.autograd.Function):
    """Identity whose backward raises: the grad_fn of an inference-only module's output when autograd would have expected a graph."""

    @staticmethod
    def forward(ctx, out, what, *deps):
        ctx.what = what
        return out.view_as(out)

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError(
            f"{ctx.what} ran on the HIP library without an autograd graph (it has no backward pass there): backward() through its output "
            f"would silently train nothing.  Freeze the module (`requires_grad_(False)`), call it under `torch.no_grad()`, or detach its output.")


def no_backward(what: str, value, deps: Iterable[Any]):
    """Inference-only modules whose REFERENCE counterpart is differentiable (Block, SetokDeTokenizer, the LLM prefill): the forward runs as
    inference whatever the grad mode — a freshly built nn.Module has `requires_grad=True` parameters and reference-style inference callers do
    not wrap their calls — and, where torch would have recorded a graph (gradients enabled + an input / parameter requiring one),
    every floating-point tensor of the result carries a grad_fn that RAISES in backward.  Nothing trains silently, nothing that only wants the
    values crashes."""
    if not torch.is_grad_enabled():
        return value
    live = [t for t in deps if isinstance(t, torch.Tensor) and t.requires_grad]
    if not live:
        return value

    def wrap(v):
        if isinstance(v, torch.Tensor):
            return _NoBackwardFn.apply(v, what, *live) if v.is_floating_point() else v
        if isinstance(v, tuple):
            return tuple(wrap(u) for u in v)
        if isinstance(v, list):
            return [wrap(u) for u in v]
        if isinstance(v, dict):
            return {k: wrap(u) for k, u in v.items()}
        return v
    return wrap(value)


_WARNED: set = set()


def warn_no_grad_once(what: str, tensors: Iterable[Any], why: str, owner: Any = None, inputs: Iterable[Any] = ()) -> None:
    """For a module whose REFERENCE forward is itself `@torch.no_grad()` (the CLIP tower, clip_encoder.py:50): parameters that require a
    gradient get none there either, so the call proceeds exactly like the reference's — with one warning PER MODULE INSTANCE (`owner`; keyed
    by name alone, a second tower with unfrozen parameters went through silently: ADVICE r04) saying so.  `inputs` (e.g. the images) that
    require a gradient get a warning of their own, worded for what it is."""
    if not torch.is_grad_enabled():
        return
    import warnings
    key = (what, id(owner) if owner is not None else None)
    if key + ("params",) not in _WARNED:
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.requires_grad:
                _WARNED.add(key + ("params",))
                warnings.warn(f"{what}: gradients are enabled and a parameter requires one, but {why}; running without a graph.", UserWarning, stacklevel=3)
                break
    if key + ("inputs",) not in _WARNED:
        for t in inputs:
            if isinstance(t, torch.Tensor) and t.requires_grad:
                _WARNED.add(key + ("inputs",))
                warnings.warn(f"{what}: an INPUT requires a gradient, but {why}; no gradient will reach it.", UserWarning, stacklevel=3)
                break


def _grad_as(g: Optional[torch.Tensor], like: torch.Tensor) -> Optional[torch.Tensor]:
    if g is None:
        return None
    return g.to(like.dtype).reshape(like.shape)


# ----------------------------------------------------------------------------------------------------------------------------
# mm_in_projector (src/model/multimodal_projector/builder.py:33-59): Linear [LayerNorm] (GELU Linear)*
# ----------------------------------------------------------------------------------------------------------------------------
class ProjectorFn(torch.autograd.Function):
    """y = projector(x) over rows.  `plan`: tuple of ("linear", fused_gelu, has_bias) | ("ln", eps) | ("gelu",) in module order; `tensors`:
    the parameters in that order (weight, bias per Linear / LayerNorm; a missing bias is simply absent)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, plan: Tuple, owner, *tensors: torch.Tensor):
        h = x.contiguous()
        saved: List[torch.Tensor] = []
        i = 0
        for li, step in enumerate(plan):
            saved.append(h)
            if step[0] == "linear":
                w = tensors[i].detach().contiguous(); i += 1
                b = None
                if step[2]:
                    b = f32_of(owner, f"autograd_bias_{li}", tensors[i]); i += 1
                h = ops.linear(h, w, b, act=ops.ACT_GELU_ERF if step[1] else ops.ACT_NONE)      # the inference path's call: same bits
            elif step[0] == "ln":
                h = ops.layernorm(h, f32_of(owner, f"autograd_lnw_{li}", tensors[i]), f32_of(owner, f"autograd_lnb_{li}", tensors[i + 1]), step[1]); i += 2
            else:
                h = ops.activation(h, ops.ACT_GELU_ERF)
        ctx.plan, ctx.owner, ctx.saved, ctx.tensors = plan, owner, saved, tensors
        return h

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        from .training import linear_bwd
        plan, owner, saved, tensors = ctx.plan, ctx.owner, ctx.saved, ctx.tensors
        need_x = ctx.needs_input_grad[0]
        need_t = ctx.needs_input_grad[3:]
        grads: List[Optional[torch.Tensor]] = [None] * len(tensors)
        # parameter index of every step
        first, i = [], 0
        for step in plan:
            first.append(i)
            i += (1 + (1 if step[2] else 0)) if step[0] == "linear" else (2 if step[0] == "ln" else 0)
        g = dy.to(saved[-1].dtype).contiguous()
        for li in reversed(range(len(plan))):
            step, h, p0 = plan[li], saved[li], first[li]
            need_dx = need_x or any(need_t[:p0])                      # anything upstream still wants a gradient?
            if step[0] == "linear":
                w = tensors[p0].detach().contiguous()
                if step[1]:                                           # fused GELU: the pre-activation is recomputed (one GEMM over sum L_i rows)
                    b = f32_of(owner, f"autograd_bias_{li}", tensors[p0 + 1]) if step[2] else None
                    pre = ops.linear(h, w, b)
                    g = ops.gelu_bwd(pre, g)
                need_dw = need_t[p0] or (step[2] and need_t[p0 + 1])
                gd: Dict[str, torch.Tensor] = {}
                g_in = linear_bwd(h, w, g, gd, "l", need_dx=need_dx, need_dw=need_dw)
                if need_dw:
                    grads[p0] = _grad_as(gd["l.weight"], tensors[p0])
                    if step[2]:
                        grads[p0 + 1] = _grad_as(gd["l.bias"], tensors[p0 + 1])
                g = g_in
            elif step[0] == "ln":
                C = h.shape[1]
                dgam = torch.empty((C,), dtype=torch.float32, device=h.device); dbet = torch.empty_like(dgam)
                g = ops.layernorm_bwd(h, g, f32_of(owner, f"autograd_lnw_{li}", tensors[p0]), step[1], dgam, dbet, accumulate=False, need_dx=need_dx)
                grads[p0], grads[p0 + 1] = _grad_as(dgam, tensors[p0]), _grad_as(dbet, tensors[p0 + 1])
            else:
                g = ops.gelu_bwd(h, g) if need_dx else None
            if g is None:
                break
        return (g if need_x else None, None, None) + tuple(grads)


def projector_plan(mods: Sequence[torch.nn.Module]):
    """(plan, parameter tensors) of an nn.Sequential-style projector for ProjectorFn."""
    import torch.nn as nn
    plan, tensors = [], []
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU)
            plan.append(("linear", fuse, m.bias is not None))
            tensors.append(m.weight)
            if m.bias is not None:
                tensors.append(m.bias)
            i += 2 if fuse else 1
        elif isinstance(m, nn.LayerNorm):
            plan.append(("ln", m.eps, True))
            tensors += [m.weight, m.bias]
            i += 1
        elif isinstance(m, nn.GELU):
            plan.append(("gelu", False, False))
            i += 1
        else:
            raise TypeError(f"unsupported projector module {type(m).__name__}")
    return tuple(plan), tensors


# ----------------------------------------------------------------------------------------------------------------------------
# the row copies of prepare_inputs_labels_for_multimodal (setokim_arch.py:266,284,290-303)
# ----------------------------------------------------------------------------------------------------------------------------
class SpliceRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed: Optional[torch.Tensor], embed_weight: torch.Tensor, src: torch.Tensor):
        ctx.src = src
        ctx.n_rows = 0 if packed is None else packed.shape[0]
        ctx.vocab = embed_weight.shape[0]
        return ops.splice_rows(src, embed_weight.detach().contiguous(), packed if (packed is not None and packed.shape[0]) else None)

    @staticmethod
    def backward(ctx, d_out: torch.Tensor):
        need_p, need_e = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_out = d_out.contiguous()
        dfe, dem = ops.splice_rows_bwd(ctx.src, d_out, ctx.n_rows if need_p else 0, ctx.vocab if need_e else 0)
        if need_p and dfe is None:
            dfe = d_out.new_zeros((0, d_out.shape[-1]))
        return dfe, (dem.to(d_out.dtype) if dem is not None else None), None


# ----------------------------------------------------------------------------------------------------------------------------
# the trainable head of the tokenizer (tokenizer.py:162-180; the tower is frozen and the clustering is no_grad, tokenizer.py:79)
# ----------------------------------------------------------------------------------------------------------------------------
class HeadFn(torch.autograd.Function):
    """tokens = head(hidden_rows) with the activations of training.head_forward_train kept for training.head_backward.  `aux` receives the
    non-differentiable results (counts, idx_cluster, score)."""

    @staticmethod
    def forward(ctx, tok, hidden_rows: torch.Tensor, B: int, k, threshold, noise, token_mask, dropout_seed, aux: dict, names: Tuple[str, ...],
                *params: torch.Tensor):
        from .training import head_forward_train
        tokens, saved = head_forward_train(tok, hidden_rows, B, k, threshold, noise, dropout_seed=dropout_seed, token_mask=token_mask)
        aux.update(counts=list(tokens.counts), idx=saved.pop("idx"), score=saved.pop("score"))
        ctx.tok, ctx.saved, ctx.names, ctx.params = tok, saved, names, params
        return tokens.packed

    @staticmethod
    def backward(ctx, dtokens: torch.Tensor):
        from .training import head_backward
        grads = head_backward(ctx.tok, ctx.saved, dtokens.contiguous())
        out = tuple(_grad_as(grads.get(n), p) for n, p in zip(ctx.names, ctx.params))
        return (None,) * 10 + out


def head_apply(tok, hidden_rows: torch.Tensor, B: int, k=None, threshold=None, noise=None, token_mask=None):
    """(packed tokens with a grad_fn, counts, idx_cluster (B, N), score (B, N)).  Dropout follows the module's mode like the reference's
    nn.Dropout (module.py:36,44,45,59,72): `tok.training` -> the proj_drop masks are active, seeded from torch's default CPU generator
    (`torch.manual_seed` makes a run repeatable); eval mode -> identity."""
    from .training import HEAD_MODULES
    named = [(n, p) for n, p in tok.named_parameters() if n.split(".")[0] in HEAD_MODULES]
    seed = None
    if tok.training and any(getattr(b, "proj_drop_p", 0.0) > 0.0 or getattr(b, "attn_drop_p", 0.0) > 0.0 for b in (tok.inner_encoder, tok.inter_encoder)):
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    aux: dict = {}
    packed = HeadFn.apply(tok, hidden_rows, B, k, threshold, noise, token_mask, seed, aux, tuple(n for n, _ in named), *[p for _, p in named])
    return packed, aux["counts"], aux["idx"], aux["score"]
