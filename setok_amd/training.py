"""Training step of the trainable SeTok head on the HIP library (BASELINE config 4).

What is trained: `inner_encoder`, `inter_encoder`, `out` (the 37.8 M head parameters).  The tower is frozen
(clip_encoder.py:50 with unfreeze_mm_vision_tower=False) and `cluster_dpc_knn` runs under no_grad (tokenizer.py:79), so the
backward pass starts at dL/dtokens (what the projector / LLM hands back) and ends at the head's parameter gradients — it never
needs dL/dfeatures.  The reference gets all of this from torch autograd; here it is written out:

    forward  (saving activations)   tokenizer.py:162-180, module.py:29-100
    backward                        out Linear <- inter_encoder Block <- segment mean <- inner_encoder Block
    gradient all-reduce             one bucketed sum over the data-parallel group (RCCL), issued per module as soon as that
                                    module's gradients exist, on a side stream, overlapping the rest of the backward pass
    AdamW                           fp32 master weights, low-precision copies for the next forward

Every GEMM (forward, dX = dY W, dW = dY^T X) is a `setok_linear` call; the rest is csrc/backward.hip.  No autograd, no fallback.
"""
from __future__ import annotations

import os

import math
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import ops
from .tokenizer import Block, RaggedTokens, SetokTokenizer

HEAD_MODULES = ("out", "inter_encoder", "inner_encoder")          # the order their gradients become available in


def _k_granule(dtype) -> int:
    return 64 if dtype in (torch.bfloat16, torch.float16) else 16                  # contraction granule of setok_linear


# ----------------------------------------------------------------------------------------------------------------------------
# Linear
# ----------------------------------------------------------------------------------------------------------------------------
def linear_bwd(x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, grads: Dict[str, torch.Tensor], name: str, need_dx: bool = True,
               residual: Optional[torch.Tensor] = None, need_dw: bool = True, alloc=None) -> Optional[torch.Tensor]:
    """y = x w^T + b.  grads[name.weight] = dy^T x (fp32), grads[name.bias] = colsum(dy); returns dx = dy w (+ residual).
    `alloc(parameter name, shape)`: where a gradient is to be written (a view of a flat all-reduce bucket, parallel.GradBuckets); None = fresh tensors."""
    pad = _k_granule(x.dtype)
    M, N, K = x.shape[0], dy.shape[1], x.shape[1]
    if need_dw:
        # the contraction runs over the M rows; a weight gradient has few output tiles (N x K), so long contractions are split into S
        # batched partial products summed in a fixed order — enough workgroups to fill the chip, still deterministic
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        S = max(1, min(16, 1024 // max(tiles, 1), M // 2048))
        wo = alloc(name + ".weight", (N, K)) if alloc else None
        bo = alloc(name + ".bias", (N,)) if alloc else None
        (dyT, db), xT = ops.transpose(dy, pad, S, with_colsum=True, colsum_out=bo), ops.transpose(x, pad, S)      # the bias gradient falls out of dY's transpose
        grads[name + ".weight"] = ops.linear_tn(dyT, xT, out=wo)
        grads[name + ".bias"] = db
    if not need_dx:
        return None
    wT = ops.transpose(w.contiguous())                            # (K, N)
    return ops.linear(dy, wT, residual=residual)


# ----------------------------------------------------------------------------------------------------------------------------
# Block (module.py:76-100): `depth` attention sub-layers sharing ONE norm1, then norm2 + Mlp
# ----------------------------------------------------------------------------------------------------------------------------
_FUSED_DROPOUT = os.environ.get("SETOK_TRAIN_FUSED_DROPOUT", "1") != "0"     # A/B switch: the two-launch forms of the Mlp's middle dropout (identical bits)
SITE_STRIDE = 1 << 40          # elements a dropout site may hold: every site draws from its own range of the counter


class DropSpec:
    """Training-mode dropout of a Block (module.py:36,44,45,59,72): rate p = the module's proj_drop, `seed` of the step, `site0` = the first of
    the depth + 2 counter ranges this Block owns (attention projection of layer i: site0 + i; Mlp activation: site0 + depth; Mlp fc2:
    site0 + depth + 1)."""

    def __init__(self, p: float, seed: int, site0: int):
        self.p, self.seed, self.site0 = float(p), int(seed), int(site0)

    def offset(self, site: int) -> int:
        return (self.site0 + site) * SITE_STRIDE


def block_forward_train(blk: Block, x: torch.Tensor, seg_offsets: torch.Tensor, n_segs: int, seg_bound: int, drop: Optional[DropSpec] = None):
    pk = blk._pack()
    H, Dh = blk.num_heads, blk.dim // blk.num_heads
    depth = len(pk["attn"])
    ctx = dict(x=[], y=[], qkv=[], o=[], seg=(seg_offsets, n_segs, seg_bound), drop=drop)
    for i, a in enumerate(pk["attn"]):
        y = ops.layernorm(x, *pk["n1"], pk["eps"])
        qkv = ops.linear(y, a["wqkv"], a["bqkv"])
        o = ops.attention(qkv, H, Dh, a["scale"], seg_len=seg_bound, seg_offsets=seg_offsets, n_segs=n_segs)
        ctx["x"].append(x); ctx["y"].append(y); ctx["qkv"].append(qkv); ctx["o"].append(o)
        if drop is None:
            x = ops.linear(o, a["wproj"], a["bproj"], residual=x)
        else:                                                        # x + proj_drop(proj(o))  (module.py:71-72,96)
            pr = ops.linear(o, a["wproj"], a["bproj"])
            x = ops.dropout(pr, drop.p, drop.seed, drop.offset(i), residual=x, out=pr)
    y2 = ops.layernorm(x, *pk["n2"], pk["eps"])
    pre = ops.linear(y2, pk["w1"], pk["b1"])
    if drop is None:
        u = ops.activation(pre, ops.ACT_GELU_ERF)                 # separate from the GEMM here: the backward pass needs `pre`
        out = ops.linear(u, pk["w2"], pk["b2"], residual=x)
    else:                                                            # x + drop(fc2(drop(act(fc1)))))  (module.py:40-45,98)
        if _FUSED_DROPOUT:
            u = ops.activation_dropout(pre, ops.ACT_GELU_ERF, drop.p, drop.seed, drop.offset(depth))  # one pass (round 4): = dropout(activation(pre))
        else:
            u = ops.activation(pre, ops.ACT_GELU_ERF)
            u = ops.dropout(u, drop.p, drop.seed, drop.offset(depth), out=u)
        f = ops.linear(u, pk["w2"], pk["b2"])
        out = ops.dropout(f, drop.p, drop.seed, drop.offset(depth + 1), residual=x, out=f)
    ctx.update(xd=x, y2=y2, pre=pre, u=u)                        # u: what fc2 read (after its dropout)
    return out, ctx


def block_backward(blk: Block, prefix: str, ctx, g: torch.Tensor, grads: Dict[str, torch.Tensor], need_dx: bool, alloc=None) -> Optional[torch.Tensor]:
    """g = dL/d(block output).  Fills grads[prefix + <reference parameter name>]; returns dL/d(block input) if need_dx."""
    pk = blk._pack()
    H, Dh = blk.num_heads, blk.dim // blk.num_heads
    C = blk.dim
    seg_offsets, n_segs, seg_bound = ctx["seg"]
    drop: Optional[DropSpec] = ctx.get("drop")
    dev = g.device
    depth = len(pk["attn"])
    # Mlp: out = xd + drop(fc2(drop(gelu(fc1(norm2(xd))))))  — a dropout's backward is the same mask and scale on the gradient
    gf = g if drop is None else ops.dropout(g, drop.p, drop.seed, drop.offset(depth + 1))
    du = linear_bwd(ctx["u"], pk["w2"], gf, grads, prefix + "mlp.fc2", alloc=alloc)
    if drop is not None:
        if _FUSED_DROPOUT:
            dpre = ops.gelu_bwd_dropout(ctx["pre"], du, drop.p, drop.seed, drop.offset(depth), out=du)   # one pass (round 4): = gelu_bwd(pre, dropout(du))
        else:
            dpre = ops.gelu_bwd(ctx["pre"], ops.dropout(du, drop.p, drop.seed, drop.offset(depth), out=du))
    else:
        dpre = ops.gelu_bwd(ctx["pre"], du)
    dy2 = linear_bwd(ctx["y2"], pk["w1"], dpre, grads, prefix + "mlp.fc1", alloc=alloc)
    if alloc:
        g2w, g2b = alloc(prefix + "norm2.weight", (C,)), alloc(prefix + "norm2.bias", (C,))
    else:
        g2w = torch.empty((C,), dtype=torch.float32, device=dev); g2b = torch.empty_like(g2w)
    g = ops.layernorm_bwd(ctx["xd"], dy2, pk["n2"][0], pk["eps"], g2w, g2b, accumulate=False, res=g)      # dL/dxd = g + LN2'(dy2)
    grads[prefix + "norm2.weight"], grads[prefix + "norm2.bias"] = g2w, g2b
    if alloc:                                                                                              # shared norm1 accumulates
        g1w, g1b = alloc(prefix + "norm1.weight", (C,)).zero_(), alloc(prefix + "norm1.bias", (C,)).zero_()
    else:
        g1w = torch.zeros((C,), dtype=torch.float32, device=dev); g1b = torch.zeros_like(g1w)
    for i in reversed(range(depth)):
        a = pk["attn"][i]
        lp = prefix + f"layers.{i}.1."
        gp = g if drop is None else ops.dropout(g, drop.p, drop.seed, drop.offset(i))
        do = linear_bwd(ctx["o"][i], a["wproj"], gp, grads, lp + "proj", alloc=alloc)
        dqkv = ops.attention_bwd(ctx["qkv"][i], ctx["o"][i], do, H, Dh, a["scale"], seg_bound, seg_offsets, n_segs)
        dy = linear_bwd(ctx["y"][i], a["wqkv"], dqkv, grads, lp + "qkv", alloc=alloc)
        last = i == 0 and not need_dx
        g = ops.layernorm_bwd(ctx["x"][i], dy, pk["n1"][0], pk["eps"], g1w, g1b, accumulate=True, need_dx=not last, res=g)
    grads[prefix + "norm1.weight"], grads[prefix + "norm1.bias"] = g1w, g1b
    return g


# ----------------------------------------------------------------------------------------------------------------------------
# the head
# ----------------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def head_forward_train(tok: SetokTokenizer, hidden_rows: torch.Tensor, B: int, k=None, threshold=None, noise=None, dropout_seed: Optional[int] = None,
                       token_mask=None):
    """tokenizer.py:162-180 for a batch, keeping what the backward pass needs.  Returns (tokens: RaggedTokens, ctx).  `dropout_seed`: run the two
    Blocks in TRAINING mode — nn.Dropout(proj_drop) at its three sites (module.py:36,44,45,59,72) with masks drawn from this seed; None = eval-mode
    arithmetic."""
    tower = tok.image_feature_encoder
    skip = 1 if tower.select_feature == "patch" else 0
    C = hidden_rows.shape[-1]
    N = hidden_rows.shape[0] // B - skip
    h = w = int(math.sqrt(N))
    pos = tok.position_embedding.table(h, w, hidden_rows.dtype, hidden_rows.device)
    x = ops.select_add_pos(hidden_rows, pos, B, N, skip)
    idx, score, index_down, counts = ops.cluster_dpc_knn(x, B, N, k if k else tok.min_cluster_num,
                                                         threshold if threshold else tok.threshold, tok.min_cluster_num, noise, token_mask)
    perm, seg_offsets, img_offsets = ops.cluster_sort(idx, counts)
    counts_h = counts.cpu().tolist()
    total = int(sum(counts_h))
    hs = ops.gather_rows(x, perm)
    d_inner = d_inter = None
    if dropout_seed is not None:
        for b in (tok.inner_encoder, tok.inter_encoder):
            if getattr(b, "attn_drop_p", 0.0) > 0.0:
                raise NotImplementedError("training-mode attn_drop (dropout on the attention probabilities, module.py:68) is not implemented; the reference's default is 0.0")
        n_inner = len(tok.inner_encoder._pack()["attn"]) + 2
        if tok.inner_encoder.proj_drop_p > 0.0:
            d_inner = DropSpec(tok.inner_encoder.proj_drop_p, dropout_seed, 0)
        if tok.inter_encoder.proj_drop_p > 0.0:
            d_inter = DropSpec(tok.inter_encoder.proj_drop_p, dropout_seed, n_inner)
    inner_out, inner_ctx = block_forward_train(tok.inner_encoder, hs, seg_offsets, total, N, d_inner)
    group = ops.segment_mean(inner_out, seg_offsets, img_offsets[B:], total)
    inter_out, inter_ctx = block_forward_train(tok.inter_encoder, group, img_offsets, B, max(counts_h), d_inter)
    w_out = tok.out.weight.detach().contiguous()
    tokens = ops.linear(inter_out, w_out, tok.out.bias.detach().float().contiguous())
    ctx = dict(inner=inner_ctx, inter=inter_ctx, inter_out=inter_out, w_out=w_out, seg_offsets=seg_offsets, img_offsets=img_offsets,
               total=total, rows=B * N, B=B, idx=idx, score=score)
    return RaggedTokens(tokens, counts_h), ctx


@torch.no_grad()
def head_backward(tok: SetokTokenizer, ctx, dtokens: torch.Tensor, on_module_done: Optional[Callable[[str, Dict[str, torch.Tensor]], None]] = None,
                  alloc=None) -> Dict[str, torch.Tensor]:
    """dtokens: (sum L_i, token_feat_dim) = dL/dtokens in the packed order of the forward's RaggedTokens.  Returns fp32 gradients
    under the reference's parameter names.  `on_module_done(module, grads_of_that_module)` fires as soon as a module's gradients
    are complete (out, then inter_encoder, then inner_encoder) — the hook the overlapped all-reduce hangs on."""
    grads: Dict[str, torch.Tensor] = {}

    def done(mod):
        if on_module_done is not None:
            on_module_done(mod, {n: g for n, g in grads.items() if n.startswith(mod + ".")})

    dtokens = dtokens.to(ctx["inter_out"].dtype).contiguous()
    g = linear_bwd(ctx["inter_out"], ctx["w_out"], dtokens, grads, "out", alloc=alloc)
    done("out")
    g = block_backward(tok.inter_encoder, "inter_encoder.", ctx["inter"], g, grads, need_dx=True, alloc=alloc)
    done("inter_encoder")
    g = ops.segment_mean_bwd(g, ctx["seg_offsets"], ctx["img_offsets"][ctx["B"]:], ctx["total"], ctx["rows"])
    block_backward(tok.inner_encoder, "inner_encoder.", ctx["inner"], g, grads, need_dx=False, alloc=alloc)
    done("inner_encoder")
    return grads


# ----------------------------------------------------------------------------------------------------------------------------
# optimizer + data-parallel step
# ----------------------------------------------------------------------------------------------------------------------------
class HeadTrainer:
    """AdamW over the head's parameters with fp32 master weights, and the data-parallel training step:

        tokens, ctx = trainer.forward(images)             # tower (frozen) + head, activations kept
        trainer.backward(ctx, dtokens)                    # gradients; per-module all-reduce overlapped on a side stream
        trainer.step()                                    # waits for the all-reduce, AdamW, refreshes the low-precision weights
    """

    def __init__(self, tok: SetokTokenizer, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, process_group=None, bucket_bytes: int = 64 << 20, dropout: str = "warn", dropout_seed: int = 0):
        """`dropout`: the reference trains the head with Attention.proj_drop and both Mlp.drop at `proj_drop` (0.2 by default,
        tokenizer.py:26, module.py:36,44,45,59,72).
          "train": the masks are active in forward() and backward() (setok_dropout: counter-based, seeded by `dropout_seed`, the step number and
                   the data-parallel rank; attn_drop > 0 is not implemented and raises) — the reference's training objective;
          "eval":  eval-mode arithmetic, the unregularised objective, accepted silently (benchmarks, parity tests against eval-mode gradients);
          "warn" (default) / "error": eval-mode arithmetic, but say so once / refuse when the module was built with a non-zero rate."""
        rates = {n: (getattr(b, "proj_drop_p", 0.0), getattr(b, "attn_drop_p", 0.0)) for n, b in (("inner_encoder", tok.inner_encoder), ("inter_encoder", tok.inter_encoder))}
        if dropout not in ("warn", "error", "eval", "train"):
            raise ValueError(f"dropout must be 'train', 'warn', 'error' or 'eval', got {dropout!r}")
        if dropout == "train" and any(a > 0.0 for _, a in rates.values()):
            raise NotImplementedError(f"HeadTrainer(dropout='train'): attn_drop > 0 (dropout on the attention probabilities, module.py:68) is not implemented: {rates}")
        if any(p > 0.0 or a > 0.0 for p, a in rates.values()) and dropout not in ("eval", "train"):
            msg = (f"HeadTrainer runs the head without dropout (eval-mode arithmetic), but the module was built with (proj_drop, attn_drop) = {rates}: "
                   "the reference trains with these masks active (module.py:29-73); pass dropout='train' for them, or dropout='eval' to accept the unregularised objective")
            if dropout == "error":
                raise NotImplementedError(msg)
            import warnings
            warnings.warn(msg, stacklevel=2)
        self.dropout = dropout
        self.dropout_seed = int(dropout_seed)
        self.tok = tok
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.group = process_group
        self.bucket_bytes = bucket_bytes
        self.t = 0
        self.params: Dict[str, torch.nn.Parameter] = {n: p for n, p in tok.named_parameters() if n.split(".")[0] in HEAD_MODULES}
        self.master = {n: p.detach().float().clone() for n, p in self.params.items()}
        self.m = {n: torch.zeros_like(v) for n, v in self.master.items()}
        self.v = {n: torch.zeros_like(v) for n, v in self.master.items()}
        self.grads: Dict[str, torch.Tensor] = {}
        # one flat fp32 bucket (or a few, <= bucket_bytes each) per module, a view per parameter: the backward pass writes the gradients in
        # place and a module's all-reduce runs on the bucket itself (scripts/zero2.json:16-22: contiguous_gradients + reduce_bucket_size)
        from .parallel import GradBuckets
        dev = next(iter(self.params.values())).device
        self.buckets = GradBuckets({m: [(n, tuple(p.shape)) for n, p in self.params.items() if n.split(".")[0] == m] for m in HEAD_MODULES},
                                   bucket_bytes=bucket_bytes, device=dev)
        self._comm_stream = None
        self._pending: List = []
        self._comm_bytes = 0                                            # gradient bytes handed to the all-reduce in the current step
        self._wait_events: List = []                                    # (before, after) events around step()'s wait for the all-reduce

    # -- data-parallel plumbing --------------------------------------------------------------------------------------------
    @property
    def world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _allreduce_module(self, mod: str, g: Dict[str, torch.Tensor]) -> None:
        """Sum-all-reduce one module's gradients on the communication stream (RCCL over xGMI: point-to-point links, a ring is per-link bound,
        so few large buckets), while the compute stream carries on with the next module's backward.  When the gradients live in this
        trainer's flat buckets (the normal case: head_backward wrote them through `self.buckets.view`) the collective runs on the bucket
        itself — zero-copy; gradients handed in as free tensors take the concatenating path of parallel.allreduce_gradients."""
        if not g:
            return
        names = sorted(g)
        self._comm_bytes += sum(g[n].numel() * g[n].element_size() for n in names)      # what a data-parallel step reduces (counted at any world size)
        if self.world == 1:
            return
        from .parallel import allreduce_gradients
        bk = getattr(self, "buckets", None)
        in_buckets = bk is not None and all(n in bk and g[n].data_ptr() == bk.view(n).data_ptr() for n in names)

        def reduce_():
            if in_buckets:
                bk.allreduce(mod, group=self.group, average=False)
            else:
                allreduce_gradients([g[n] for n in names], group=self.group, bucket_bytes=self.bucket_bytes, average=False)
        if g[names[0]].is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                reduce_()
                ev = torch.cuda.Event(); ev.record(self._comm_stream)
            self._pending.append(ev)
        else:
            reduce_()

    # -- the step -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, images: torch.Tensor, k=None, threshold=None, noise=None):
        B = images.shape[0]
        hidden = self.tok.image_feature_encoder.hidden_rows(images)
        if hidden.dtype != self.tok.dtype:
            hidden = hidden.to(self.tok.dtype)
        return head_forward_train(self.tok, hidden, B, k, threshold, noise, dropout_seed=self.step_seed() if self.dropout == "train" else None)

    def step_seed(self) -> int:
        """The dropout seed of the step about to run: a function of (dropout_seed, step number, data-parallel rank) — every rank and every step
        draws different masks, and a resumed run (self.t restored) repeats them."""
        import torch.distributed as dist
        rank = dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0
        z = (self.dropout_seed * 0x9E3779B97F4A7C15 + self.t * 0xBF58476D1CE4E5B9 + rank * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 29
        return (z * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF

    @torch.no_grad()
    def backward(self, ctx, dtokens: torch.Tensor) -> Dict[str, torch.Tensor]:
        for ev in self._pending:                                      # a backward without a step() in between: its all-reduce still reads the buckets
            torch.cuda.current_stream().wait_event(ev)
        self._pending = []
        self._comm_bytes = 0
        self.grads = head_backward(self.tok, ctx, dtokens, on_module_done=self._allreduce_module, alloc=self.buckets.view)
        return self.grads

    @torch.no_grad()
    def step(self) -> None:
        if self._pending:                                             # exposed all-reduce time = how long the compute stream stalls here
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for ev in self._pending:
                torch.cuda.current_stream().wait_event(ev)
            e1.record()
            self._wait_events = (self._wait_events + [(e0, e1)])[-64:]
        self._pending = []
        self.t += 1
        scale = 1.0 / self.world                                      # mean over the data-parallel ranks
        for n, p in self.params.items():
            ops.adamw(self.master[n].view(-1), self.grads[n].reshape(-1), self.m[n].view(-1), self.v[n].view(-1), p.data.view(-1), self.lr,
                      self.betas[0], self.betas[1], self.eps, self.wd, self.t, scale)
        # The parameters were rewritten through raw pointers (`p.data.view(-1)`), which does NOT bump their `_version` counters — every cache
        # keyed on those would keep serving the pre-step weights: the library-side encode context (device COPIES of all weights), the Blocks'
        # packed fp32 biases / LayerNorm affines, the cached fp32 copies of bf16 biases.  Drop them all.
        for p in self.params.values():
            torch.autograd.graph.increment_version(p)                 # what an in-place torch op would have done: every version-keyed cache notices
        self.tok.__dict__["_ctx"] = None
        for top in HEAD_MODULES:                                      # (not the frozen tower: its fused / folded copies stay valid)
            for m in getattr(self.tok, top).modules():
                m.__dict__.pop("_f32_cache", None)
                if hasattr(m, "_drop_pack"):
                    m._drop_pack()

    def comm_stats(self) -> Dict[str, float]:
        """Of the last step(s): gradient bytes all-reduced per step and the exposed (not overlapped with the backward pass) all-reduce time —
        the compute stream's stall at the top of step().  Synchronises."""
        ms = 0.0
        if self._wait_events:
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in self._wait_events) / len(self._wait_events)
        return dict(allreduce_bytes_per_step=int(self._comm_bytes), exposed_allreduce_ms=round(ms, 4), world=self.world)
