"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The encode path shards embarrassingly — every image is an independent unit through the whole path
 — so there is NO collective on the data path: each rank encodes its contiguous slice of
the batch with replicated weights.  Collectives appear only (a) when a caller wants the ragged result
on every rank (`gather_ragged`), (b) in the training-time gradient all-reduce of the trainable head
(DeepSpeed ZeRO-2's job in the reference, scripts/zero2.json:16-22; `allreduce_gradients` here), and
(c) in the benchmark's barrier / MAX-over-ranks timing.  Everything in this file is backend-agnostic
and is covered by world_size-2 `gloo` tests on CPU (tests/test_parallel_cpu.py)."""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n units: the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(images: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    s, e = shard_range(images.shape[0], rank, world)
    return images[s:e]


def gather_ragged(packed: torch.Tensor, counts: Sequence[int], group=None) -> Tuple[torch.Tensor, List[int]]:
    """All-gather a ragged result (packed (sum L_i, D) + per-image counts) in rank order.  Two collectives:
    the counts (as an object list) and the token rows padded to the largest per-rank total."""
    world = dist.get_world_size(group)
    all_counts: List[List[int]] = [None] * world
    dist.all_gather_object(all_counts, [int(c) for c in counts], group=group)
    totals = [sum(c) for c in all_counts]
    pad = max(totals)
    buf = packed.new_zeros((pad, packed.shape[1]))
    buf[: packed.shape[0]] = packed
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    rows = torch.cat([o[:t] for o, t in zip(out, totals)], dim=0)
    return rows, [c for cs in all_counts for c in cs]


def allreduce_gradients(params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, group=None, average: bool = True) -> int:
    """Bucketed gradient all-reduce for the trainable head (37.8 M parameters at ViT-L dims -> 151 MB fp32).
    xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is per-link bound: a few
    large buckets (default 64 MiB) amortise launch latency without delaying the first bucket.  Returns the
    number of collectives issued."""
    world = dist.get_world_size(group)
    # nn.Parameters contribute their .grad; plain tensors (the hand-written backward's fp32 gradient buffers) are reduced themselves
    grads = [(p.grad if isinstance(p, torch.nn.Parameter) else p) for p in params]
    grads = [g for g in grads if g is not None]
    n_coll, bucket, size = 0, [], 0

    def flush():
        nonlocal n_coll, bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off: off + g.numel()].view_as(g)); off += g.numel()
        n_coll += 1
        bucket, size = [], 0

    for g in grads:
        if bucket and (size + g.numel() * g.element_size() > bucket_bytes or g.dtype != bucket[0].dtype):
            flush()
        bucket.append(g); size += g.numel() * g.element_size()
    flush()
    return n_coll


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """The benchmark's timing rule: wall time of the slowest rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
