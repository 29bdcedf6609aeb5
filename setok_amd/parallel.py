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


class GradBuckets:
    """Pre-allocated flat fp32 gradient buckets with one view per parameter — the layout DeepSpeed ZeRO-2's `reduce_bucket_size` /
    `contiguous_gradients` give the reference (scripts/zero2.json:16-22).  The hand-written backward writes every gradient straight into its
    view (`view(name)`), so a module's all-reduce is a collective on the bucket itself: no concatenation before and no copy back after it
    (round 2 did both: two extra passes over 151 MB per step).  `modules`: {module name: [(parameter name, shape), ...]} in the order the
    backward pass completes them; a module larger than `bucket_bytes` is split at parameter boundaries (xGMI is point-to-point and a ring is
    per-link bound: few large buckets)."""

    def __init__(self, modules, bucket_bytes: int = 64 << 20, device=None):
        self.flat = {}                 # module -> [flat fp32 tensors]
        self._views = {}               # parameter name -> view
        self.order = []                # (module, bucket index) in creation order == reduction order
        for mod, plist in modules.items():
            groups, cur, size = [], [], 0
            for name, shape in plist:
                n = 1
                for d in shape:
                    n *= int(d)
                if cur and (size + n) * 4 > bucket_bytes:
                    groups.append(cur); cur, size = [], 0
                cur.append((name, tuple(shape), n)); size += n
            if cur:
                groups.append(cur)
            self.flat[mod] = []
            for gi, grp in enumerate(groups):
                pad = [(n + 63) // 64 * 64 for _, _, n in grp]                      # every view starts on a 256-byte boundary (16-byte vector stores)
                buf = torch.zeros((sum(pad),), dtype=torch.float32, device=device)
                off = 0
                for (name, shape, n), pn in zip(grp, pad):
                    self._views[name] = buf[off: off + n].view(shape)
                    off += pn
                self.flat[mod].append(buf)
                self.order.append((mod, gi))

    def view(self, name: str, shape=None) -> torch.Tensor:
        v = self._views[name]
        if shape is not None and tuple(v.shape) != tuple(shape):
            raise ValueError(f"gradient bucket view {name}: shape {tuple(v.shape)}, asked for {tuple(shape)}")
        return v

    def __contains__(self, name: str) -> bool:
        return name in self._views

    def names(self):
        return list(self._views)

    def nbytes(self, mod: str) -> int:
        return sum(b.numel() * 4 for b in self.flat[mod])

    def allreduce(self, mod: str, group=None, average: bool = False) -> int:
        """Sum (or mean) every bucket of `mod` over the data-parallel group IN PLACE; returns the number of collectives."""
        world = dist.get_world_size(group)
        for buf in self.flat[mod]:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            if average:
                buf.div_(world)
        return len(self.flat[mod])


def pin_host_to_gpu_numa_node(local_gpu: int) -> dict:
    """One process per GPU: keep this rank's host threads (the launcher of ~200 kernels per step, the RCCL proxy thread) on the CPU socket its
    GPU hangs off — on an 8-GPU node eight unpinned ranks migrate between sockets and launch across the inter-socket link.  Best effort and
    silent: reads the GPU's PCI address from HIP, its `numa_node` and that node's `cpulist` from sysfs, and restricts the process to the
    intersection with the CPUs it is already allowed (a launcher's own pinning is respected).  Returns what it did for the bench record."""
    import os
    info = {"numa_node": None, "cpus": None}
    try:
        bus = torch.cuda.get_device_properties(local_gpu).pci_bus_id
        dev_id = torch.cuda.get_device_properties(local_gpu).pci_device_id
        dom = getattr(torch.cuda.get_device_properties(local_gpu), "pci_domain_id", 0)
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev_id:02x}.0/numa_node"
        with open(path) as f:
            node = int(f.read().strip())
        info["numa_node"] = node
        if node < 0:
            return info
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if allowed:
            os.sched_setaffinity(0, allowed)
            info["cpus"] = len(allowed)
    except Exception:                                               # no sysfs, a container without the node files, a non-Linux host: leave the scheduler alone
        pass
    return info


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """The benchmark's timing rule: wall time of the slowest rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
