#!/usr/bin/env python
"""Small-batch latency of the whole encode call (the reference's per-sample dataset-side calls, pairDataset.py:419-421):
python tools/latency.py [batches ...]   — eager single call vs the graph-replayed form, ms per call (median of 50), bit-equality of the two."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
import bench
from setok_amd.context import GraphedEncode
dev = torch.device("cuda", 0)
tok, proj = bench.build_model(dev, 224)
ctx = tok._context()
for B in [int(a) for a in sys.argv[1:]] or [1, 8]:
    images = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(B)).to(device=dev, dtype=torch.bfloat16)
    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]
    e_med, e_min = timed(lambda: ctx.encode(images))
    g = GraphedEncode(ctx, B)
    g_med, g_min = timed(lambda: g(images))
    a, b = ctx.encode(images), g(images)
    same = torch.equal(a[0], b[0]) and a[1] == b[1] and torch.equal(a[2], b[2])
    floor_ms = 0.68e9 / 6.3e12 * 1e3                       # 0.68 GB of bf16 weights at the measured 6.3 TB/s copy rate
    print(f"B={B}: eager {e_med:.3f} ms (min {e_min:.3f}), graph replay {g_med:.3f} ms (min {g_min:.3f}), identical results: {same}, "
          f"weight-streaming floor {floor_ms:.3f} ms -> {floor_ms / g_min:.3f} of it", flush=True)
