#!/usr/bin/env python
"""Small-M GEMM latencies (a handful of images): python tools/bench_small_gemm.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
for M in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (257, 1028, 2056, 4112, 6168)):
    for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("out", 4096, 1024), ("dec", 768, 768), ("llm", 4096, 4096)):
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(3): ops.linear(a, w, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.linear(a, w, b, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"M={M:5d} {name:4s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
