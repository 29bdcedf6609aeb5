#!/usr/bin/env python
"""Micro-benchmark of setok_cluster_dpc_knn (run on the GPU box): python tools/bench_cluster.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops

for dt, B, N, C, k, fused in [(torch.bfloat16, 256, 256, 1024, 64, "1"), (torch.bfloat16, 256, 256, 1024, 64, "0"), (torch.bfloat16, 1, 256, 1024, 64, "1"),
                              (torch.bfloat16, 128, 576, 1024, 64, "1"), (torch.bfloat16, 128, 576, 1024, 64, "0"), (torch.bfloat16, 16, 576, 1024, 64, "1"),
                              (torch.bfloat16, 1, 576, 1024, 64, "1"), (torch.float32, 256, 256, 1024, 64, "1")]:
    os.environ["SETOK_CLUSTER_FUSED"] = fused          # "0": the multi-kernel form (distance matrix through memory)
    x = torch.randn(B * N, C, device="cuda").to(dt)
    for _ in range(3):
        ops.cluster_dpc_knn(x, B, N, k, 0.125, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.cluster_dpc_knn(x, B, N, k, 0.125, 64)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    alg = B * (N * C * x.element_size() + N * 20)
    print(f"{dt} B={B} N={N} C={C} fused={fused}: {ms * 1e3:.1f} us/call  algorithmic {alg / 1e6:.1f} MB -> {alg / ms / 1e6:.0f} GB/s", flush=True)
