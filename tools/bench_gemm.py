#!/usr/bin/env python
"""Micro-benchmark of setok_linear on the GEMM shapes of the cfg2 workload (run on the GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops

SHAPES = [("qkv", 65792, 3072, 1024, 0, False), ("proj", 65792, 1024, 1024, 0, True), ("fc1", 65792, 4096, 1024, 1, False),
          ("fc2", 65792, 1024, 4096, 0, True), ("sq4k", 4096, 4096, 4096, 0, False), ("sq8k", 8192, 8192, 8192, 0, False),
          ("inter", 9300, 1024, 1024, 0, True)]
res = {}
for name, M, N, K, act, use_res in SHAPES:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").bfloat16() if use_res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(a, w, b, r, act=act, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.linear(a, w, b, r, act=act, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    res[name] = dict(M=M, N=N, K=K, ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1))
    print(name, res[name], flush=True)
print(json.dumps(res))
