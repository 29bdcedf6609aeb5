#!/usr/bin/env python
"""The library's bf16 GEMM WITHOUT bias / activation / residual on random operands, the shapes tools/micro/gemm_pp runs (same-box comparison)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
for name, M, N, K in [("sq8k", 8192, 8192, 8192), ("sq4k", 4096, 4096, 4096), ("qkv", 65792, 3072, 1024), ("fc1", 65792, 4096, 1024), ("fc2", 65792, 1024, 4096)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 1.7 * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = time.time(); ms = 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.time() - t0 < 1.5:
        e0.record()
        for _ in range(20):
            ops.linear(a, w, None, None, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
    print(f"library plain {name:5s} {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF", flush=True)
