#!/usr/bin/env python
"""Steady-state GEMM rates (each shape looped ~2 s so the power controller settles; tools/bench_gemm.py's 10-launch bursts read higher):
python tools/bench_gemm_steady.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
SHAPES = [("qkv", 65792, 3072, 1024, 0, False), ("proj+res", 65792, 1024, 1024, 0, True), ("fc1+qgelu", 65792, 4096, 1024, 1, False),
          ("fc2+res", 65792, 1024, 4096, 0, True), ("sq8k", 8192, 8192, 8192, 0, False)]
for name, M, N, K, act, res in SHAPES:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = time.time(); ms = 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.time() - t0 < 2.0:
        e0.record()
        for _ in range(50):
            ops.linear(a, w, b, r, act=act, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
    print(f"{name:10s} {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF", flush=True)
