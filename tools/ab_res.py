#!/usr/bin/env python
"""The two residual GEMMs of a ViT-L layer (proj, fc2), sustained, on the library named by SETOK_HIP_LIB (same-box A/B of epilogue variants)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
M = 65792
for name, N, K in (("proj+residual", 1024, 1024), ("fc2+residual", 1024, 4096)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.zeros(N, device="cuda"); r = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for rnd in range(2):
        ops.linear(a, w, b, r, out=out); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.0:
            for _ in range(20):
                ops.linear(a, w, b, r, out=out)
            torch.cuda.synchronize(); n += 20
        dt = (time.perf_counter() - t0) / n
    print(f"{name:16s} {dt * 1e6:7.1f} us  {2.0 * M * N * K / dt / 1e12:7.1f} TFLOP/s", flush=True)
