#!/usr/bin/env python
"""Where the vendor library stands on the same shapes: torch.nn.functional.linear (hipBLASLt / rocBLAS behind PyTorch-ROCm), bias included,
against setok_linear without activation / residual.  A reference point for DESIGN.md, not a code path of the package."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from setok_amd import ops


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, N, K in (("qkv", 65792, 3072, 1024), ("proj", 65792, 1024, 1024), ("fc1", 65792, 4096, 1024), ("fc2", 65792, 1024, 4096), ("sq8k", 8192, 8192, 8192)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    bb = b.bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for rep in range(2):
        t_v = timed(lambda: F.linear(a, w, bb))
        t_o = timed(lambda: ops.linear(a, w, b, out=out))
        fl = 2.0 * M * N * K
        print(f"{name:5s} vendor {t_v * 1e3:7.1f} us {fl / t_v / 1e9:7.1f} TF | setok_linear {t_o * 1e3:7.1f} us {fl / t_o / 1e9:7.1f} TF", flush=True)
