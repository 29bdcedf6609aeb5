#!/usr/bin/env python
"""Launch one GEMM shape a few times (for rocprofv3 --pmc runs): python tools/one_gemm.py M N K [act] [res]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
M, N, K = map(int, sys.argv[1:4])
act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
use_res = len(sys.argv) > 5 and sys.argv[5] == "1"
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").bfloat16() if use_res else None
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.linear(a, w, b, r, act=act, out=out)
torch.cuda.synchronize()
