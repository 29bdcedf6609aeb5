#!/usr/bin/env python
"""LayerNorm micro-benchmark at the cfg2 shape (65792 x 1024 bf16): python tools/bench_ln.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
rows, C = 65792, 1024
x = torch.randn(rows, C, device="cuda").bfloat16(); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
y = torch.empty_like(x)
for _ in range(3): ops.layernorm(x, g, b, 1e-5, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.layernorm(x, g, b, 1e-5, out=y)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f"layernorm {rows} x {C}: {us:.1f} us  {rows * C * 4 / us / 1e6:.2f} TB/s (read + write)")
ref = torch.nn.functional.layer_norm(x.float(), (C,), g, b, 1e-5)
print("max err vs torch fp32:", float((y.float() - ref).abs().max()))
