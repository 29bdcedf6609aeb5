#!/usr/bin/env python
"""Micro-benchmark of the causal attention kernel at the cfg5 shape (Vicuna-7B: 32 heads x 128, ~550-token sequences, batch 32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
B, T, H, Dh = 32, 550, 32, 128
qkv = torch.randn(B * T, 3 * H * Dh, device="cuda").bfloat16()
km = torch.ones(B * T, device="cuda", dtype=torch.uint8)
for _ in range(3):
    o = ops.attention_causal(qkv, km, B, T, H, Dh, Dh ** -0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    o = ops.attention_causal(qkv, km, B, T, H, Dh, Dh ** -0.5)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"attn_causal B={B} T={T}: {ms*1e3:.1f} us  {2.0 * B * H * T * T * Dh / ms / 1e9:.1f} TFLOP/s (causal half counted)")
