#!/usr/bin/env python
"""Micro-benchmark of the ViT attention kernel at the cfg2 shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
B, T, H = 256, 257, 16
if len(sys.argv) > 4:                                        # python tools/bench_attn.py 64 257 4096 1 : one head per "image" (rows of 3 * 64 elements)
    H = int(sys.argv[4])
if len(sys.argv) > 2:                                        # python tools/bench_attn.py 64 577 128 : the 336^2 tower (cfg4)
    T, B = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 128
Dh = int(sys.argv[1]) if len(sys.argv) > 1 else 64          # 64: the ViT-L tower; 48: the reconstruction decoder's ViT blocks
qkv = torch.randn(B * T, 3 * H * Dh, device="cuda").bfloat16()
out = torch.empty(B * T, H * Dh, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.attention(qkv, H, Dh, Dh ** -0.5, seg_len=T, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention(qkv, H, Dh, Dh ** -0.5, seg_len=T, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 4.0 * B * H * T * T * Dh
by = B * T * H * Dh * 2 * 4
print(f"attn_vit Dh={Dh} T={T} B={B}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s  {by/ms/1e6:.0f} GB/s (algorithmic q,k,v,o)")
