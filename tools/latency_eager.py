#!/usr/bin/env python
"""N eager single-image encode calls (a kernel-trace target): python tools/latency_eager.py [B] [calls]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
tok, proj = bench.build_model(dev, 224)
ctx = tok._context()
images = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(B)).to(device=dev, dtype=torch.bfloat16)
for _ in range(n):
    ctx.encode(images)
torch.cuda.synchronize()
