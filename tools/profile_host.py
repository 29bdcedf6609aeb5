#!/usr/bin/env python
"""Host-side cost of one small-batch encode step (cProfile): python tools/profile_host.py [batch]"""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
import bench, setok_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
tok, proj = bench.build_model(dev, 224)
images = torch.randn(B, 3, 224, 224).to(device=dev, dtype=torch.bfloat16)
for _ in range(3):
    setok_amd.encode_images(tok, proj, images)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    setok_amd.encode_images(tok, proj, images)
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    setok_amd.encode_images(tok, proj, images)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
