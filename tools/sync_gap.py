#!/usr/bin/env python
"""Where the idle time between `setok_encode`'s last kernel and the projector's first goes (cfg2, frozen path): host-side intervals of one
encode_images call — the tokenizer call (enqueue everything, read the B token counts: the one synchronisation), then the projector's two launches.
python tools/sync_gap.py [B] [reps]"""
import sys, time, torch
sys.path.insert(0, ".")
import bench, setok_amd
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
tok, proj = bench.build_model(dev, 224, torch.bfloat16, -1)
images = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(3)).to(device=dev, dtype=torch.bfloat16)
for _ in range(3): setok_amd.encode_images(tok, proj, images)
torch.cuda.synchronize()
ta = tb = tc = 0.0
for _ in range(reps):
    t0 = time.perf_counter()
    feats, _, _ = tok(images)
    t1 = time.perf_counter()
    out = proj(feats)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    ta += t1 - t0; tb += t2 - t1; tc += t3 - t2
print(f"tokenizer call (enqueue + wait for the counts) {ta / reps * 1e3:.3f} ms; projector call (host: two launches) {tb / reps * 1e6:.0f} us; "
      f"the projector's kernels after that {tc / reps * 1e6:.0f} us")
# the same with the device idle: what the host needs to get through the tokenizer's ~220 launches
x = images[:1]
for _ in range(3): tok(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): tok(x)
torch.cuda.synchronize(); print(f"one image, whole call: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
