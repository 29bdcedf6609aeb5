#!/usr/bin/env python
"""Would two half-batches of the ViT tower on two streams hide the idle time between dependent dispatches (each stream's gap filled by the other stream's
kernel)?  python tools/two_stream_tower.py [B] [reps]: the tower on B images on one stream against B/2 + B/2 on two streams, ms per pass."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
tok, _ = bench.build_model(dev, 224, torch.bfloat16, -1)
tower = tok.image_feature_encoder
images = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(3)).to(device=dev, dtype=torch.bfloat16)
halves = [images[: B // 2].contiguous(), images[B // 2:].contiguous()]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
def one():
    return tower.hidden_rows(images)
def two():
    cur = torch.cuda.current_stream(dev)
    outs = []
    for st, im in zip(streams, halves):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(tower.hidden_rows(im))
    for st in streams: cur.wait_stream(st)
    return outs
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for _ in range(3): one(); two()
for blk in range(3):
    print(f"block {blk}: one stream {timed(one):.3f} ms   two streams {timed(two):.3f} ms", flush=True)
a = one(); b = torch.cat(two(), 0)
print("bit-identical:", bool(torch.equal(a, b)))
