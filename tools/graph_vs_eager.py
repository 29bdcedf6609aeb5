#!/usr/bin/env python
"""Does a HIP-graph replay of `setok_encode` shorten the idle time between its ~300 dependent dispatches at a LARGE batch?
python tools/graph_vs_eager.py [B] [reps]  — alternating blocks of eager calls and replays of the same call (frozen path, projector included),
ms per call, and whether the two give the same bits."""
import sys, time, torch
sys.path.insert(0, ".")
import bench, setok_amd
from setok_amd.context import GraphedEncode
from setok_amd.tokenizer import RaggedTokens
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
tok, proj = bench.build_model(dev, 224, torch.bfloat16, -1)
images = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(3)).to(device=dev, dtype=torch.bfloat16)
g = GraphedEncode(tok, B)
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
eager = lambda: setok_amd.encode_images(tok, proj, images)
def graphed():
    t, cl, _, _, _ = g(images)
    return proj(RaggedTokens(t, cl))
for _ in range(3): eager(); graphed()
for blk in range(4):
    print(f"block {blk}: eager {timed(eager):.3f} ms   graph replay {timed(graphed):.3f} ms", flush=True)
a, b = eager(), graphed()
print("bit-identical:", bool(torch.equal(a.packed, b.packed)), tuple(a.packed.shape))
