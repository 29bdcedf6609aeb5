import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
torch.manual_seed(0)
def check(M, N, K, res, act=0):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.linear(a, w, b, r, act=act, out=out)
    torch.cuda.synchronize()
    ref = torch.nn.functional.linear(a.float(), w.float(), b)
    if act == 2: ref = torch.nn.functional.gelu(ref)
    ref = ref.bfloat16().float()
    if res: ref = ref + r.float()
    o = out.float()
    bad = ~torch.isfinite(o)
    err = (o - ref).abs().max().item() if not bad.any() else float("nan")
    rows = bad.any(1).nonzero().flatten()
    print(f"M={M} N={N} K={K} res={res} act={act}: nonfinite={int(bad.sum())} rows[{rows[:4].tolist()}..{rows[-4:].tolist()}] n_rows={len(rows)} max_err={err:.4f}", flush=True)
for M in (7776, 7680, 65792, 256 * 90, 256 * 256):
    for (N, K, res, act) in ((768, 768, True, 0), (768, 3072, True, 0), (3072, 768, False, 2), (2304, 768, False, 0), (1024, 1024, True, 0), (3072, 1024, False, 0)):
        check(M, N, K, res, act)
