import sys, torch
sys.path.insert(0, ".")
from setok_amd import ops
torch.manual_seed(0)
ok = True
for M in (2056, 2048, 1028, 1100, 1500, 2176, 2180):
    for N, K in ((1024, 1024), (1024, 4096), (512, 1024)):
        for act in (0, 2):
            a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
            b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").bfloat16()
            out = ops.linear(a, w, b, r, act=act)
            ref = a.float() @ w.float().t() + b
            if act == 2: ref = torch.nn.functional.gelu(ref)
            ref = ref.bfloat16().float() + r.float()
            err = (out.float() - ref).abs().max().item()
            # the same rows in a small call (another kernel shape): bit-identical
            sub = ops.linear(a[:257].contiguous(), w, b, r[:257].contiguous(), act=act)
            tail = ops.linear(a[M - 100:].contiguous(), w, b, r[M - 100:].contiguous(), act=act)
            same = bool(torch.equal(sub, out[:257])) and bool(torch.equal(tail, out[M - 100:]))
            print(M, N, K, act, f"max err {err:.4f}", "rows bit-identical:", same, flush=True)
            ok = ok and same and err < 0.1
print("ALL OK" if ok else "FAILED")
