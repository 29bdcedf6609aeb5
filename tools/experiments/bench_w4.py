#!/usr/bin/env python
"""A/B of the four-wave GEMM kernel (SETOK_GEMM_W4=1) against the eight-wave one on the ViT shapes without residual: bit-identity of the
results, then sustained rates, alternating.  python tools/bench_w4.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
for name, M, N, K, act in (("qkv", 65792, 3072, 1024, 0), ("fc1+quick_gelu", 65792, 4096, 1024, 1), ("sq8k", 8192, 8192, 8192, 0), ("edge", 65792 - 100, 3072, 1024, 2)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    outs = {}
    for v in ("0", "1"):
        os.environ["SETOK_GEMM_W4"] = v
        outs[v] = ops.linear(a, w, b, act=act)
    torch.cuda.synchronize()
    same = torch.equal(outs["0"], outs["1"])
    diff = (outs["0"].float() - outs["1"].float()).abs().max().item()
    print(f"{name}: four-wave == eight-wave bit for bit: {same} (max |diff| {diff:.3g})", flush=True)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for rnd in range(2):
        for v in ("0", "1"):
            os.environ["SETOK_GEMM_W4"] = v
            ops.linear(a, w, b, act=act, out=out); torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < secs:
                for _ in range(20):
                    ops.linear(a, w, b, act=act, out=out)
                torch.cuda.synchronize(); n += 20
            dt = (time.perf_counter() - t0) / n
            print(f"  {name:16s} {'four-wave ' if v == '1' else 'eight-wave'} round {rnd}: {dt * 1e6:7.1f} us  {2.0 * M * N * K / dt / 1e12:7.1f} TFLOP/s", flush=True)
os.environ.pop("SETOK_GEMM_W4", None)
