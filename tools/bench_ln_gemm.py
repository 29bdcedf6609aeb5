#!/usr/bin/env python
"""A/B of a Linear with the LayerNorm folded in (setok_linear_ln) against the plain Linear on the two ViT-L shapes that use it, sustained.
SETOK_GEMM_TIMING=1 prints the per-tile s_memtime breakdown of each launch.  python tools/bench_ln_gemm.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
M, K = 65792, 1024
x = (torch.randn(M, K, device="cuda") + 0.3).bfloat16()
gamma, beta = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
stats = ops.row_stats(x)
for name, N, act in (("qkv", 3072, 0), ("fc1+quick_gelu", 4096, 1)):
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.zeros(N, device="cuda")
    folded = ops.ln_fold(w, gamma, beta, b)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    variants = {"plain": lambda: ops.linear(x, w, b, act=act, out=out), "ln-fold": lambda: ops.linear_ln(x, folded, stats, act=act, out=out)}
    for rnd in range(2):
        for vn, fn in variants.items():
            print(f"--- {name} {vn}", file=sys.stderr, flush=True)
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < secs:
                for _ in range(20):
                    fn()
                torch.cuda.synchronize(); n += 20
            dt = (time.perf_counter() - t0) / n
            print(f"{name:16s} {vn:8s} round {rnd}: {dt * 1e6:7.1f} us  {2.0 * M * N * K / dt / 1e12:7.1f} TFLOP/s", flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
y = torch.empty_like(x)
for name, fn in (("row_stats", lambda: ops.row_stats(x, out=stats)), ("layernorm", lambda: ops.layernorm(x, gamma, beta, out=y))):
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
