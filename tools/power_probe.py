#!/usr/bin/env python
"""Run one GEMM shape in a loop for a few seconds and sample rocm-smi (clock / power) meanwhile: python tools/power_probe.py M N K [secs]"""
import sys, os, subprocess, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from setok_amd import ops
M, N, K = map(int, sys.argv[1:4]); secs = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
zero = len(sys.argv) > 5 and sys.argv[5] == "zero"
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
if zero:
    a.zero_(); w.zero_()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
samples = []
stop = False
def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
        samples.append([l.strip() for l in r.splitlines() if any(k in l for k in ("sclk", "Power", "junction", "mclk"))])
        time.sleep(0.3)
t = threading.Thread(target=sampler); t.start()
torch.cuda.synchronize(); t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.time() - t0 < secs:
    e0.record()
    for _ in range(50):
        ops.linear(a, w, None, out=out)
    e1.record(); torch.cuda.synchronize(); n += 50
    ms = e0.elapsed_time(e1) / 50
print(f"last-batch {ms:.4f} ms/GEMM = {2.0 * M * N * K / ms / 1e9:.1f} TF  ({'zeros' if zero else 'random'})")
stop = True; t.join()
for s in samples[1::3]:
    print(" | ".join(s))
