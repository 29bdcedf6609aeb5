#!/bin/bash
# Where a tile's cycles go in the ping-pong GEMM (wave 0, s_memtime): build the instrumented library first,
#   bash tools/build_variant.sh tim gemm_persist.hip -DPP_TIMING
# then on the GPU box: bash tools/pp_timing.sh   (prints the LAST launch of each case: K loop / accumulator start / epilogue, cycles per tile)
export SETOK_GEMM_TIMING=1 SETOK_HIP_LIB=setok_amd/libsetok_hip_tim.so
python tools/bench_ln_gemm.py 0.25 2>&1 >/dev/null | awk '/^--- /{name=$0} /gemm timing\]/{last[name]=$0} /gemm timing [a-z]*\] wave row 0/{last[name]=last[name] "\n   " $0} /gemm timing [a-z]*\] wave row 1/{last[name]=last[name] "\n   " $0}  /gemm timing span/{last[name]=last[name] "\n   " $0} END{for (n in last) print n "\n   " last[n]}'
python - <<'PY' 2>&1 | awk '/^--- /{name=$0} /gemm timing\]/{last[name]=$0} /gemm timing [a-z]*\] wave row 0/{last[name]=last[name] "\n   " $0} /gemm timing [a-z]*\] wave row 1/{last[name]=last[name] "\n   " $0}  /gemm timing span/{last[name]=last[name] "\n   " $0} END{for (n in last) print n "\n   " last[n]}'
import sys, torch
sys.path.insert(0, ".")
from setok_amd import ops
M = 65792
for name, N, K, res in (("proj+residual", 1024, 1024, True), ("proj plain (no residual)", 1024, 1024, False), ("fc2+residual", 1024, 4096, True), ("sq8k", 8192, 8192, False)):
    m = 8192 if name == "sq8k" else M
    a = torch.randn(m, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.zeros(N, device="cuda"); r = torch.randn(m, N, device="cuda").bfloat16() if res else None
    out = torch.empty(m, N, device="cuda", dtype=torch.bfloat16)
    print(f"--- {name}", file=sys.stderr, flush=True)
    for _ in range(60):
        ops.linear(a, w, b, r, out=out)
    torch.cuda.synchronize()
PY
