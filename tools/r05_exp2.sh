#!/bin/bash
# round 5, experiment 2: the last K-tile slot by slot; streaming loads / stores in the ViT attention; residual rows as streaming loads
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/pp_timing.sh > gpurun_out/r05_pp_timing2.log 2>&1
for r in 1 2; do for tag in base ant1 ant2 ant4 ant7; do
  if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
  echo "== $r $tag"; SETOK_HIP_LIB=$lib python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/r05_ab_attn_nt.log 2>&1
bash tools/ab_gemm.sh 2 nt1 nt1r > gpurun_out/r05_ab_nt1r.log 2>&1
cat gpurun_out/r05_pp_timing2.log gpurun_out/r05_ab_attn_nt.log gpurun_out/r05_ab_nt1r.log
