#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for tag in rs0 rs1 ntold rs1nt ep0; do
  echo "== $tag"
  SETOK_HIP_LIB=setok_amd/libsetok_hip_$tag.so timeout 300 python -m pytest tests/test_detok_gpu.py -m gpu -x -q -k full_dims 2>&1 | tail -3
done > gpurun_out/r05_dbg1.log 2>&1
cat gpurun_out/r05_dbg1.log
