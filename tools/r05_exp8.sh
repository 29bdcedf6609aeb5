#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SETOK_HIP_LIB=setok_amd/libsetok_hip_dar.so timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "linear" 2>&1 | tail -2
bash tools/ab_gemm.sh 2 base dar 2>&1 | tee gpurun_out/r05_ab_dar.log
bash tools/ab_step.sh base dar base dar 2>&1 | tee gpurun_out/r05_ab_step_dar.log
