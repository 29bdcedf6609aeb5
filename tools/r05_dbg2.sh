#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for tag in rs1 rs1nt; do echo "== $tag"; SETOK_HIP_LIB=setok_amd/libsetok_hip_$tag.so python tools/r05_dbg2.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_dbg2.log 2>&1
cat gpurun_out/r05_dbg2.log
