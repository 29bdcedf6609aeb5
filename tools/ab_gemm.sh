#!/bin/bash
# Same-box A/B of library builds on the sustained GEMM shapes: tools/ab_gemm.sh <rounds> <tag> [<tag> ...]   (tag "base" = the default library)
rounds=$1; shift
cd "$(dirname "$0")/.."
for r in $(seq 1 $rounds); do
  for tag in "$@"; do
    if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
    echo "== round $r  $tag"
    SETOK_HIP_LIB=$lib python tools/bench_gemm_steady.py 2>&1 | grep -v amdgpu.ids
  done
done
