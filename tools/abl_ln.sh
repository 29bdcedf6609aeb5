#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.sh) on the two LayerNorm-folded ViT shapes + their plain forms:
#   bash tools/abl_ln.sh base s0 e1 ...      ("base" = the default build)
for tag in "$@"; do
  if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
  echo "== $tag"; SETOK_HIP_LIB=$lib python tools/bench_ln_gemm.py 1.0 2>/dev/null | grep -v amdgpu | grep "round 1"
done
