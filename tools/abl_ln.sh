for tag in base noscale noinit; do
  if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
  echo "== $tag"; SETOK_HIP_LIB=$lib python tools/bench_ln_gemm.py 1.0 2>/dev/null | grep -v amdgpu | grep "round 1"
done
