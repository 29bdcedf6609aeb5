#!/bin/bash
# does the policy of the residual launches' C stores (h = the residual stream, 135 MB, read next by row_stats and by the LayerNorm-folded GEMM) decide where those reads come from?
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for tag in base pcr0 pcr4 base pcr0; do
  if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
  rm -rf /tmp/sc
  ( cd $GRAFT_REPO_ROOT && SETOK_HIP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/sc -o k -- python bench.py --steps 6 --warmup 2 --timed-only --no-live-traffic ) > /tmp/sc.log 2>&1
  echo "== $tag: $(tail -1 /tmp/sc.log | cut -c1-120)"
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/sc -name "*.db" | head -1) | head -12 | cut -c1-150
done > $out/store_policy_residual_kernel_stats.log 2>&1
cat $out/store_policy_residual_kernel_stats.log
