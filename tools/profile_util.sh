#!/bin/bash
# MFMA / LDS utilisation of the bench's kernels from PMC-derived metrics (separate passes, kernel-trace only): bash tools/profile_util.sh r01
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in MfmaUtil LdsUtil LdsBankConflict; do
  rm -rf /tmp/pu_$m
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --pmc $m -d /tmp/pu_$m -o u -- python bench.py --steps 2 --warmup 1 --timed-only ) > $out/pmc_$m.log 2>&1
  db=$(find /tmp/pu_$m -name "*.db" | head -1)
  if [ -n "$db" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db > $out/bench_pmc_$m.txt; grep -A3 "gemm_pp_kernel\|gemm_persist_kernel\|attn_vit" $out/bench_pmc_$m.txt | head -40; else tail -3 $out/pmc_$m.log; fi
done
