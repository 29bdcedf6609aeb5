#!/bin/bash
# Durations of ONE kernel (substring) inside the cfg2 step for several library builds: bash tools/ab_kernel.sh <kernel-substring> base tagA tagB ...
k=$1; shift
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" == "base" ]; then lib=""; else lib="$GRAFT_REPO_ROOT/setok_amd/libsetok_hip_$tag.so"; fi
  rm -rf /tmp/abk
  ( cd $GRAFT_REPO_ROOT && SETOK_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abk -o k -- python bench.py --steps 3 --warmup 1 --timed-only >/dev/null 2>&1 )
  echo "== $tag"; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/abk -name "*.db" | head -1) | grep "$k" | cut -c1-140
done
