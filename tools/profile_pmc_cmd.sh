#!/bin/bash
# PMC-derived metrics of one command's kernels (one pass per metric, kernel-trace only): tools/profile_pmc_cmd.sh <tag> "<metrics>" <cmd...>
tag=$1; metrics=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r02}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in $metrics; do
  rm -rf /tmp/pp_$m
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $m -d /tmp/pp_$m -o u -- "$@" ) > /tmp/pp_$m.log 2>&1
  db=$(find /tmp/pp_$m -name "*.db" | head -1)
  if [ -n "$db" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db > $out/${tag}_pmc_$m.txt; echo "== $m"; head -8 $out/${tag}_pmc_$m.txt; else echo "== $m failed"; tail -3 /tmp/pp_$m.log; fi
done
