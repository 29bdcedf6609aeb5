#!/bin/bash
# Kernel table of a bench invocation: bash tools/stats_cmd.sh <tag> [bench args]  ->  gpurun_out/<tag>_kernel_stats.csv (head printed)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sc
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/sc -o k -- python bench.py --steps 10 --warmup 3 --timed-only "$@" ) > /tmp/sc.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/sc -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
head -22 $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv | cut -c1-130; tail -1 /tmp/sc.log | cut -c1-160
