#!/bin/bash
# Kernel-trace summary of one command on the GPU box: tools/profile_cmd.sh <tag> <cmd...>   -> gpurun_out/${ROUND:-r02}/<tag>_stats.csv
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r02}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- "$@" ) > /tmp/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db > $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r02}/${tag}_stats.csv
head -30 $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r02}/${tag}_stats.csv
