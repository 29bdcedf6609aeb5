#!/bin/bash
# Kernel-trace summary of N eager encode calls of B images: bash tools/latency_trace.sh <tag> [B] [calls]  ->  gpurun_out/<tag>/lat_b<B>_stats.csv
tag=${1:-r04}; B=${2:-1}; n=${3:-20}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/lt_$B
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/lt_$B -o k -- python tools/latency_eager.py $B $n ) > $out/lat_b${B}_trace.log 2>&1
db=$(find /tmp/lt_$B -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db > $out/lat_b${B}_stats.csv
head -16 $out/lat_b${B}_stats.csv
