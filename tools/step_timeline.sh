#!/bin/bash
# bash tools/step_timeline.sh <tag> [bench args]  ->  gpurun_out/<tag>/step_timeline.txt
tag=${1:-r05}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace -d /tmp/tl -o k -- python bench.py --steps 2 --warmup 1 --timed-only "$@" ) > $out/step_timeline_run.log 2>&1
db=$(find /tmp/tl -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/step_timeline.py $db | tee $out/step_timeline.txt
python $GRAFT_REPO_ROOT/tools/trace_seq.py $db > $out/step_seq.txt
