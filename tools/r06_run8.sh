#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
for tag in tim tim4; do
  echo "===== $tag"
  sed "s/libsetok_hip_tim.so/libsetok_hip_$tag.so/" tools/pp_timing.sh > /tmp/ppt.sh
  bash /tmp/ppt.sh
done > $out/pp_timing_phases.log 2>&1
cat $out/pp_timing_phases.log
