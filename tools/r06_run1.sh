#!/bin/bash
# round 6, first lease: what a kernel boundary costs (micro), the C-tile store policy against the step's gaps (same-box A/B), timelines
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 ./tools/micro/launch_gap > $out/launch_gap.log 2>&1
ab() {
  for tag in "$@"; do
    if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
    SETOK_HIP_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --probe-every 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['ms_per_step'], r['achieved'], r['frac'], {k:v['tflops'] for k,v in r['per_class'].items()})"
  done
}
ab base pc4 pc6 pc5 base pc4 pc6 pc5 > $out/ab_store_policy.log 2>&1
bash tools/step_timeline.sh r06/tl_base --no-cpu-baseline --no-live-traffic > /dev/null 2>&1
SETOK_HIP_LIB=setok_amd/libsetok_hip_pc4.so bash tools/step_timeline.sh r06/tl_pc4 --no-cpu-baseline --no-live-traffic > /dev/null 2>&1
SETOK_HIP_LIB=setok_amd/libsetok_hip_pc6.so bash tools/step_timeline.sh r06/tl_pc6 --no-cpu-baseline --no-live-traffic > /dev/null 2>&1
cat $out/launch_gap.log $out/ab_store_policy.log
head -12 $out/tl_base/step_timeline.txt $out/tl_pc4/step_timeline.txt $out/tl_pc6/step_timeline.txt
