#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
lib() { if [ "$1" == "base" ]; then echo ""; else echo "setok_amd/libsetok_hip_$1.so"; fi; }
ab() {
  for tag in "$@"; do
    SETOK_HIP_LIB=$(lib $tag) timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --probe-every 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['ms_per_step'], r['achieved'], r['frac'], {k:v['tflops'] for k,v in r['per_class'].items()})"
  done
}
ab base stg3 stg6 stg10 base stg3 stg6 stg10 > $out/ab_stagger.log 2>&1
cat $out/ab_stagger.log
