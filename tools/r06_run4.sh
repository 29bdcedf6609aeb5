#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
# the 2-phase K-tile build: bits (every GEMM test + the fuzzer's GEMM part run against it), then the step, alternating
SETOK_HIP_LIB=setok_amd/libsetok_hip_ph2.so python -m pytest tests/test_ops_gpu.py -m gpu -q -k "linear or gemm or layernorm" 2>&1 | tail -5 > $out/ph2_tests.log
ab() {
  for tag in "$@"; do
    if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
    SETOK_HIP_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --probe-every 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['ms_per_step'], r['achieved'], r['frac'], {k:v['tflops'] for k,v in r['per_class'].items()})"
  done
}
ab base ph2 base ph2 base ph2 > $out/ab_phases.log 2>&1
cat $out/ph2_tests.log $out/ab_phases.log
