#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
lib() { if [ "$1" == "base" ]; then echo ""; else echo "setok_amd/libsetok_hip_$1.so"; fi; }
for tag in ph4 base ph2s ph4 base ph2s; do
  echo "== $tag zeros"; SETOK_HIP_LIB=$(lib $tag) python tools/bench_gemm_zeros.py 2>&1 | tail -5
  echo "== $tag random"; SETOK_HIP_LIB=$(lib $tag) python tools/bench_gemm_plain.py 2>&1 | tail -5
done > $out/ab_phases_micro.log 2>&1
for tag in ph4 base; do
  echo "== $tag power fc1 random"; SETOK_HIP_LIB=$(lib $tag) python tools/power_probe.py 65792 4096 1024 4 2>&1 | tail -8
done > $out/ab_phases_power.log 2>&1
ab() {
  for tag in "$@"; do
    SETOK_HIP_LIB=$(lib $tag) timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --probe-every 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['ms_per_step'], r['achieved'], r['frac'], {k:v['tflops'] for k,v in r['per_class'].items()})"
  done
}
ab base ph2s ph4 base ph2s ph4 > $out/ab_phases2.log 2>&1
cat $out/ab_phases_micro.log $out/ab_phases_power.log $out/ab_phases2.log
