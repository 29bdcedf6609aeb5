#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
lib() { if [ "$1" == "base" ]; then echo ""; else echo "setok_amd/libsetok_hip_$1.so"; fi; }
ab() {
  for tag in "$@"; do
    SETOK_HIP_LIB=$(lib $tag) timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --probe-every 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['ms_per_step'], r['achieved'], r['frac'], {k:v['tflops'] for k,v in r['per_class'].items()})"
  done
}
ab base ph2r base ph2r base ph2r > $out/ab_phases_reads_first.log 2>&1
for tag in base ph2r; do echo "== $tag zeros"; SETOK_HIP_LIB=$(lib $tag) python tools/bench_gemm_zeros.py 2>&1 | tail -5; done >> $out/ab_phases_reads_first.log 2>&1
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $out/pytest_run6.log
timeout 600 python tools/fuzz_gpu.py 150 5 > $out/fuzz_seed5.log 2>&1
cat $out/ab_phases_reads_first.log $out/pytest_run6.log; tail -3 $out/fuzz_seed5.log
