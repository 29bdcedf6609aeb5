#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
rm -f $out/parity_raw.txt
SETOK_PARITY_LOG=$out/parity_raw.txt python -m pytest tests/test_e2e_gpu.py tests/test_fullsize_gpu.py tests/test_fp16_gpu.py tests/test_context_gpu.py -m gpu -q -s 2>&1 | grep -v "^\s*$" | tail -120 > $out/pytest_run3.log
ab() {
  for tag in "$@"; do
    if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
    SETOK_HIP_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --probe-every 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['ms_per_step'], r['achieved'], r['frac'], {k:v['tflops'] for k,v in r['per_class'].items()})"
  done
}
ab base pcr4 pcr6 st0 base pcr4 pcr6 st0 base pcr4 > $out/ab_store_policy_res.log 2>&1
tail -5 $out/pytest_run3.log; cat $out/ab_store_policy_res.log; wc -l $out/parity_raw.txt
