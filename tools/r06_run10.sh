#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_fullsize_gpu.py tests/test_e2e_gpu.py tests/test_detok_gpu.py -m gpu -q -x 2>&1 | tail -15 > $out/pytest_run10.log
lib() { if [ "$1" == "base" ]; then echo ""; else echo "setok_amd/libsetok_hip_$1.so"; fi; }
for i in 1 2 3; do
  for tag in base lr0; do
    echo "$tag $(SETOK_HIP_LIB=$(lib $tag) python tools/bench_attn.py 2>&1 | tail -1) | $(SETOK_HIP_LIB=$(lib $tag) python tools/bench_attn.py 64 577 128 2>&1 | tail -1)"
  done
done > $out/ab_attn_lone_row.log 2>&1
ab() {
  for tag in "$@"; do
    SETOK_HIP_LIB=$(lib $tag) timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --probe-every 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['ms_per_step'], r['achieved'], r['frac'])"
  done
}
ab base lr0 base lr0 base lr0 >> $out/ab_attn_lone_row.log 2>&1
cat $out/pytest_run10.log $out/ab_attn_lone_row.log
