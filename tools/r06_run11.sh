#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $out/pytest_final.log
timeout 400 python tools/fuzz_gpu.py 120 7 f16 > $out/fuzz_f16_seed7.log 2>&1
timeout 400 python tools/fuzz_gpu.py 120 11 bf16 > $out/fuzz_seed11.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $out/bench_second_box.json 2> /dev/null
cat $out/pytest_final.log; tail -1 $out/fuzz_f16_seed7.log; tail -1 $out/fuzz_seed11.log; tail -1 $out/bench_second_box.json | cut -c1-260
