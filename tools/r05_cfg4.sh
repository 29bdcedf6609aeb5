#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_autograd_gpu.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do
  ( cd _ab/head && python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head', d['ms_per_step'])" )
  python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r05_ab_cfg4_colsum.log
