#!/bin/bash
# round 5: the whole GPU suite (twice: the failures of this round were races) + the fuzzer on the build that is kept
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_exp7_pytest.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -k "full or batch or invariance or linear" > gpurun_out/r05_exp7_pytest2.log 2>&1
bash tools/ab_step.sh rs0 base rs0 base > gpurun_out/r05_ab_step_final2.log 2>&1
tail -4 gpurun_out/r05_exp7_pytest.log; tail -4 gpurun_out/r05_exp7_pytest2.log; cat gpurun_out/r05_ab_step_final2.log
