#!/bin/bash
# round 5, experiment 5: the whole GPU suite on the new default (per-tile resync + streaming stores + hand-counted residual waits), step A/B against the round-4 behaviour
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp5_pytest.log 2>&1
bash tools/ab_step.sh rs0 base rs0 base > gpurun_out/r05_ab_step_final1.log 2>&1
bash tools/ab_gemm.sh 2 rs0 base > gpurun_out/r05_ab_gemm_final1.log 2>&1
bash tools/pp_timing.sh > gpurun_out/r05_pp_timing5.log 2>&1
tail -5 gpurun_out/r05_exp5_pytest.log; cat gpurun_out/r05_ab_step_final1.log gpurun_out/r05_ab_gemm_final1.log gpurun_out/r05_pp_timing5.log
