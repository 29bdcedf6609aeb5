#!/bin/bash
# round 5, experiment 4: default build = per-tile resync + streaming stores + hand-counted residual waits; the VALU priority of the two wave rows in the epilogue
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp4_pytest.log 2>&1
bash tools/pp_timing.sh > gpurun_out/r05_pp_timing4.log 2>&1
bash tools/ab_gemm.sh 2 rs1nt ep0 ep1 ep2 > gpurun_out/r05_ab_ep.log 2>&1
bash tools/ab_step.sh rs1nt ep0 ep1 ep2 rs1nt ep0 ep1 ep2 > gpurun_out/r05_ab_step_ep.log 2>&1
tail -5 gpurun_out/r05_exp4_pytest.log; cat gpurun_out/r05_pp_timing4.log gpurun_out/r05_ab_ep.log gpurun_out/r05_ab_step_ep.log
