#!/bin/bash
# round 5, experiment 1: where the ticks of a tile go (fine split), and streaming (nt) stores of the C tile — same-box A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/pp_timing.sh > gpurun_out/r05_pp_timing.log 2>&1
bash tools/ab_gemm.sh 2 base nt1 nt2 > gpurun_out/r05_ab_nt.log 2>&1
bash tools/ab_step.sh base nt2 nt1 base nt2 nt1 > gpurun_out/r05_ab_step_nt.log 2>&1
tail -30 gpurun_out/r05_pp_timing.log; cat gpurun_out/r05_ab_nt.log gpurun_out/r05_ab_step_nt.log
