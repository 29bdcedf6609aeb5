#!/bin/bash
# round 5, experiment 3: the wave rows' offset set up and taken back per tile (PP_RESYNC), with and without streaming C stores
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "linear or gemm or ln" > gpurun_out/r05_exp3_pytest.log 2>&1
bash tools/pp_timing.sh > gpurun_out/r05_pp_timing3.log 2>&1
bash tools/ab_gemm.sh 2 rs0 rs1 rs1nt > gpurun_out/r05_ab_rs.log 2>&1
bash tools/ab_step.sh rs0 rs1 rs1nt rs0 rs1 rs1nt > gpurun_out/r05_ab_step_rs.log 2>&1
tail -5 gpurun_out/r05_exp3_pytest.log; cat gpurun_out/r05_pp_timing3.log gpurun_out/r05_ab_rs.log gpurun_out/r05_ab_step_rs.log
