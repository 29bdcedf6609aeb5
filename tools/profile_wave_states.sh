#!/bin/bash
# Wave-state split of the bench's kernels (SQ counters, one PMC pass each, kernel-trace only): bash tools/profile_wave_states.sh r02
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  i=$((i+1)); rm -rf /tmp/pw_$i
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pw_$i -o u -- python bench.py --steps 2 --warmup 1 --timed-only ) > $out/pmc_wave_states_$i.log 2>&1
  db=$(find /tmp/pw_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db > $out/bench_pmc_wave_states_$i.txt; else tail -3 $out/pmc_wave_states_$i.log; fi
done
cat $out/bench_pmc_wave_states_*.txt | grep -A11 "gemm_pp_kernel<1, true, false>\|gemm_pp_kernel<0, false, true>\|attn_vit_kernel" | head -120
