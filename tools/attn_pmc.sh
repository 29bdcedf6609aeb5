#!/bin/bash
# Wave-state counters of the ViT attention kernels at one shape: bash tools/attn_pmc.sh <tag> [T] [B]  (SETOK_ATTN_ROW=0|1 chooses the kernel)
tag=${1:-attn}; T=${2:-257}; B=${3:-256}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_WAIT_IFETCH SQ_LEVEL_WAVES"; do
  i=$((i+1)); rm -rf /tmp/ap_$i
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/ap_$i -o u -- python tools/bench_attn.py 64 $T $B ) > $out/pmc_${i}.log 2>&1
  db=$(find /tmp/ap_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db | grep -A10 "attn_vit" | head -12; else tail -3 $out/pmc_${i}.log; fi
done > $out/pmc_row${SETOK_ATTN_ROW:-1}_T${T}.txt 2>&1
cat $out/pmc_row${SETOK_ATTN_ROW:-1}_T${T}.txt
