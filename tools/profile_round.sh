#!/bin/bash
# Round profile on the GPU box: kernel-trace summary of the bench + the two PMC passes for HBM traffic.
# usage (through gpurun): bash tools/profile_round.sh r02
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk /tmp/pf /tmp/pw
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python bench.py --steps 5 --warmup 2 --timed-only ) > $out/bench_under_rocprof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pk -name "*.db" | head -1) > $out/bench_kernel_stats.csv
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o f -- python bench.py --steps 2 --warmup 1 --timed-only ) > $out/pmc_fetch.log 2>&1
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o w -- python bench.py --steps 2 --warmup 1 --timed-only ) > $out/pmc_write.log 2>&1
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $F > $out/bench_pmc_FETCH_SIZE.txt
python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $W > $out/bench_pmc_WRITE_SIZE.txt
python $GRAFT_REPO_ROOT/tools/gemm_traffic.py $F $W > $out/traffic.json || { echo "gemm_traffic.py FAILED"; rm -f $out/traffic.json; }
head -24 $out/bench_kernel_stats.csv; cat $out/traffic.json; tail -2 $out/bench_under_rocprof.log
