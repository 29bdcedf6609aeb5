#!/bin/bash
# Everything under profiles/ for one round, on the GPU box: bash tools/profile_all.sh r01   (≈10 GPU-minutes)
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $tag > $out/profile_round.log 2>&1
bash tools/profile_util.sh $tag > $out/profile_util.log 2>&1
python bench.py > $out/bench_final.json 2> $out/bench_final.err
for w in cfg3 cfg4-forward cfg4 cfg5; do
  python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err
done
cd /tmp && export TMPDIR=/tmp
for w in cfg3 cfg4 cfg5; do
  rm -rf /tmp/pw_$w
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pw_$w -o k -- python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline ) > $out/prof_$w.log 2>&1
  db=$(find /tmp/pw_$w -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db > $out/bench_${w}_kernel_stats.csv
done
cd $GRAFT_REPO_ROOT
python tools/bench_gemm.py > $out/gemm_microbench.log 2>&1
python tools/bench_attn.py 64 >> $out/gemm_microbench.log 2>&1
python tools/bench_attn.py 48 >> $out/gemm_microbench.log 2>&1
python tools/bench_attn_causal.py >> $out/gemm_microbench.log 2>&1
python tools/bench_cluster.py > $out/cluster_microbench.log 2>&1
python tools/latency.py > $out/latency.log 2>&1
python tools/bench_vendor_gemm.py > $out/vendor_gemm.log 2>&1
tail -1 $out/bench_final.json | cut -c1-200
for w in cfg3 cfg4-forward cfg4 cfg5; do tail -1 $out/bench_$w.json | cut -c1-160; done
