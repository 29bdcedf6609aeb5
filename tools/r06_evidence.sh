#!/bin/bash
# round-6 evidence on ONE box: the driver-like bench (both probe modes), fp16, kernel table + PMC passes + wave states of the final build, timeline, the other workloads,
# micro-benchmarks, latency, vendor comparison, a 2-rank shared-GPU gloo validation of the multi-rank path (cfg2 and cfg4), smoke()
tag=r06
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 --probe-every 1 --no-live-traffic > $out/bench_probe_every_1.json 2> $out/bench_probe_every_1.err
python bench.py --steps 10 --warmup 3 --dtype f16 --no-cpu-baseline --no-live-traffic > $out/bench_f16.json 2> $out/bench_f16.err
bash tools/profile_round.sh $tag > $out/profile_round.log 2>&1
bash tools/profile_util.sh $tag > $out/profile_util.log 2>&1
bash tools/profile_wave_states.sh $tag > $out/profile_wave_states.log 2>&1
bash tools/step_timeline.sh $tag/tl --no-cpu-baseline --no-live-traffic > /dev/null 2>&1
for w in cfg3 cfg4-forward cfg4 cfg5; do
  python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err
done
cd /tmp && export TMPDIR=/tmp
for w in cfg4 cfg5; do
  rm -rf /tmp/pw_$w
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pw_$w -o k -- python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline ) > $out/prof_$w.log 2>&1
  db=$(find /tmp/pw_$w -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db > $out/bench_${w}_kernel_stats.csv
done
cd $GRAFT_REPO_ROOT
python tools/bench_gemm.py > $out/gemm_microbench.log 2>&1
python tools/bench_attn.py 64 >> $out/gemm_microbench.log 2>&1
python tools/bench_attn.py 64 577 128 >> $out/gemm_microbench.log 2>&1
python tools/bench_cluster.py > $out/cluster_microbench.log 2>&1
python tools/latency.py > $out/latency.log 2>&1
python tools/bench_vendor_gemm.py > $out/vendor_gemm.log 2>&1
# the multi-rank path on one GPU (validation only: gloo, both ranks on cuda:0)
timeout 600 python bench.py --gpus 2 --share-gpu --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic > $out/bench_2ranks_shared_gpu_gloo.json 2> $out/bench_2ranks_shared_gpu_gloo.err
timeout 600 python bench.py --gpus 2 --share-gpu --backend gloo --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_cfg4_2ranks_shared_gpu_gloo.json 2> $out/bench_cfg4_2ranks_shared_gpu_gloo.err
tail -1 $out/smoke.log; tail -1 $out/bench.json | cut -c1-400; tail -1 $out/bench_probe_every_1.json | cut -c1-200; tail -1 $out/bench_f16.json | cut -c1-200
for w in cfg3 cfg4-forward cfg4 cfg5; do tail -1 $out/bench_$w.json | cut -c1-160; done
tail -1 $out/bench_2ranks_shared_gpu_gloo.json | cut -c1-300; tail -1 $out/bench_cfg4_2ranks_shared_gpu_gloo.json | cut -c1-300
cat $out/latency.log | tail -3
