#!/bin/bash
# round-6 evidence refresh of the FINAL build on one box (after the SwiGLU epilogue, the attention changes and the fp16 fix): parity log, driver-like bench in both probe modes,
# fp16, kernel table, cfg3-5 with their kernel tables, attention micro-benchmarks, latency
tag=r06b
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
rm -f $out/parity_raw.txt
SETOK_PARITY_LOG=$out/parity_raw.txt python -m pytest tests -m gpu -q 2>&1 | tail -4 > $out/pytest_final.log
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 --probe-every 1 --no-live-traffic --no-cpu-baseline > $out/bench_probe_every_1.json 2> /dev/null
python bench.py --steps 10 --warmup 3 --dtype f16 --no-cpu-baseline --no-live-traffic > $out/bench_f16.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pk
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python bench.py --steps 5 --warmup 2 --timed-only ) > $out/bench_under_rocprof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pk -name "*.db" | head -1) > $out/bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT
for w in cfg3 cfg4-forward cfg4 cfg5; do
  python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_$w.json 2> /dev/null
done
cd /tmp
for w in cfg4 cfg5; do
  rm -rf /tmp/pw_$w
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pw_$w -o k -- python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline ) > $out/prof_$w.log 2>&1
  db=$(find /tmp/pw_$w -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db > $out/bench_${w}_kernel_stats.csv
done
cd $GRAFT_REPO_ROOT
python tools/bench_attn.py 64 > $out/attn_microbench.log 2>&1
python tools/bench_attn.py 64 577 128 >> $out/attn_microbench.log 2>&1
python tools/bench_attn_causal.py >> $out/attn_microbench.log 2>&1
python tools/latency.py > $out/latency.log 2>&1
cat $out/pytest_final.log; tail -1 $out/bench.json | cut -c1-330; tail -1 $out/bench_probe_every_1.json | cut -c1-200; tail -1 $out/bench_f16.json | cut -c1-200
for w in cfg3 cfg4-forward cfg4 cfg5; do tail -1 $out/bench_$w.json | cut -c1-160; done
cat $out/attn_microbench.log; tail -2 $out/latency.log; head -8 $out/bench_kernel_stats.csv | cut -c1-120
