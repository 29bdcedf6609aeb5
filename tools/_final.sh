mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_full.log 2>&1; tail -3 gpurun_out/r02/pytest_full.log
timeout 600 python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; tail -c 300 gpurun_out/r02/bench.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for w in cfg4 cfg4-forward cfg3 cfg5; do timeout 600 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r02/bench_$w.json 2> gpurun_out/r02/bench_$w.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/r02/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['gemm_share_of_step'])"; done
timeout 600 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r02/bench_cfg2_f32.json 2> gpurun_out/r02/bench_cfg2_f32.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/r02/bench_cfg2_f32.json').read().strip().splitlines()[-1]); print('f32', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
bash tools/profile_round.sh r02 > gpurun_out/r02/profile_round.log 2>&1; tail -30 gpurun_out/r02/profile_round.log
bash tools/profile_util.sh r02 > gpurun_out/r02/profile_util.log 2>&1; tail -12 gpurun_out/r02/profile_util.log
timeout 200 python tools/bench_cluster.py > gpurun_out/r02/cluster_microbench.log 2>&1; timeout 300 python tools/bench_gemm.py > gpurun_out/r02/gemm_microbench.log 2>&1; tail -3 gpurun_out/r02/gemm_microbench.log | cut -c1-300
