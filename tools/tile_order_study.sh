#!/bin/bash
# FETCH_SIZE, step time, socket power and shader clock per tile order (PP_GM = M-tiles per group; an XCD's 32 consecutive
# tiles are a GM x 32 / GM patch).  Libraries built beforehand with tools/build_variant.sh gm<N> gemm_persist.hip -DPP_GM=<N>.
# usage (through gpurun): bash tools/tile_order_study.sh r04
tag=${1:-r04}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
: > $out/tile_order.log
for v in base gm2 gm4 gm16; do
  if [ "$v" == "base" ]; then lib=""; else lib="$GRAFT_REPO_ROOT/setok_amd/libsetok_hip_$v.so"; fi
  echo "== $v (base = PP_GM 8)" >> $out/tile_order.log
  ( cd $GRAFT_REPO_ROOT && SETOK_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null ) | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('   ms_per_step', d['ms_per_step'], ' GEMM TFLOP/s', r['achieved'], ' sclk MHz', r.get('sclk_mhz_under_load'), ' socket W', r.get('socket_power_w_under_load'), ' per class', {k:v['tflops'] for k,v in r['per_class'].items()})" >> $out/tile_order.log
  rm -rf /tmp/to_$v
  ( cd $GRAFT_REPO_ROOT && SETOK_HIP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/to_$v -o f -- python bench.py --steps 2 --warmup 1 --timed-only ) > /tmp/to_$v.log 2>&1
  db=$(find /tmp/to_$v -name "*.db" | head -1)
  if [ -n "$db" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db gemm_pp_kernel | grep -A1 "gemm_pp_kernel" | grep -v "^--" >> $out/tile_order.log; else echo "   pmc pass failed" >> $out/tile_order.log; fi
done
cat $out/tile_order.log
