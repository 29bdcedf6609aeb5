#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $out/pytest_run2.log
for i in 1 2 3; do
  for tag in base st0; do
    if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
    echo "$tag $(SETOK_HIP_LIB=$lib python tools/bench_attn.py 2>&1 | tail -1) | $(SETOK_HIP_LIB=$lib python tools/bench_attn.py 64 577 128 2>&1 | tail -1)"
  done
done > $out/ab_attn_short_tail.log 2>&1
cat $out/pytest_run2.log $out/ab_attn_short_tail.log
