#!/bin/bash
# Same-box A/B of library builds on the whole cfg2 step, alternating: bash tools/ab_step.sh base old base old   ("base" = the default build)
for tag in "$@"; do
  if [ "$tag" == "base" ]; then lib=""; else lib="setok_amd/libsetok_hip_$tag.so"; fi
  SETOK_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['ms_per_step'], r['achieved'], r['frac'], {k:v['tflops'] for k,v in r['per_class'].items()})"
done
