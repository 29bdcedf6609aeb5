#!/bin/bash
# Same-box A/B of an ENVIRONMENT switch on a whole bench step, alternating: bash tools/ab_env.sh VAR valueA valueB [pairs] [bench args ...]
var=$1; a=$2; b=$3; n=${4:-3}; shift 4
for i in $(seq $n); do
  for v in "$a" "$b"; do
    env $var=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$var=$v', d['ms_per_step'], d['value'], r.get('frac'))"
  done
done
