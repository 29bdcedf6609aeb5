for a in "65792 1024 1024 0 1" "65792 1024 4096 0 1" "65792 3072 1024 0 0"; do for v in base setsrc; do echo -n "$v "; SETOK_HIP_LIB=_ab/libsetok_hip_$v.so SETOK_GEMM_TIMING=1 timeout 100 python tools/one_gemm.py $a 2>&1 | grep "gemm timing" | tail -1 | cut -c1-200; done; done
for i in 1 2 3; do for v in base setsrc; do echo "== $v"; SETOK_HIP_LIB=_ab/libsetok_hip_$v.so timeout 200 python tools/bench_gemm.py 2>&1 | grep -E "^(qkv|proj|fc1|fc2|inter) " | cut -c1-110; done; done
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "linear" 2>&1 | tail -2
for i in 1 2 3; do for v in base setsrc; do echo -n "$v "; SETOK_HIP_LIB=_ab/libsetok_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done; done
