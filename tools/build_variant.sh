#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <tag> <file.hip> <-DFLAG ...>  ->  setok_amd/libsetok_hip_<tag>.so (same ABI; every object but
# <file.hip> is the default build's).  Run a tool against it with SETOK_HIP_LIB=setok_amd/libsetok_hip_<tag>.so (setok_amd/_lib.py).
set -e
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/../setok_amd/csrc"
make -j8 >/dev/null
mkdir -p build_$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $src -o build_$tag/${src%.hip}.o
objs=""
for o in build/*.o; do
  b=$(basename $o)
  if [ "$b" == "${src%.hip}.o" ]; then objs="$objs build_$tag/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../libsetok_hip_$tag.so
echo built ../libsetok_hip_$tag.so
