#!/bin/bash
# Experiments: a variant of librgnn.so with extra -D flags on ONE source file, linked into tools/var/<name>/librgnn.so.
#   tools/build_variant.sh <name> <source.hip> "<flags>"      then   LD_LIBRARY_PATH=tools/var/<name> tools/x3_bench.bin
set -e
name=$1; src=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/tools/var/$name
obj=$root/tools/var/$name/$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form $flags -c $root/radargnn_amd/csrc/$src -o $obj 
objs=""
for o in $root/radargnn_amd/build/*.o; do
  if [ "$(basename $o)" == "$(basename $obj)" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $root/tools/var/$name/librgnn.so
echo built tools/var/$name/librgnn.so
