#!/bin/bash
# build_exp/<name>/{libgeotr_hip.so, abi_bench.bin}: the library with ONE source rebuilt under extra flags (measurement variants; git-ignored,
# shipped to the GPU box by gpurun).  usage: scripts/build_variant.sh NAME SOURCE.hip "-DFLAG=1 ..."
set -e
NAME=$1; SRC=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build_exp/$NAME
mkdir -p $OUT
cd $ROOT/geotransformer_amd/csrc
for f in *.o; do [ "$f" != "${SRC%.hip}.o" ] && cp -u $f $OUT/; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function $FLAGS -c $SRC -o $OUT/${SRC%.hip}.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libgeotr_hip.so $OUT/*.o
cd $ROOT
hipcc -O2 -std=c++17 $FLAGS -I include scripts/abi_bench.cpp -L $OUT -lgeotr_hip -Wl,-rpath,'$ORIGIN' -o $OUT/abi_bench.bin
ls -la $OUT/libgeotr_hip.so $OUT/abi_bench.bin
