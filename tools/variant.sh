#!/bin/bash
# tools/variant.sh NAME SRC.hip "-DFLAG=.. ..."  ->  bert.cpp_amd/libbert_NAME.so: libbert.so with ONE translation unit rebuilt
# with extra flags (A/B runs of a kernel variant in a single gpurun call: BERT_HIP_LIB=bert.cpp_amd/libbert_NAME.so).
set -e
cd "$(dirname "$0")/../bert.cpp_amd"
name=$1; src=$2; flags=$3
make -s libbert.so >/dev/null
mkdir -p build_var
obj=build_var/${name}_$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value -mllvm -structurizecfg-skip-uniform-regions=true $flags -c csrc/$src -o $obj
objs=$(ls build/*.o | grep -v "build/$(basename ${src%.hip}).o" | grep -v build/test_api.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libbert_${name}.so $objs $obj -ldl -lpthread
echo "built bert.cpp_amd/libbert_${name}.so"
