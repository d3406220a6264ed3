#!/bin/bash
# tools/variant.sh NAME "-DFLAG=.. ..."  ->  bert.cpp_amd/libbert_NAME.so: the whole library rebuilt with extra compiler flags (the
# Makefile's per-object flags kept), for A/B runs of kernel variants in a single gpurun call: BERT_HIP_LIB=bert.cpp_amd/libbert_NAME.so
set -e
cd "$(dirname "$0")/../bert.cpp_amd"
name=$1; flags=$2
make -s -j16 OBJ=build_var/$name LIB=libbert_$name.so TESTLIB=libbert_${name}_test.so \
     CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -Wno-unused-value $flags" libbert_$name.so libbert_${name}_test.so
echo "built bert.cpp_amd/libbert_${name}.so"
