#!/bin/bash
export TMPDIR=/tmp BERT_HIP_QUIET=1
for lib in f1; do
BERT_HIP_SK1_TIMELINE=1 BERT_HIP_LIB=${lib:+$PWD/bert.cpp_amd/libbert_$lib.so} BERT_HIP_LATENCY=2 timeout 200 python - <<'PY' 2>&1 | grep -v "^$" | cut -c1-1500 | head -9
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import bench
with tempfile.TemporaryDirectory() as d:
    r = bench.latency_b1(d, calls=100)
PY
done
