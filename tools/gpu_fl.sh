#!/bin/bash
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency or skinny or route" 2>&1 | tail -3
BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_kp2.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency or skinny or route" 2>&1 | tail -3
for lib in "" kp kp2 "" kp kp2; do
BERT_HIP_LIB=${lib:+$PWD/bert.cpp_amd/libbert_$lib.so} timeout 200 python - <<'PY' 2>&1 | grep -v "^$" | cut -c1-600
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import bench
with tempfile.TemporaryDirectory() as d:
    r = bench.latency_b1(d, calls=400)
    print(os.environ.get("BERT_HIP_LIB", "")[-11:] or "default", {k: round(v["median_us"], 1) for k, v in r.items() if isinstance(v, dict)})
PY
done
