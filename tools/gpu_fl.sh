#!/bin/bash
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency or skinny or route or hidden or promise" 2>&1 | tail -3
for m in 1 2; do
timeout 200 python - <<'PY' 2>&1 | grep -v "^$" | cut -c1-600
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import bench
with tempfile.TemporaryDirectory() as d:
    r = bench.latency_b1(d, calls=400)
    print({k: round(v["median_us"], 1) for k, v in r.items() if isinstance(v, dict)}, r["f16_n128"]["kernel_us"])
PY
done
