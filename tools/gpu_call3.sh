#!/bin/bash
# round 4, third GPU call: the pass-ordering event without a system-scope fence (L2 stays warm from step to step)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
show() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]
print({k:round(d[k],4) for k in ("value","ms_per_step")}, "h2h", round(d.get("host_to_host",{}).get("value",0)), {k:round(r[k],2) for k in ("avg_launch_us","avg_launch_us_timed_alone","frac","step_share")}, d["kernel_ms_per_step"])
if "also" in d and "latency_b1" in d["also"]: print({k:round(v["median_us"],1) for k,v in d["also"]["latency_b1"].items() if isinstance(v,dict)})
PY
}
timeout 300 python bench.py --also --no-cpu-baseline > $OUT/bench_c3a.log 2> $OUT/bench_c3a.err; echo "bench (system-scope host events) rc=$?"; show $OUT/bench_c3a.log; tail -2 $OUT/bench_c3a.err
BERT_HIP_HOST_EVENT_SCOPE=device timeout 300 python bench.py --also --no-cpu-baseline > $OUT/bench_c3b.log 2> $OUT/bench_c3b.err; echo "bench (device-scope host events) rc=$?"; show $OUT/bench_c3b.log; tail -2 $OUT/bench_c3b.err
for sc in system device; do
BERT_HIP_HOST_EVENT_SCOPE=$sc timeout 200 python - <<'PY'
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
with tempfile.TemporaryDirectory() as d:
    print(os.environ["BERT_HIP_HOST_EVENT_SCOPE"], {k: round(v["median_us"], 1) for k, v in bench.latency_b1(d).items() if isinstance(v, dict)})
PY
done
BERT_HIP_HOST_EVENT_SCOPE=device timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_device.py tests/test_gpu_examples.py -m gpu -x -q -k "api_ or host_path or full_size or one_launch or latency or config5 or gather or device_api or encode or examples or multi_rank or workspace" > $OUT/pytest_c3.log 2>&1; echo "pytest (device-scope host events) rc=$?"; tail -3 $OUT/pytest_c3.log
BERT_HIP_HOST_EVENT_SCOPE=device timeout 300 python tools/stress_determinism.py > $OUT/stress_c3.log 2>&1; echo "stress rc=$?"; tail -3 $OUT/stress_c3.log
