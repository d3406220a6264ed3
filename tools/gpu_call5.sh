#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "residual_through or full_size_batch_properties_bert_base or mpnet or (test_gemm_kernel and tile256 and f16)" > $OUT/pytest_c5.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_c5.log | cut -c1-400
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]
    print(sys.argv[2], {k:round(d[k],3) for k in ("value","ms_per_step")}, r["kernel"], round(r["avg_launch_us"],1), d["kernel_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for cfg in 3 4; do
  for v in 0 1 0 1; do
    BERT_HIP_RESID_STREAM=$v timeout 300 python bench.py --config $cfg --also --no-cpu-baseline --steps 3 --warmup 1 --repeat 2 > $OUT/bench_c5_${cfg}_$v.log 2> $OUT/bench_c5_${cfg}_$v.err; show $OUT/bench_c5_${cfg}_$v.log "config$cfg resid_stream=$v"
  done
done
