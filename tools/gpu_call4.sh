#!/bin/bash
# round 4, GPU call: gemm2x (two workgroups per CU) — parity, then configs 3 / 4 with and without it, start-skew sweep
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm2x or (test_gemm_kernel and tile2x) or (persistent and tile2x)" > $OUT/pytest_c4.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_c4.log | cut -c1-300
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
  for v in 0 1; do
    BERT_HIP_GEMM2X=$v timeout 300 python bench.py --config $cfg --also --no-cpu-baseline --steps 3 --warmup 1 --repeat 2 > $OUT/bench_c4_${cfg}_$v.log 2> $OUT/bench_c4_${cfg}_$v.err; show $OUT/bench_c4_${cfg}_$v.log "config$cfg gemm2x=$v"
  done
done
for sk in 0 32 192 384; do
  BERT_HIP_GEMM2X=1 BERT_HIP_GEMM2X_SKEW=$sk timeout 300 python bench.py --config 3 --also --no-cpu-baseline --steps 3 --warmup 1 --repeat 2 > $OUT/bench_c4_sk$sk.log 2> $OUT/bench_c4_sk$sk.err; show $OUT/bench_c4_sk$sk.log "config3 gemm2x skew=$sk"
done
timeout 300 python bench.py --also --no-cpu-baseline > $OUT/bench_c4_1.log 2> $OUT/bench_c4_1.err; echo "config1 rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_c4_1.log") if l.startswith("{")][-1]); print({k:round(d[k],4) for k in ("value","ms_per_step")}, d["roofline"])
PY
tail -2 $OUT/bench_c4_1.err
