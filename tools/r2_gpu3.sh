#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "qkv_attention or full_size or families or baseline_models"  > $OUT/r2f_pytest_q2.log 2>&1; echo "pytest q2 rc=$?"; tail -5 $OUT/r2f_pytest_q2.log
BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_tl.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also > $OUT/r2f_tl.json 2> $OUT/r2f_tl.err; echo "tl rc=$?"
grep rawtimeline $OUT/r2f_tl.err | head -1 | cut -c1-1800
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --also"
for v in ${VARIANTS:-default}; do
  lib=bert.cpp_amd/libbert.so; [ $v != default ] && lib=bert.cpp_amd/libbert_$v.so
  BERT_HIP_LIB=$PWD/$lib timeout 300 $B > $OUT/r2f_bench_${v}.json 2> $OUT/r2f_bench_${v}.err; echo "bench $v rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2f_bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d['kernel_ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
