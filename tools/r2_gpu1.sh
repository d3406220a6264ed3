#!/bin/bash
# round 2, GPU call 1: parity of the new kernels, A/B timings of kernel variants, config-3 breakdown
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/r2a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r2a_pytest.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "qkv_attention2" > $OUT/r2a_pytest_q2.log 2>&1; echo "pytest q2 rc=$?"; tail -15 $OUT/r2a_pytest_q2.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --also"
for rep in 1 2; do
  for v in default g0b0 g1b0; do
    lib=bert.cpp_amd/libbert.so; [ $v != default ] && lib=bert.cpp_amd/libbert_$v.so
    BERT_HIP_LIB=$PWD/$lib timeout 300 $B > $OUT/r2a_bench_${v}_$rep.json 2> $OUT/r2a_bench_${v}_$rep.err; echo "bench $v rc=$?"
  done
  BERT_HIP_QKV2=0 timeout 300 $B > $OUT/r2a_bench_noq2_$rep.json 2>/dev/null; echo "bench noq2 rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2a_bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout 600 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --also > $OUT/r2a_bench_c3.json 2> $OUT/r2a_bench_c3.err; echo "c3 rc=$?"; cut -c1-1500 $OUT/r2a_bench_c3.json
timeout 300 python tools/mixed_len_bench.py 16384 > $OUT/r2a_mixed_q2.log 2>&1; cat $OUT/r2a_mixed_q2.log
BERT_HIP_QKV2=0 timeout 300 python tools/mixed_len_bench.py 16384 > $OUT/r2a_mixed_noq2.log 2>&1; cat $OUT/r2a_mixed_noq2.log
