#!/bin/bash
# round-5 GPU call 1: parity of the new softmax / pipelined attention, A/B of the headline, bench line size and time
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
t0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_c1.log 2>&1; echo "pytest rc=$? $(grep -E 'passed|failed' $OUT/pytest_c1.log | tail -1) [$(( $(date +%s) - t0 )) s]"
grep -E "^FAILED|Error|assert" $OUT/pytest_c1.log | head -20
for lib in "" exp0; do
  if [ -n "$lib" ]; then export BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_$lib.so; else unset BERT_HIP_LIB; fi
  timeout 120 python tools/rate_probe.py 3 2>&1 | tail -1
done
unset BERT_HIP_LIB
for lib in "" exp0; do
  if [ -n "$lib" ]; then export BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_$lib.so; else unset BERT_HIP_LIB; fi
  timeout 120 python tools/rate_probe.py 3 2>&1 | tail -1
done
unset BERT_HIP_LIB
timeout 200 python tools/kernel_times.py 3 2>&1 | tail -1
timeout 200 python tools/kernel_times.py 5 2>&1 | tail -1
t1=$(date +%s)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.log 2> $OUT/bench_c1.err; echo "bench rc=$? [$(( $(date +%s) - t1 )) s] line bytes $(tail -1 $OUT/bench_c1.log | wc -c)"
tail -1 $OUT/bench_c1.log | cut -c1-3000
grep "^\[bench\]" $OUT/bench_c1.err
cp bench_detail.json $OUT/bench_detail_c1.json 2>/dev/null
echo "total $(( $(date +%s) - t0 )) s"
