#!/bin/bash
# round-5 GPU call 2: instruction costs, attention variants (pipelined / straight loop x fp16 / f32 softmax x 8 / 4 waves), f32 route tests
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
t0=$(date +%s)
timeout 120 tools/ubench/valu_cost > $OUT/valu_cost.txt 2>&1; echo "valu_cost rc=$?"; cat $OUT/valu_cost.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "f32 or attention" > $OUT/pytest_c2.log 2>&1; echo "pytest rc=$? $(grep -E 'passed|failed' $OUT/pytest_c2.log | tail -1) [$(( $(date +%s) - t0 )) s]"
grep -E "^FAILED|^E  " $OUT/pytest_c2.log | head -30
for lib in "" p1e0 p0e1 p0e0; do
  if [ -n "$lib" ]; then export BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_$lib.so; else unset BERT_HIP_LIB; fi
  echo "== ${lib:-default} (8 waves)"; timeout 200 python tools/kernel_times.py 3 2>&1 | tail -1
done
for lib in "" p1e0; do
  if [ -n "$lib" ]; then export BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_$lib.so; else unset BERT_HIP_LIB; fi
  echo "== ${lib:-default} (4 waves)"; BERT_HIP_ATT_WAVES=4 timeout 200 python tools/kernel_times.py 3 2>&1 | tail -1
done
unset BERT_HIP_LIB
for lib in p0e0 p1e0; do
  export BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_$lib.so
  timeout 120 python tools/rate_probe.py 3 2>&1 | tail -1
done
unset BERT_HIP_LIB
timeout 120 python tools/rate_probe.py 3 2>&1 | tail -1
echo "total $(( $(date +%s) - t0 )) s"
