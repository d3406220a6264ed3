#!/bin/bash
# q4 forms of the two layer kernels: kernel parity tests, then config 2 as written (BERT_HIP_Q4=fused) next to the expanded default
export TMPDIR=/tmp BERT_HIP_QUIET=1
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "layer_tail_kernel or qkv_attention2_kernel_q4 or q4 or embed" > $OUT/q4_tail_tests.log 2>&1
tail -5 $OUT/q4_tail_tests.log
for c in 2 22 1; do STEPS=20 REPEAT=2 timeout 300 python tools/kernel_times.py $c 2>&1 | tail -1; done > $OUT/q4_tail_times.log
cat $OUT/q4_tail_times.log
