#!/bin/bash
# round 4, first GPU call: the new parity tests, then the whole bench line
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 900 python -m pytest tests/test_gpu_exactness.py tests/test_gpu_parity.py -m gpu -x -q -k "exactness or gpu_box or gpu_context or gemm256_q4 or test_gemm_kernel or gemm_persistent or full_size or config5 or q4_expanded" > $OUT/pytest_new.log 2>&1; echo "pytest-new rc=$?"; tail -5 $OUT/pytest_new.log
timeout 300 python -m pytest tests/test_multi_device.py -m gpu -x -q > $OUT/pytest_md.log 2>&1; echo "pytest-md rc=$?"; tail -3 $OUT/pytest_md.log
timeout 900 python bench.py > $OUT/bench_c1.log 2> $OUT/bench_c1.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench_c1.log; tail -5 $OUT/bench_c1.err
