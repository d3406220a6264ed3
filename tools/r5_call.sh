#!/bin/bash
set -u
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 200 python tools/kernel_times.py 3 2>&1 | tail -1
BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_tl.so timeout 200 python tools/kernel_times.py 3 2>&1 | grep -E "attphase wg   0|attphase wg 100|cfg3" | cut -c1-400
timeout 200 python tools/kernel_times.py 4 2>&1 | tail -1
timeout 300 python -m pytest tests -m gpu -x -q -k "attention or bert_base or mpnet or full_size or latency or route" 2>&1 | tail -2
