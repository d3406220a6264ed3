#!/bin/bash
# tools/pmc_gemm.sh [extra env ...]: FETCH_SIZE / WRITE_SIZE / TCC hit-miss per kernel of config 3 (one pass each), for the library and
# switches in the environment.  Writes gpurun_out/pmc_gemm_<tag>.txt (tag = $TAG).
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=${TAG:-x}
export TMPDIR=/tmp BERT_HIP_QUIET=1
ROOT=$PWD
cd /tmp
P="python $ROOT/bench.py --config 3 --steps 2 --warmup 1 --repeat 1 --no-cpu-baseline --also"
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $set | tr ' ' '_')
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/prof_${TAG}_$n -o pmc -- $P > $OUT/pmc_run_$TAG.log 2>&1 || echo "rc=$? for $set"
  python $ROOT/tools/rocpd_summary.py pmc $(find $OUT/prof_${TAG}_$n -name '*_results.db' | head -1) >> $OUT/pmc_gemm_$TAG.txt 2>&1
  rm -rf $OUT/prof_${TAG}_$n
done
grep -E "gemm|Kernel|kernel" $OUT/pmc_gemm_$TAG.txt | head -60
