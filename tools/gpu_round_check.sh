#!/bin/bash
# One gpurun call that validates HEAD on the MI355X box and collects everything profiles/ is built from:
#   pytest -m gpu, smoke(), the default bench line, rocprofv3 kernel stats of the bench command, and the
#   PMC passes (each counter set in its own run, never mixed with trace domains other than --kernel-trace).
# usage (from the repo root on the GPU box):  bash tools/gpu_round_check.sh [tag]
set -u
TAG=${1:-r1}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
BENCH="python $PWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --also"

timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > $OUT/bench_$TAG.log 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench_$TAG.log

cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$TAG -o stats -- $BENCH > $OUT/bench_prof_$TAG.log 2>&1; echo "stats rc=$?"
PMC_BENCH="python $OUT/../bench.py --steps 2 --warmup 1 --no-cpu-baseline --also"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/prof_pmc${i}_$TAG -o pmc -- $PMC_BENCH > $OUT/bench_pmc${i}_$TAG.log 2>&1; echo "pmc$i rc=$?"
done
cd $OUT/..
DB=$(find $OUT/prof_stats_$TAG -name '*_results.db' | head -1)
python tools/rocpd_summary.py stats $DB > $OUT/stats_$TAG.txt 2>&1
for j in 1 2; do python tools/rocpd_summary.py pmc $(find $OUT/prof_pmc${j}_$TAG -name '*_results.db' | head -1) > $OUT/pmc${j}_$TAG.txt 2>&1; done
python tools/rocpd_summary.py pmc $(find $OUT/prof_pmc3_$TAG $OUT/prof_pmc4_$TAG $OUT/prof_pmc5_$TAG -name '*_results.db') > $OUT/pmc3_$TAG.txt 2>&1
python tools/pmc_traffic.py config1=$(find $OUT/prof_pmc3_$TAG -name '*_results.db' | head -1),$(find $OUT/prof_pmc4_$TAG -name '*_results.db' | head -1) > $OUT/traffic_$TAG.json 2>&1
grep -h '^{' $OUT/bench_prof_$TAG.log | tail -2 > $OUT/bench_prof_line_$TAG.txt
# raw databases are large; keep only the text summaries in gpurun_out
rm -rf $OUT/prof_stats_$TAG $OUT/prof_pmc*_$TAG
head -12 $OUT/stats_$TAG.txt
