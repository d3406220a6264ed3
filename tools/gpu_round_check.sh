#!/bin/bash
# One gpurun call that validates HEAD on the MI355X box and collects everything profiles/ is built from:
#   pytest -m gpu, smoke(), the default bench line, rocprofv3 kernel stats of the bench command (configs 1 and 3), SQ counter
#   sets for config 1, FETCH_SIZE / WRITE_SIZE passes for configs 1, 2 (as written and expanded), 3, 4 (both also with 4-bit-resident
#   weights: 33, 42), 5 (each counter set in its own run, never mixed with trace domains other than --kernel-trace), the vendor GEMM
#   calibration.
# usage (from the repo root on the GPU box):  bash tools/gpu_round_check.sh [tag]      then, here: python tools/assemble_profiles.py [tag]
set -u
trap '' PIPE        # (a reader that stops early — `| head` — must not end the run half way)
TAG=${1:-r6}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_$TAG.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_$TAG.log 2> $OUT/bench_$TAG.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s] line bytes $(tail -1 $OUT/bench_$TAG.log | wc -c)"; cut -c1-600 $OUT/bench_$TAG.log
cp bench_detail.json $OUT/bench_detail_$TAG.json
# the driver's own invocation (its clock around the run is what BENCH_rNN.json records)
t0=$(date +%s)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_$TAG.log 2> $OUT/bench_driver_$TAG.err; echo "driver-style bench rc=$? [$(( $(date +%s) - t0 )) s] line bytes $(tail -1 $OUT/bench_driver_$TAG.log | wc -c)"

cd /tmp
# (100-step regions like the un-profiled line: a region's fixed costs — first launch, final synchronisation, the profiler's flush —
# spread over 10 steps would inflate ms_per_step, which the in-place kernel's roofline time is derived from)
B1="python $OUT/../bench.py --steps 100 --warmup 10 --repeat 2 --no-cpu-baseline --also"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$TAG -o stats -- $B1 > $OUT/bench_prof_$TAG.log 2>&1; echo "stats rc=$?"
B3="python $OUT/../bench.py --config 3 --steps 3 --warmup 1 --repeat 1 --no-cpu-baseline --also"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats3_$TAG -o stats -- $B3 > $OUT/bench_prof3_$TAG.log 2>&1; echo "stats3 rc=$?"
B33="python $OUT/../bench.py --config 33 --steps 3 --warmup 1 --repeat 1 --no-cpu-baseline --also"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats33_$TAG -o stats -- $B33 > $OUT/bench_prof33_$TAG.log 2>&1; echo "stats33 rc=$?"
P1="python $OUT/../bench.py --steps 2 --warmup 1 --repeat 1 --no-cpu-baseline --also"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/prof_pmc${i}_$TAG -o pmc -- $P1 > $OUT/bench_pmc${i}_$TAG.log 2>&1; echo "pmc$i rc=$?"
done
TR=""
for cfg in 1 2 22 3 33 4 42 5; do
  key=config$cfg; [ $cfg = 22 ] && key=config2_expanded; [ $cfg = 5 ] && key=mixed_len; [ $cfg = 33 ] && key=config3_fused; [ $cfg = 42 ] && key=config4_fused
  Pc="python $OUT/../bench.py --config $cfg --steps 2 --warmup 1 --repeat 1 --no-cpu-baseline --also"
  timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof_fetch${cfg}_$TAG -o pmc -- $Pc > $OUT/bench_fetch${cfg}_$TAG.log 2>&1; echo "fetch$cfg rc=$?"
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write${cfg}_$TAG -o pmc -- $Pc > $OUT/bench_write${cfg}_$TAG.log 2>&1; echo "write$cfg rc=$?"
  TR="$TR $key=$(find $OUT/prof_fetch${cfg}_$TAG -name '*_results.db' | head -1),$(find $OUT/prof_write${cfg}_$TAG -name '*_results.db' | head -1)"
done
cd $OUT/..
python tools/rocpd_summary.py stats $(find $OUT/prof_stats_$TAG -name '*_results.db' | head -1) > $OUT/stats_$TAG.txt 2>&1
python tools/rocpd_summary.py gaps $(find $OUT/prof_stats_$TAG -name '*_results.db' | head -1) >> $OUT/stats_$TAG.txt 2>&1
python tools/rocpd_summary.py stats $(find $OUT/prof_stats3_$TAG -name '*_results.db' | head -1) > $OUT/stats_config3_$TAG.txt 2>&1
python tools/rocpd_summary.py stats $(find $OUT/prof_stats33_$TAG -name '*_results.db' | head -1) > $OUT/stats_config33_$TAG.txt 2>&1
for j in 1 2; do python tools/rocpd_summary.py pmc $(find $OUT/prof_pmc${j}_$TAG -name '*_results.db' | head -1) > $OUT/pmc${j}_$TAG.txt 2>&1; done
python tools/rocpd_summary.py pmc $(find $OUT/prof_fetch1_$TAG $OUT/prof_write1_$TAG -name '*_results.db') > $OUT/pmc3_$TAG.txt 2>&1
python tools/rocpd_summary.py pmc $(find $OUT/prof_fetch3_$TAG $OUT/prof_write3_$TAG -name '*_results.db') > $OUT/pmc3c3_$TAG.txt 2>&1
python tools/pmc_traffic.py $TR > $OUT/traffic_$TAG.json 2>&1
grep -h '^{' $OUT/bench_prof_$TAG.log | tail -1 > $OUT/bench_prof_line_$TAG.txt
rm -rf $OUT/prof_stats_$TAG $OUT/prof_stats3_$TAG $OUT/prof_stats33_$TAG $OUT/prof_pmc*_$TAG $OUT/prof_fetch*_$TAG $OUT/prof_write*_$TAG
# the attention kernel's phase clock (tuning build with -DBERT_HIP_TIMELINE, built by `make timeline`), instruction costs, the
# shift experiment of the P.V accumulation
if [ -f bert.cpp_amd/libbert_tl.so ]; then BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_tl.so timeout 200 python tools/probe.py kernels 3 2>&1 | grep -E "attphase|attention phase|cfg3" > $OUT/att_phase_$TAG.txt; fi
[ -x tools/ubench/valu_cost ] && timeout 120 tools/ubench/valu_cost > $OUT/valu_cost_$TAG.txt 2>&1
[ -x tools/ubench/mfma_shift ] && timeout 60 tools/ubench/mfma_shift > $OUT/mfma_shift_$TAG.txt 2>&1
# calibration lines: the vendor GEMM library on the same shapes and box
timeout 120 python tools/gemm_calibration.py > $OUT/gemm_calibration_$TAG.txt 2>&1; echo "calibration rc=$?"
head -14 $OUT/stats_$TAG.txt
