#!/bin/bash
# SQ counter sets of one bench config (each set in its own rocprofv3 run, kernel trace only): tools/pmc_config.sh <config> [tag]
set -u
CFG=${1:-3}; TAG=${2:-c$CFG}
export TMPDIR=/tmp BERT_HIP_QUIET=1
OUT=$PWD/gpurun_out; mkdir -p $OUT
P="python $OUT/../bench.py --config $CFG --steps 2 --warmup 1 --repeat 1 --no-cpu-baseline --also"
cd /tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM" \
  "GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/prof_pmc${i}_$TAG -o pmc -- $P > $OUT/bench_pmc${i}_$TAG.log 2>&1; echo "pmc$i rc=$?"
done
cd $OUT/..
for j in 1 2 3; do python tools/rocpd_summary.py pmc $(find $OUT/prof_pmc${j}_$TAG -name '*_results.db' | head -1) > $OUT/pmc${j}_$TAG.txt 2>&1; done
rm -rf $OUT/prof_pmc*_$TAG
cat $OUT/pmc1_$TAG.txt $OUT/pmc2_$TAG.txt $OUT/pmc3_$TAG.txt | cut -c1-260
