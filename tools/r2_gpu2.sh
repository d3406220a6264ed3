#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_tl.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also > $OUT/r2b_tl.json 2> $OUT/r2b_tl.err; echo "tl rc=$?"
grep rawtimeline $OUT/r2b_tl.err | head -4
cd /tmp
PMC_BENCH="python $OUT/../bench.py --steps 2 --warmup 1 --no-cpu-baseline --also"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
  "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/prof_pmc${i}_r2b -o pmc -- $PMC_BENCH > $OUT/r2b_pmc${i}.log 2>&1; echo "pmc$i rc=$?"
done
cd $OUT/..
for j in 1 2 3; do python tools/rocpd_summary.py pmc $(find $OUT/prof_pmc${j}_r2b -name '*_results.db' | head -1) > $OUT/r2b_pmc${j}.txt 2>&1; done
rm -rf $OUT/prof_pmc*_r2b
cat $OUT/r2b_pmc1.txt $OUT/r2b_pmc2.txt $OUT/r2b_pmc3.txt | head -60
