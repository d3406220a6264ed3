#!/bin/bash
# PMC passes for the layer tail kernels (BERT_HIP_TAIL=1 / 3): each counter set in its own run, kernel trace only
export TMPDIR=/tmp BERT_HIP_QUIET=1
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_]*" | sort -u > $OUT/counters.txt
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for tail in 1 3; do
    BERT_HIP_TAIL=$tail timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/t3pmc${i}_$tail -o pmc -- python $OUT/../bench.py --steps 3 --warmup 2 --repeat 1 --no-cpu-baseline > $OUT/t3pmc${i}_$tail.log 2>&1
    python $OUT/../tools/rocpd_summary.py pmc $(find $OUT/t3pmc${i}_$tail -name '*_results.db' | head -1) 2>&1 | grep -i "kernel \|layer_tail" | cut -c1-400 >> $OUT/t3_pmc.txt
    rm -rf $OUT/t3pmc${i}_$tail
  done
done
cat $OUT/t3_pmc.txt; grep -c . $OUT/counters.txt; grep -i "ifetch\|INST_CACHE\|ICACHE" $OUT/counters.txt | head
