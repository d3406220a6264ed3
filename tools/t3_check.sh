#!/bin/bash
# layer_tail3 tuning: parity against layer_tail, A/B times, ablation variants (tools/variant.sh t3aN layer_tail3.hip -DT3_ABLATE=N),
# in-kernel timeline
export TMPDIR=/tmp BERT_HIP_QUIET=1
OUT=$PWD/gpurun_out; mkdir -p $OUT; : > $OUT/t3_abl.log
timeout 300 python - > $OUT/t3_parity.log 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from bert_cpp_amd import pybert
def run(M,H,I,impl):
    rng = np.random.default_rng(M + H + I)
    ctx = rng.normal(0, 1, (M, H)).astype(np.float16); x = rng.normal(0, 1, (M, H)).astype(np.float16)
    Wo = (rng.normal(0, 1, (H, H)) / np.sqrt(H)).astype(np.float16)
    W1 = (rng.normal(0, 1, (I, H)) / np.sqrt(H)).astype(np.float16)
    W2 = (rng.normal(0, 1, (H, I)) / np.sqrt(I)).astype(np.float16)
    bo, b2 = rng.normal(0, 0.2, H), rng.normal(0, 0.2, H); b1 = rng.normal(0, 0.5, I)
    g1, g2 = 1 + rng.normal(0, 0.1, H), 1 + rng.normal(0, 0.1, H); be1, be2 = rng.normal(0, 0.1, H), rng.normal(0, 0.1, H)
    args = (ctx, x, Wo.view(np.uint8), W1.view(np.uint8), W2.view(np.uint8), 1, I, bo, g1, be1, b1, b2, g2, be2)
    return pybert.test_layer_tail(*args, impl).astype(np.float64)
for (M,H,I) in [(128,384,256),(130,384,1536),(256,256,512),(1000,256,1024),(32768,384,1536)]:
    a = run(M,H,I,1); b = run(M,H,I,4); b2 = run(M,H,I,4)
    d = np.abs(a-b)
    print(M,H,I,'max|t1-t3|',d.max(),'mean',d.mean(),'nan',np.isnan(b).sum(),'repeatable',bool((b==b2).all()), flush=True)
PY
cat $OUT/t3_parity.log
for v in ${VARIANTS:-"1:" "3:" "1:" "3:"}; do
  tail=${v%%:*}; lib=${v#*:}
  echo "TAIL=$tail lib='$lib'" >> $OUT/t3_abl.log
  BERT_HIP_TAIL=$tail BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert$lib.so STEPS=100 REPEAT=2 timeout 200 python tools/kernel_times.py 1 2>&1 | tail -1 >> $OUT/t3_abl.log
done
[ -n "$NO_TL" ] || BERT_HIP_TAIL=3 BERT_HIP_LIB=$PWD/bert.cpp_amd/libbert_tl.so STEPS=20 REPEAT=1 timeout 200 python tools/kernel_times.py 1 > $OUT/t3_tl.log 2>&1
cat $OUT/t3_abl.log; grep "^t3 wg" $OUT/t3_tl.log | cut -c1-6000
