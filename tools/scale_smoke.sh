#!/bin/bash
# Multi-GPU first contact for an N-GPU node (nothing here has run on more than one physical GPU: see BASELINE.md).  For N in 1 2 4 8
# (up to max_gpus): (a) the one-process-per-GPU path (torchrun + bert.cpp_amd/dist.py) and (b) the in-process path (libbert.so's own
# devices, a worker thread per device, (c) the RCCL exchange issued per device thread: ncclAllGather for the fixed-length batch, the
# grouped broadcast for the ragged one) on one fixed pair of batches; every arm prints how many ranks / devices it saw and must print
# the digests of N = 1 (per-sentence bits do not depend on the number of GPUs): a mismatch — or an arm that printed nothing — makes
# the script exit non-zero.  Then bench.py's scaling lines, both ways.
#   usage: bash tools/scale_smoke.sh [max_gpus]          STEPS=10 (bench steps per line)
#   BERT_BENCH_SHARED_GPU=1: validation on a box with fewer GPUs than ranks (the 1-GPU box: max_gpus 2) — the torchrun ranks share the
#   GPU and exchange over gloo; the in-process arms run for N <= GPUs present only (one engine per device, RCCL refuses two ranks on one).
set -u
export BERT_HIP_QUIET=1 HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
MAX=${1:-$HAVE}
STEPS=${STEPS:-10}
SHARED=${BERT_BENCH_SHARED_GPU:-0}
[ "$SHARED" = "0" ] && [ "$MAX" -gt "$HAVE" ] && MAX=$HAVE
REF=""
rc=0
for n in 1 2 4 8; do
  [ $n -gt $MAX ] && break
  a="(skipped: $HAVE GPU(s) present)"
  if [ $n -le $HAVE ]; then
    la=$(timeout 600 python tools/scale_smoke.py inproc $n | grep '^scale_smoke'); echo "$la"
    a=$(echo "$la" | sed 's/.*digests //')
    [ -z "$REF" ] && REF="$a"
    [ "$a" = "$REF" ] && [ -n "$a" ] || { echo "MISMATCH (inproc) at n=$n"; rc=1; }
  fi
  lb=$(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) tools/scale_smoke.py torchrun | grep '^scale_smoke'); echo "$lb"
  b=$(echo "$lb" | sed 's/.*digests //')
  [ -z "$REF" ] && REF="$b"
  echo "n=$n inproc [$a] torchrun [$b] reference [$REF]"
  [ "$b" = "$REF" ] && [ -n "$b" ] || { echo "MISMATCH (torchrun) at n=$n"; rc=1; }
done
for n in 1 2 4 8; do
  [ $n -gt $MAX ] && break
  if [ $n -le $HAVE ]; then
    timeout 900 python bench.py --gpus $n --inproc --steps $STEPS --warmup 3 --repeat 3 --no-cpu-baseline | cut -c1-400 || rc=1
  fi
  if [ $n -gt 1 ]; then
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps $STEPS --warmup 3 --no-cpu-baseline --also | cut -c1-2500 || rc=1
  fi
done
exit $rc
