#!/bin/bash
# Multi-GPU smoke for an N-GPU node (nothing here has run on more than one physical GPU: see BASELINE.md).  For N in 1 2 4 8
# (up to the GPUs present): the in-process path (libbert.so's own devices, worker threads, RCCL exchange issued per device
# thread) and the one-process-per-GPU path (torchrun + dist.py), each on a fixed batch; every arm must print the digests of
# N = 1 (per-sentence bits do not depend on the number of GPUs).  Then the bench's scaling lines, both ways.
#   usage: bash tools/scale_smoke.sh [max_gpus]
set -u
export BERT_HIP_QUIET=1 HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
MAX=${1:-$HAVE}
REF=""
rc=0
for n in 1 2 4 8; do
  [ $n -gt $MAX ] && break
  a=$(timeout 600 python tools/scale_smoke.py inproc $n | grep '^scale_smoke' | sed 's/.*digests //')
  b=$(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) tools/scale_smoke.py torchrun | grep '^scale_smoke' | sed 's/.*digests //')
  [ -z "$REF" ] && REF="$a"
  echo "n=$n inproc [$a] torchrun [$b] reference [$REF]"
  [ "$a" = "$REF" ] && [ "$b" = "$REF" ] && [ -n "$REF" ] || { echo "MISMATCH at n=$n"; rc=1; }
done
for n in 1 2 4 8; do
  [ $n -gt $MAX ] && break
  timeout 900 python bench.py --gpus $n --inproc --steps 10 --warmup 3 --repeat 3 --no-cpu-baseline | cut -c1-400
  if [ $n -gt 1 ]; then
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --also | cut -c1-400
  fi
done
exit $rc
