#!/bin/bash
# tools/encode_rate.sh LIB [LIB ...]: strings in, embeddings out (bench.py's encode_batch_text entry) and the host-to-host rate on
# mixed-length batches (config 55), once per library given (paths relative to the repo root), alternating — for A/B runs of the
# host side in one gpurun call (libraries from tools/variant.sh).  A library built with -DBERT_HIP_HOST_TRACE (tools/variant.sh trace
# "-DBERT_HIP_HOST_TRACE") prints what bert_encode_batch and the host path spend where (per group: tokenizer, staging, queueing, the wait
# for the GPU, the rows' copy-out); the last call's lines are shown.
cd "$(dirname "$0")/.."
export BERT_HIP_QUIET=1
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    BERT_HIP_LIB=$PWD/$lib python - <<'PY' 2>&1 | grep -v "^bert_load\|^$" | tail -40
import tempfile, bench
r = bench.encode_batch_rate(tempfile.mkdtemp())
print("encode_batch_text %.0f texts/s  %.2f ms per call of %d texts, row 0 equals bert_encode: %s" % (r["value"], r["ms_per_call"], r["n_texts"], r.get("row0_equals_bert_encode")))
PY
    BERT_HIP_LIB=$PWD/$lib python bench.py --config 55 --steps 10 --warmup 3 --repeat 3 --no-cpu-baseline --also 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mixed_len_host_api %.0f sentences/s  %.3f ms per step' % (l['value'], l['ms_per_step']))"
  done
done
