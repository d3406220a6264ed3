for rep in $(seq ${REPS:-2}); do for v in "$@"; do
  lib=bert.cpp_amd/libbert${v}.so
  BERT_HIP_LIB=$PWD/$lib timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), d.get('kernel_ms_per_step'))"
done; done
