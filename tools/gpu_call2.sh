#!/bin/bash
# round 4, second GPU call: the profile's dispatch timestamps against rocprofv3's kernel trace of the same command
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp BERT_HIP_QUIET=1
timeout 300 python bench.py --also --no-cpu-baseline > $OUT/bench_c2.log 2> $OUT/bench_c2.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_c2.log") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"], d["kernel_ms_per_step"])
PY
tail -3 $OUT/bench_c2.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o stats -- python $OUT/../bench.py --also --no-cpu-baseline > $OUT/bench_c2_prof.log 2>&1; echo "prof rc=$?"
cd $OUT/..
python tools/rocpd_summary.py stats $(find $OUT/prof_c2 -name '*_results.db' | head -1) 2>&1 | head -12
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_c2_prof.log") if l.startswith("{")][-1])
print("under rocprof:", {k:d[k] for k in ("value","ms_per_step")}, {k:d["roofline"][k] for k in ("avg_launch_us","frac","avg_launch_us_timed_alone","step_share")})
PY
timeout 300 python bench.py --config 3 --also --no-cpu-baseline --steps 3 --warmup 1 --repeat 2 > $OUT/bench_c2_3.log 2>&1; python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_c2_3.log") if l.startswith("{")][-1])
print("config3:", {k:d[k] for k in ("value","ms_per_step")}, {k:d["roofline"][k] for k in ("kernel","avg_launch_us","frac","step_share")}, d["kernel_ms_per_step"])
PY
rm -rf $OUT/prof_c2
