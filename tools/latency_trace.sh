export TMPDIR=/tmp BERT_HIP_QUIET=1
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 120 python tools/probe.py latency 128 300
timeout 120 python tools/probe.py latency 25 300
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_lat -o lat -- python $OUT/../tools/probe.py latency 128 200 > $OUT/lat_prof.log 2>&1
cd $OUT/..; DB=$(find $OUT/prof_lat -name '*_results.db' | head -1)
python tools/rocpd_summary.py stats $DB | head -20
python tools/rocpd_summary.py gaps $DB | head -20
python - <<'PY'
import sqlite3,glob,os
db=sqlite3.connect(glob.glob(os.environ.get('OUT','gpurun_out')+'/prof_lat/**/*_results.db',recursive=True)[0])
rows=sorted(db.execute("select name,start,end from kernels").fetchall(), key=lambda r:r[1])
# one forward pass near the end: find last embed_ln
idx=[i for i,r in enumerate(rows) if 'embed_ln' in r[0]]
i0=idx[-5]; i1=idx[-4]
t0=rows[i0][1]
for r in rows[i0:i1]:
    print(f"{(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:7.1f}  {r[0][:60]}")
print("pass span us", (rows[i1-1][2]-t0)/1e3, "next pass starts after", (rows[i1][1]-rows[i1-1][2])/1e3)
PY
rm -rf $OUT/prof_lat
