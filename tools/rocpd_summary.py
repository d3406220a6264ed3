#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite output) into the small text tables kept under profiles/.

usage: rocpd_summary.py stats <results.db>            per-kernel count / total / avg / min / max duration
       rocpd_summary.py pmc   <results.db> [...]      per-kernel mean of every collected counter
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name).replace("bert_hip::", "").replace("void ", "")
    return name.replace(" [clone .kd]", "").replace(".kd", "")


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = defaultdict(list)
    for name, s, e in rows:
        agg[short(name)].append((e - s) / 1e3)
    total = sum(sum(v) for v in agg.values())
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[:70]:70s} {len(v):6d} {sum(v):12.1f} {sum(v)/len(v):10.2f} {min(v):10.2f} {max(v):10.2f} {100*sum(v)/total:6.2f}")


def pmc(paths):
    agg = defaultdict(lambda: defaultdict(list))
    for path in paths:
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        rows = db.execute("select * from counters_collection").fetchall()
        ci = {c: i for i, c in enumerate(cols)}
        kn = ci.get("kernel_name", ci.get("name"))
        for r in rows:
            agg[short(r[kn])][r[ci["counter_name"]]].append(float(r[ci["value"]]))
    counters = sorted({c for k in agg.values() for c in k})
    print(f"{'kernel':60s} {'n':>5s} " + " ".join(f"{c[-22:]:>22s}" for c in counters))
    for k, v in sorted(agg.items()):
        n = max(len(x) for x in v.values())
        print(f"{k[:60]:60s} {n:5d} " + " ".join(f"{(sum(v[c])/len(v[c]) if v.get(c) else float('nan')):22.1f}" for c in counters))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
