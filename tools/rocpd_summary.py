#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite output) into the small text tables kept under profiles/.

usage: rocpd_summary.py stats <results.db>            per-kernel count / total / avg / min / max duration
       rocpd_summary.py pmc   <results.db> [...]      per-kernel mean of every collected counter
       rocpd_summary.py gaps  <results.db>            idle time between consecutive kernels of one forward pass
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


def gaps(path, same_pass_us=50.0):
    """Time the GPU sits between the end of one kernel and the start of the next inside a forward pass (pairs further apart
    than `same_pass_us` are host-side pauses between timed regions, not launch boundaries)."""
    db = sqlite3.connect(path)
    # (runtime copy / fill kernels belong to the benchmark's set-up, not to the forward pass)
    rows = sorted((r for r in db.execute("select name, start, end from kernels").fetchall() if "__amd_rocclr" not in r[0]), key=lambda r: r[1])
    busy = gap = 0.0
    n = 0
    per = defaultdict(list)
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = (s1 - e0) / 1e3
        if g > same_pass_us:
            continue
        busy += (e0 - s0) / 1e3
        gap += max(g, 0.0)
        n += 1
        per[f"{short(n0)[:34]} -> {short(n1)[:34]}"].append(g)
    print(f"{n} kernel boundaries inside forward passes: kernels {busy:.1f} us, idle between them {gap:.1f} us = {100 * gap / (busy + gap):.2f} % of the pass")
    print(f"{'boundary':72s} {'n':>6s} {'avg_gap_us':>11s} {'max_gap_us':>11s}")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print(f"{k:72s} {len(v):6d} {sum(v) / len(v):11.2f} {max(v):11.2f}")


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
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2])
    else:
        pmc(sys.argv[2:])
