#!/usr/bin/env python3
"""Per-kernel HBM traffic per launch from rocprofv3 PMC passes -> profiles/traffic.json (read by bench.py).

usage: pmc_traffic.py config1=<fetch.db>,<write.db> [config2=...] > profiles/traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section) and is doubled here.  Kernel names are mapped to the engine's profile names.
"""
import json
import sqlite3
import sys
from collections import defaultdict

NAMES = {"qkv_attention2_kernel": "qkv_attention2", "layer_tail_kernel": "layer_tail", "model_kernel": "model_kernel", "attention_mfma_kernel": "attention",
         "embed_ln_kernel": "embed_ln", "embed_ln_rows_kernel": "embed_ln", "pool_normalize_kernel": "pool_normalize",
         "layernorm_rows_kernel": "layernorm",
         # gemm256_kernel<EPI>: 0 = bias (Q|K|V), 1 = bias + GELU (FFN up), 2 = bias + residual (attention output AND FFN down:
         # one kernel, two shapes; their mean is booked under gemm_ffn_down)
         # (second template argument: the weight form — 0 f16 image, 1 q4_0 planes, 2 q4_1 planes)
         "gemm256_kernel<0,": "gemm_qkv", "gemm256_kernel<1,": "gemm_ffn_up", "gemm256_kernel<2,": "gemm_ffn_down"}


def counter_means(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ci = {c: i for i, c in enumerate(cols)}
    kn = ci.get("kernel_name", ci.get("name"))
    agg = defaultdict(list)
    for r in db.execute("select * from counters_collection"):
        if r[ci["counter_name"]] == counter:
            agg[r[kn]].append(float(r[ci["value"]]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    out = {}
    for arg in sys.argv[1:]:
        cfg, paths = arg.split("=")
        fetch_db, write_db = paths.split(",")
        fetch, write = counter_means(fetch_db, "FETCH_SIZE"), counter_means(write_db, "WRITE_SIZE")
        d = {}
        for k, f in fetch.items():
            short = next((v for n, v in NAMES.items() if n in k), None)
            if short is None:
                continue
            d[short] = {"bytes": round(2 * f * 1024 + write.get(k, 0.0) * 1024), "fetch_bytes_x2": round(2 * f * 1024),
                        "write_bytes": round(write.get(k, 0.0) * 1024)}
        out[cfg] = d
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
