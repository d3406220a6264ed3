#!/usr/bin/env python3
# host-to-host rate of bert_hip_eval_packed on the bench's mixed-length batch against the chunk size of the pinned two-slot pipeline
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np
import bench
from bert_cpp_amd import ggml_file as gf, pybert
cfg = bench.CONFIGS[5]
hp = gf.MODEL_DIMS[cfg["dims"]]
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.bin"); gf.make_synthetic_model(p, cfg["dims"], cfg["ftype"], seed=0)
    m = pybert.BertModel(p)
    flat, cu, max_len = bench.config_inputs(cfg, 5, hp, 0)
    B, T = len(cu) - 1, int(cu[-1])
    for chunk in (32768, 65536, 131072, 262144, 524288, 65536, 262144):
        m.set_option("chunk_tokens", str(chunk))
        for _ in range(3): m.eval_packed(flat, cu)
        t0 = time.perf_counter()
        for _ in range(12): m.eval_packed(flat, cu)
        dt = time.perf_counter() - t0
        print("chunk_tokens", chunk, f"{B * 12 / dt:,.0f} sent/s  {dt / 12 * 1e3:.2f} ms", flush=True)
    m.close()
