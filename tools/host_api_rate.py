#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-shaped host API: bert_eval_batch from per-sentence host pointers
(pack, pinned staging, H2D, forward, D2H, scatter; blocking).  Reported in DESIGN.md, never as bench value."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np
from bert_cpp_amd import ggml_file as gf, pybert
for dims, ftype, B, N in (("minilm-l6", "f16", 256, 128), ("minilm-l6", "q4_0", 1024, 128), ("minilm-l6", "f16", 16384, 128)):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.bin"); hp = gf.make_synthetic_model(p, dims, ftype, seed=0)
        m = pybert.BertModel(p)
        ids = gf.synthetic_token_ids(B, N, hp.n_vocab, seed=7)
        cu = (np.arange(B + 1) * N).astype(np.int32)
        flat = ids.reshape(-1).copy()
        for _ in range(3): m.eval_packed(flat, cu)
        t0 = time.perf_counter(); K = 10 if B <= 1024 else 3
        for _ in range(K): m.eval_packed(flat, cu)
        dt = (time.perf_counter() - t0) / K
        sents = [ids[i] for i in range(B)]
        m.eval_batch(sents)
        t0 = time.perf_counter()
        for _ in range(3): m.eval_batch(sents)
        dt2 = (time.perf_counter() - t0) / 3
        if B * N > 262144:                 # more than one device chunk: the host path overlaps staging / unpacking with compute
            m.set_option("chunk_tokens", str(B * N))
            m.eval_packed(flat, cu)
            t0 = time.perf_counter()
            for _ in range(3): m.eval_packed(flat, cu)
            dt1 = (time.perf_counter() - t0) / 3
            print(f"  (the same call as ONE chunk, nothing to overlap: {B/dt1:,.0f} sent/s)")
        print(f"{dims} {ftype} B={B} N={N}: bert_hip_eval_packed (host) {B/dt:,.0f} sent/s, {dt*1e3:.2f} ms; "
              f"bert_eval_batch via ctypes pointers {B/dt2:,.0f} sent/s")
