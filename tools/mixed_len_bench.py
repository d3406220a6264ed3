#!/usr/bin/env python3
"""Sentences/s on real-text-like length distributions (mean ~25 tokens, clipped to [3, 128]) through the host API
(bert_hip_eval_packed: pinned staging, H2D, forward, D2H, blocking), with the per-kernel breakdown of one pass.
usage: mixed_len_bench.py [n_sentences]   env BERT_HIP_KERNELS=tiled selects the GEMM + attention kernels for comparison"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np
from bert_cpp_amd import ggml_file as gf, pybert

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(5)
lens = np.clip(np.round(rng.lognormal(np.log(21.0), 0.55, B)), 3, 128).astype(np.int32)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.bin"); hp = gf.make_synthetic_model(p, "minilm-l6", "f16", seed=0)
    m = pybert.BertModel(p)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    flat = rng.integers(1000, hp.n_vocab, size=int(cu[-1])).astype(np.int32)
    for _ in range(2): out = m.eval_packed(flat, cu)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m.eval_packed(flat, cu); ts.append(time.perf_counter() - t0)
    m.profile(True); m.eval_packed(flat, cu); rep = m.profile_report(); m.profile(False)
    print(f"mixed lengths: B={B} mean_len={lens.mean():.1f} max={lens.max()} tokens={int(cu[-1])} KERNELS={os.environ.get('BERT_HIP_KERNELS', 'fused')}: "
          f"median {B / np.median(ts):,.0f} sent/s (min {B / max(ts):,.0f}, max {B / min(ts):,.0f}), {np.median(ts) * 1e3:.2f} ms")
    print("  kernels ms:", {k: round(v["total_ms"], 3) for k, v in sorted(rep.items())})
    print("  checksum", float(np.abs(out).sum()), "finite", bool(np.isfinite(out).all()))
