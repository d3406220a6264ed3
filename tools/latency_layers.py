#!/usr/bin/env python3
# latency route vs batch route: per-kernel times of one-sentence calls, embedding and per-layer hidden-state differences
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np
from bert_cpp_amd import ggml_file as gf, pybert
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "m.bin")
    hp = gf.make_synthetic_model(path, "minilm-l6", "f16", seed=0)
    m = pybert.BertModel(path)
    rng = np.random.default_rng(0)
    for n in (128, 25, 64):
        ids = rng.integers(1000, hp.n_vocab, size=n).astype(np.int32)
        other = rng.integers(1000, hp.n_vocab, size=128).astype(np.int32)
        for _ in range(5): m.eval_batch([ids])
        m.profile(True)
        for _ in range(5): m.eval_batch([ids])
        rep = m.profile_report(); m.profile(False)
        print(n, {k: round(1e3 * v["total_ms"] / v["launches"], 1) for k, v in sorted(rep.items())})
        e_alone, h_alone = m.eval_hidden(ids)
        batch = m.eval_batch([ids, other, other])
        alone = m.eval_batch([ids])[0]
        print("  alone vs batch max diff", float(np.abs(alone - batch[0]).max()), "hidden shape", h_alone.shape)
        m.set_option("latency", "0")
        e_b, h_b = m.eval_hidden(ids)
        m.set_option("latency", "1")
        L = hp.n_layer + 1
        ha, hb = np.asarray(h_alone).reshape(L, n, -1), np.asarray(h_b).reshape(L, n, -1)
        print("   per-layer max |hidden diff| latency vs fused:", [float(np.abs(ha[l] - hb[l]).max()) for l in range(L)])
