#!/usr/bin/env python3
"""One sentence per call through bert_hip_eval_packed, in a loop (for rocprofv3 --kernel-trace --stats): tools/latency_probe.py [n_tokens] [calls]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np
from bert_cpp_amd import ggml_file as gf, pybert
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "m.bin")
    hp = gf.make_synthetic_model(path, "minilm-l6", "f16", seed=0)
    m = pybert.BertModel(path)
    ids = gf.synthetic_token_ids(1, n, hp.n_vocab, seed=77).reshape(-1)
    cu = np.array([0, n], dtype=np.int32)
    for _ in range(10): m.eval_packed(ids, cu)
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter(); m.eval_packed(ids, cu); ts.append(time.perf_counter() - t0)
    print(f"n={n}: median {1e6 * np.median(ts):.1f} us over {calls} calls")
