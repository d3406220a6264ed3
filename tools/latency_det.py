#!/usr/bin/env python3
# determinism probe of the latency route: the same one-sentence call repeated, against the batch route
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
    for n in (128, 25, 64, 33):
        ids = rng.integers(1000, hp.n_vocab, size=n).astype(np.int32)
        other = rng.integers(1000, hp.n_vocab, size=128).astype(np.int32)
        ref = m.eval_batch([ids, other, other])[0]
        runs = [m.eval_batch([ids])[0] for _ in range(6)]
        print(n, "distinct results over 6 runs:", len({r.tobytes() for r in runs}), "max diff vs batch per run:", [float(np.abs(r - ref).max()) for r in runs])
