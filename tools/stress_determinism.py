#!/usr/bin/env python3
"""Stress test: a sentence's embedding must have the same bits whatever else is in the batch and however often
the launch sequence is repeated.  usage: stress_determinism.py [dims] [ftype] [iterations]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BERT_HIP_QUIET", "1")
from bert_cpp_amd import ggml_file as gf  # noqa: E402
from bert_cpp_amd import pybert  # noqa: E402


def main():
    dims = sys.argv[1] if len(sys.argv) > 1 else "tiny-h128"
    ftype = sys.argv[2] if len(sys.argv) > 2 else "q4_0"
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    hp = gf.MODEL_DIMS[dims]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.bin")
        gf.make_synthetic_model(path, dims, ftype, seed=11)
        m = pybert.BertModel(path)
        rng = np.random.default_rng(0)
        lens = [1, 2, 3, 4, 5, 7, 9, 12, 17, 25, 33, 48, 63, 64][: 14]
        lens = [min(n, hp.n_max_tokens) for n in lens]
        sents = [rng.integers(0, hp.n_vocab, size=n).astype(np.int32) for n in lens]
        ref = [m.eval(s).copy() for s in sents]
        bad = 0
        for it in range(iters):
            k = int(rng.integers(1, 6))
            idx = rng.choice(len(sents), size=k, replace=True)
            out = m.eval_batch([sents[i] for i in idx])
            for j, i in enumerate(idx):
                if not np.array_equal(out[j], ref[i]):
                    bad += 1
                    if bad <= 10:
                        print(f"iter {it}: sentence {i} (len {lens[i]}) in batch {list(idx)} differs, max abs {np.abs(out[j] - ref[i]).max():.3e}, "
                              f"nan={np.isnan(out[j]).any()}")
        print(f"{dims} {ftype}: {iters} batches, {bad} mismatches")
        return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
