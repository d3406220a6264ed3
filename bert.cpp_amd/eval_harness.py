#!/usr/bin/env python3
"""Offline accuracy harness in the style of the reference's benchmarks/run_mteb.py:29-95.

The reference scores its models with two MTEB tasks — STSBenchmark (Spearman correlation between the cosine
similarity of sentence-pair embeddings and human scores) and EmotionClassification (accuracy of a logistic
regression trained on the embeddings) — downloaded by the `mteb` package.  This container has neither network
nor `mteb`; the harness computes the same two scores from local files so that the published table
(reference README.md:147-178) can be reproduced as soon as real weights and the datasets are supplied:

    eval_harness.py --model ggml-model-q4_0.bin --sts sts-test.tsv            # lines: score<TAB>sentence1<TAB>sentence2
    eval_harness.py --model ... --cls-train train.tsv --cls-test test.tsv     # lines: label<TAB>text

Embeddings come from libbert.so through bert_encode_batch, exactly as the reference's ctypes BertModel does.
Prints one JSON object per task with MTEB's field names (cos_sim.spearman / accuracy, evaluation_time).
"""
import argparse
import json
import time
from typing import Callable, List, Sequence, Tuple

import numpy as np


def spearman(a: Sequence[float], b: Sequence[float]) -> float:
    """Spearman rank correlation (average ranks for ties), as scipy.stats.spearmanr."""
    def ranks(v):
        v = np.asarray(v, dtype=np.float64)
        order = np.argsort(v, kind="mergesort")
        r = np.empty(len(v), dtype=np.float64)
        sv = v[order]
        i = 0
        while i < len(v):
            j = i
            while j + 1 < len(v) and sv[j + 1] == sv[i]:
                j += 1
            r[order[i:j + 1]] = 0.5 * (i + j) + 1.0
            i = j + 1
        return r
    ra, rb = ranks(a), ranks(b)
    ra -= ra.mean(); rb -= rb.mean()
    den = np.sqrt((ra * ra).sum() * (rb * rb).sum())
    return float((ra * rb).sum() / den) if den > 0 else 0.0


def cosine_rows(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    num = (a * b).sum(axis=1)
    return num / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def sts_task(encode: Callable[[List[str]], np.ndarray], pairs: Sequence[Tuple[float, str, str]]) -> dict:
    """STSBenchmark main score: Spearman of cosine similarity vs gold scores."""
    t0 = time.perf_counter()
    s1 = encode([p[1] for p in pairs])
    s2 = encode([p[2] for p in pairs])
    sims = cosine_rows(s1, s2)
    return {"cos_sim": {"spearman": spearman(sims, [p[0] for p in pairs])}, "n_pairs": len(pairs),
            "evaluation_time": time.perf_counter() - t0}


def classification_task(encode: Callable[[List[str]], np.ndarray], train: Sequence[Tuple[str, str]],
                        test: Sequence[Tuple[str, str]], max_iter: int = 100) -> dict:
    """EmotionClassification-style score: accuracy of a logistic regression on the embeddings."""
    from sklearn.linear_model import LogisticRegression

    t0 = time.perf_counter()
    xtr, xte = encode([t for _, t in train]), encode([t for _, t in test])
    clf = LogisticRegression(max_iter=max_iter)
    clf.fit(xtr, [l for l, _ in train])
    acc = float((clf.predict(xte) == np.array([l for l, _ in test])).mean())
    return {"accuracy": acc, "n_train": len(train), "n_test": len(test), "evaluation_time": time.perf_counter() - t0}


def read_tsv(path: str, n_cols: int) -> List[Tuple[str, ...]]:
    rows = []
    with open(path, encoding="utf-8") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) >= n_cols:
                rows.append(tuple(parts[:n_cols]))
    return rows


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", required=True)
    ap.add_argument("--sts")
    ap.add_argument("--cls-train")
    ap.add_argument("--cls-test")
    ap.add_argument("--batch-size", type=int, default=64)
    a = ap.parse_args()
    from bert_cpp_amd import pybert

    m = pybert.BertModel(a.model)
    encode = lambda texts: m.encode_batch(texts, batch_size=a.batch_size)
    if a.sts:
        pairs = [(float(s), x, y) for s, x, y in read_tsv(a.sts, 3)]
        print(json.dumps({"task": "STS", **sts_task(encode, pairs)}))
    if a.cls_train and a.cls_test:
        print(json.dumps({"task": "Classification", **classification_task(encode, read_tsv(a.cls_train, 2), read_tsv(a.cls_test, 2))}))


if __name__ == "__main__":
    main()
