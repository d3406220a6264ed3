#!/usr/bin/env python3
"""Host-side rate of the WordPiece tokenizer behind bert_encode_batch (no GPU): texts/s and tokens/s of
bert_hip_tokenize_batch on 1..N host threads, on English-like text with a vocabulary built from the same text
(whole words for the frequent ones, pieces for the rest) — what the GPU path needs from the host to stay fed:
~1.1 M sentences/s of ~25 tokens.   usage: tokenizer_rate.py [n_texts] [text file]"""
import os, re, sys, tempfile, time, collections, random, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert_cpp_amd import pybert

n_texts = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
src = sys.argv[2] if len(sys.argv) > 2 else None
rng = random.Random(1)
if src and os.path.exists(src):
    lines = [l.strip() for l in open(src, encoding="utf-8", errors="ignore") if l.strip()]
else:
    syll = ["ta", "re", "mo", "in", "ul", "es", "ka", "do", "vi", "ne", "or", "shi", "pla", "con", "ter", "ing", "ed", "ly", "un", "pre"]
    words = ["".join(rng.choice(syll) for _ in range(rng.randint(1, 4))) for _ in range(6000)]
    lines = [" ".join(rng.choice(words) for _ in range(rng.randint(5, 40))) + rng.choice([".", "?", "!", ""]) for _ in range(3000)]
words = collections.Counter(w for l in lines for w in re.findall(r"[a-z0-9]+", l.lower()))
vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
vocab += [chr(c) for c in range(33, 127)] + ["##" + chr(c) for c in range(97, 123)] + ["##" + str(d) for d in range(10)]
vocab += [w for w, _ in words.most_common(12000)]
vocab += ["##" + s for s in ("s", "ed", "ing", "ly", "er", "es", "tion", "al", "ment", "ness")]
seen, v2 = set(), []
for t in vocab:
    if t not in seen:
        seen.add(t); v2.append(t)
vocab = v2
texts = [lines[i % len(lines)] for i in range(n_texts)]
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "vocab_only.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<i", 0x67676d6c))
        f.write(struct.pack("<7i", len(vocab), 512, 384, 1536, 12, 6, 1))
        for tok in vocab:
            b = tok.encode("utf-8")
            f.write(struct.pack("<i", len(b))); f.write(b)
    tk = pybert.BertTokenizer(path) if hasattr(pybert, "BertTokenizer") else pybert.BertModel(path, tokenizer_only=True)
    enc = [t.encode("utf-8") for t in texts]
    import ctypes as C
    import numpy as np
    arr = (C.c_char_p * n_texts)(*enc)
    toks = np.zeros((n_texts, tk.n_max_tokens), dtype=np.int32)
    cnt = np.zeros(n_texts, dtype=np.int32)
    i32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    tk.lib.bert_hip_tokenize_batch(tk.ctx, 8, n_texts, arr, i32p(toks), i32p(cnt))      # (first touch of the output pages)
    for nt in (1, 2, 4, 8, 16):
        t0 = time.perf_counter()
        rc = tk.lib.bert_hip_tokenize_batch(tk.ctx, nt, n_texts, arr, i32p(toks), i32p(cnt))
        dt = time.perf_counter() - t0
        assert rc == 0
        ntok = int(cnt.sum())
        print(f"{nt:2d} threads: {n_texts / dt:12,.0f} texts/s  {ntok / dt:14,.0f} tokens/s  (mean {ntok / n_texts:.1f} tokens per text; the C call alone)")
