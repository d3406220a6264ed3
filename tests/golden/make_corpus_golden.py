#!/usr/bin/env python3
"""Writes tests/golden/sample_texts_golden.json: the first 200 lines of the reference's sample texts (a DATA file of the reference,
examples/sample_client_texts.txt — what its sample client and server embed) with the token ids the ORACLE's line-for-line restatement
of bert_tokenize (oracle/bert_oracle.cpp, reference bert.cpp:199-325) gives them on a small English vocabulary written here.  Run in
the container that has /root/reference; the fixture travels to the GPU box, where the reference tree does not exist.  The product's
tokenizer is tested against these ids (tests/test_host.py, tests/test_gpu_exactness.py); nothing in the product reads the fixture."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import write_vocab_only_model  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SRC = "/root/reference/examples/sample_client_texts.txt"
N_LINES = 200
N_MAX_TOKENS = 64


def corpus_vocab(lines):
    """[PAD] .. [MASK] at BERT's ids, single characters, then the corpus' own frequent words and a few continuation pieces: enough for
    whole words, multi-piece splits, punctuation and unmatched bytes to all occur."""
    import collections
    import re
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab += list("abcdefghijklmnopqrstuvwxyz0123456789") + list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")
    words = collections.Counter(w for line in lines for w in re.findall(r"[a-z]+", line.decode("utf-8", "replace").lower()))
    vocab += [w for w, c in words.most_common(400) if len(w) > 1 and c >= 2]
    vocab += ["##s", "##ing", "##ed", "##ly", "##er", "##tion", "##ment", "##al", "##e", "##a", "##i", "##o", "##n", "##t", "##y", "##r", "##d"]
    seen, out = set(), []
    for v in vocab:
        if v not in seen:
            seen.add(v); out.append(v)
    return out


def main():
    with open(SRC, "rb") as f:
        lines = [ln.rstrip(b"\n").replace(b"\x00", b"") for ln in list(f)[:N_LINES]]
    vocab = corpus_vocab(lines)
    model = os.path.join(os.environ.get("TMPDIR", "/tmp"), "corpus_vocab.bin")
    write_vocab_only_model(model, vocab, n_max_tokens=N_MAX_TOKENS)
    o = orc.Oracle(model, vocab_only=True)
    fd, saved = os.open(os.devnull, os.O_WRONLY), os.dup(2)
    os.dup2(fd, 2)                       # (the reference prints a line per unmatched byte)
    try:
        ids = [o.tokenize(ln) for ln in lines]
        short = [o.tokenize(ln, 12) for ln in lines[:40]]
    finally:
        os.dup2(saved, 2)
    out = {"source": "reference examples/sample_client_texts.txt, lines 1-%d" % N_LINES, "n_max_tokens": N_MAX_TOKENS, "vocab": vocab,
           "texts": [ln.decode("latin-1") for ln in lines], "ids": ids, "ids_n_max_12": short}
    with open(os.path.join(ROOT, "tests", "golden", "sample_texts_golden.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True, separators=(",", ":"))
    n_multi = sum(len(i) for i in ids)
    print(f"{len(lines)} texts, {len(vocab)} vocabulary entries, {n_multi} ids")


if __name__ == "__main__":
    main()
