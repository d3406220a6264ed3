"""The offline accuracy harness: score functions on CPU, and on the GPU the same scores through
libbert.so against scores computed from the CPU oracle's embeddings."""
import numpy as np
import pytest

from bert_cpp_amd import eval_harness as eh


def test_spearman_matches_scipy_with_ties():
    from scipy.stats import spearmanr

    rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.integers(0, 8, size=50).astype(float)        # many ties
        b = a * rng.normal(1, 0.5, size=50) + rng.normal(0, 1, size=50)
        assert abs(eh.spearman(a, b) - spearmanr(a, b).statistic) < 1e-12
    assert eh.spearman([1, 2, 3], [1, 2, 3]) == pytest.approx(1.0)
    assert eh.spearman([1, 2, 3], [3, 2, 1]) == pytest.approx(-1.0)


def test_tasks_with_a_toy_encoder(tmp_path):
    vocab = {"good": 0, "bad": 1, "fine": 2, "awful": 3}
    def encode(texts):
        out = np.zeros((len(texts), 4), dtype=np.float32)
        for i, t in enumerate(texts):
            for w in t.split():
                out[i, vocab[w]] += 1
        return out + 0.01
    pairs = [(5.0, "good fine", "good fine"), (0.0, "good", "awful"), (3.0, "good fine", "good bad"), (1.0, "fine", "bad awful")]
    r = eh.sts_task(encode, pairs)
    assert r["cos_sim"]["spearman"] > 0.9 and r["n_pairs"] == 4
    train = [("pos", "good fine"), ("pos", "good"), ("pos", "fine"), ("neg", "bad"), ("neg", "awful"), ("neg", "bad awful")] * 3
    test = [("pos", "fine good"), ("neg", "awful bad")]
    assert eh.classification_task(encode, train, test)["accuracy"] == 1.0
    p = tmp_path / "p.tsv"
    p.write_text("4.5\ta b\tc d\nbroken line\n1.0\tx\ty\n", encoding="utf-8")
    assert eh.read_tsv(str(p), 3) == [("4.5", "a b", "c d"), ("1.0", "x", "y")]


@pytest.mark.gpu
def test_sts_and_classification_scores_match_the_oracle(tmp_path):
    from bert_cpp_amd import ggml_file as gf, pybert
    from oracle import oracle as orc

    hp = gf.MODEL_DIMS["tiny-h128"]
    words = ["[PAD]", "[UNK]"] + [f"w{i}" for i in range(2, 101)] + ["[CLS]", "[SEP]"] + \
            ["happy", "sad", "angry", "calm", "##ly", "##ness", "very", "not", "so", "is", "the", "cat", "dog", "day", "."]
    vocab = [w.encode() for w in words] + [f"[unused{i}]".encode() for i in range(len(words), hp.n_vocab)]
    path = str(tmp_path / "m.bin")
    gf.write_model(path, hp, gf.synthetic_weights(hp, 21), gf.FTYPE_F16, vocab=vocab)
    rng = np.random.default_rng(1)
    lex = ["happy", "sad", "angry", "calm", "very", "not", "so", "is", "the", "cat", "dog", "day", "happily", "sadness", "."]
    sent = lambda: " ".join(rng.choice(lex, size=int(rng.integers(3, 12))))
    m, o = pybert.BertModel(path), orc.Oracle(path)
    gpu_encode = lambda texts: m.encode_batch(texts, batch_size=32)
    cpu_encode = lambda texts: np.stack([o.eval(o.tokenize(t)) for t in texts])
    pairs_txt = [(sent(), sent()) for _ in range(120)]
    gold = eh.cosine_rows(cpu_encode([a for a, _ in pairs_txt]), cpu_encode([b for _, b in pairs_txt]))
    r = eh.sts_task(gpu_encode, [(float(g), a, b) for g, (a, b) in zip(gold, pairs_txt)])
    assert r["cos_sim"]["spearman"] > 0.999                  # the GPU path ranks the pairs like the oracle
    texts = [sent() for _ in range(160)]
    labels = ["a" if v > 0 else "b" for v in cpu_encode(texts)[:, 0]]       # a label the oracle's embedding determines
    data = list(zip(labels, texts))
    acc_gpu = eh.classification_task(gpu_encode, data[:120], data[120:])["accuracy"]
    acc_cpu = eh.classification_task(cpu_encode, data[:120], data[120:])["accuracy"]
    assert abs(acc_gpu - acc_cpu) <= 0.05 and acc_gpu > 0.6
