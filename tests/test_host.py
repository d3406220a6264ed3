"""CPU-side tests of the product library: the C-ABI loads and exports every declared symbol, the
host tokenizer is bit-exact with the reference's known answers and with the oracle on fuzzed
text, model-file validation, and the loud failure when no GPU is present."""
import contextlib
import os
import random
import re
import struct

import numpy as np
import pytest

from bert_cpp_amd import ggml_file as gf
from bert_cpp_amd import pybert as libbert
from oracle import oracle as orc

from conftest import ROOT, write_vocab_only_model


def has_gpu():
    return os.path.exists("/dev/kfd")


@contextlib.contextmanager
def silenced_stderr():
    """The reference prints one stderr line per unmatched byte; keep the test log clean."""
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)
    try:
        yield
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)


def test_library_exports_every_declared_symbol():
    L = libbert.lib()
    for sym in libbert.BERT_H_SYMBOLS + libbert.BERT_HIP_H_SYMBOLS:
        assert hasattr(L, sym), sym
    # and the headers declare exactly these
    T = libbert.test_lib()
    for sym in libbert.BERT_HIP_TEST_H_SYMBOLS:
        assert hasattr(T, sym), sym
        assert not hasattr(L, sym), f"{sym} must not ship in the product library"
    for hdr, syms in (("bert.h", libbert.BERT_H_SYMBOLS), ("bert_hip.h", libbert.BERT_HIP_H_SYMBOLS),
                      ("bert_hip_test.h", libbert.BERT_HIP_TEST_H_SYMBOLS)):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        declared = set(re.findall(r"BERT_API[^;(]*?\b(bert_\w+)\s*\(", text))
        assert declared == set(syms), (hdr, declared ^ set(syms))
    assert b"gfx950" in L.bert_hip_version()
    # nm -D of the product library shows no test hook
    import subprocess
    names = subprocess.run(["nm", "-D", "--defined-only", libbert.LIB_PATH], capture_output=True, text=True).stdout
    assert "_test_" not in names and "bench_ffn" not in names


def test_tokenizer_reference_known_answers(sparse_vocab_model, tok_golden):
    """reference examples/test_tokenizer.cpp:70-73 through the product's bert_tokenize."""
    m = libbert.BertModel(sparse_vocab_model, tokenizer_only=True)
    assert m.n_max_tokens == 512
    for t in tok_golden["tests"]:
        assert m.tokenize(t["text"]) == t["ids"], t["text"][:30]
    assert m.id_to_token(1037) == b"a"
    assert m.id_to_token(5358) == b"##om"
    assert m.id_to_token(-1) == b"[UNK TOKEN from bert_vocab]"
    assert m.id_to_token(10 ** 6) == b"[UNK TOKEN from bert_vocab]"


WORDS = ("the of and to in a is that for it as was with be by on not he i this are or his from at which but have an "
         "had they you were their one all we can her has there been if more when will would who so no embedding "
         "tokenizer quantization sentence transformer playing unbelievable internationalization xylophone").split()
PIECES = ["##s", "##ing", "##ed", "##ly", "##er", "##ation", "##ize", "##able", "##un", "##a", "##b", "##c", "##e", "##i",
          "##n", "##o", "##t", "##x", "##y", "##z", "##1", "##2", "##00"]


@pytest.fixture(scope="module")
def fuzz_vocab_model(tmp_path_factory):
    rnd = random.Random(0)
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    tail = list("abcdefghijklmnopqrstuvwxyz0123456789") + list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")
    tail += WORDS + PIECES + ["10", "100", "2023", "##", "#", "play", "un", "believ", "inter", "national"]
    tail += ["the", "##s"]                      # duplicates: first wins for words, last wins for subwords
    tail += ["Ab", "é", "日本"]                   # never matchable / non-ASCII entries
    rnd.shuffle(tail)
    path = str(tmp_path_factory.mktemp("fuzz") / "fuzz_vocab.bin")
    write_vocab_only_model(path, vocab + tail, n_max_tokens=64)
    return path


def random_text(rnd: random.Random) -> bytes:
    parts = []
    for _ in range(rnd.randint(0, 40)):
        k = rnd.random()
        if k < 0.45:
            w = rnd.choice(WORDS)
            if rnd.random() < 0.3:
                w = w.capitalize() if rnd.random() < 0.5 else w.upper()
            if rnd.random() < 0.3:
                w += rnd.choice(["s", "ing", "ed", "ly", "ization", "x1", "123"])
            parts.append(w.encode())
        elif k < 0.55:
            parts.append(str(rnd.randint(0, 10 ** rnd.randint(1, 6))).encode())
        elif k < 0.7:
            parts.append(rnd.choice("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~").encode() * rnd.randint(1, 3))
        elif k < 0.8:
            parts.append(rnd.choice(["Québec", "syömme", "täällä", "Ñandú", "Ça", "Ýmir", "ÀÉÎÕÜ", "ß", "æø", "日本語",
                                     "😀", "ÿ"]).encode())
        elif k < 0.9:
            # raw / malformed bytes: stray continuation bytes, truncated leads, 0xC3 before ASCII upper-case
            parts.append(bytes(rnd.choice([[0xC3], [0xC3, 0x41], [0x80, 0x42], [0xE2, 0x82], [0xF0, 0x9F], [0xFF, 0x5A],
                                           [0xC3, 0x80, 0xC3], [0xE0, 0x41, 0x42, 0x43]])))
        else:
            parts.append(rnd.choice([b"a1b2", b"x_y", b"p.m.", b"don't", b"C++", b"##ing", b"#", b"e-mail", b"U.S.A."]))
        parts.append(rnd.choice([b" ", b" ", b"  ", b"\t", b"\n", b""]))
    return b"".join(parts).replace(b"\x00", b"")


def test_tokenizer_fuzz_matches_oracle(fuzz_vocab_model):
    """Differential test against the line-for-line restatement of reference bert.cpp:199-325."""
    m = libbert.BertModel(fuzz_vocab_model, tokenizer_only=True)
    o = orc.Oracle(fuzz_vocab_model, vocab_only=True)
    rnd = random.Random(1234)
    n_nontrivial = 0
    with silenced_stderr():
        for _ in range(3000):
            text = random_text(rnd)
            for n_max in (64, rnd.randint(2, 20)):
                a = m.tokenize(text, n_max)
                b = o.tokenize(text, n_max)
                assert a == b, (text, n_max, a, b)
                n_nontrivial += len(a) > 2
    assert n_nontrivial > 4000
    for i in range(0, 400):
        assert m.id_to_token(i) == o.id_to_token(i)


def test_tokenize_batch_on_threads_equals_one_by_one(fuzz_vocab_model):
    """bert_hip_tokenize_batch (the first stage of bert_encode_batch) on 1, 4 and 9 threads gives, text for text, the
    ids of bert_tokenize."""
    m = libbert.BertModel(fuzz_vocab_model, tokenizer_only=True)
    rnd = random.Random(77)
    texts = [random_text(rnd) for _ in range(1000)]
    with silenced_stderr():
        want = [m.tokenize(t) for t in texts]
        for n_threads in (1, 4, 9):
            assert m.tokenize_batch(texts, n_threads) == want
        assert m.tokenize_batch([], 4) == []
        assert m.tokenize_batch(texts[:3], 64) == want[:3]


def test_tokenizer_on_reference_corpus_if_present(sparse_vocab_model):
    """Real text (reference examples/sample_client_texts.txt) when the reference tree is mounted."""
    path = "/root/reference/examples/sample_client_texts.txt"
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    m = libbert.BertModel(sparse_vocab_model, tokenizer_only=True)
    o = orc.Oracle(sparse_vocab_model, vocab_only=True)
    with silenced_stderr():
        with open(path, "rb") as f:
            for line in list(f)[:600]:
                line = line.rstrip(b"\n").replace(b"\x00", b"")
                assert m.tokenize(line) == o.tokenize(line)


@pytest.fixture(scope="module")
def corpus_golden(tmp_path_factory):
    import json
    with open(os.path.join(ROOT, "tests", "golden", "sample_texts_golden.json")) as f:
        g = json.load(f)
    path = str(tmp_path_factory.mktemp("corpus") / "corpus_vocab.bin")
    write_vocab_only_model(path, g["vocab"], n_max_tokens=g["n_max_tokens"])
    return g, path


def test_tokenizer_on_committed_reference_texts(corpus_golden):
    """The first 200 of the reference's own sample texts (tests/golden/sample_texts_golden.json, written by make_corpus_golden.py where the
    reference tree is mounted): the product's bert_tokenize gives the ids the oracle's restatement of bert.cpp:199-325 gave — whole words,
    multi-piece splits, punctuation, unmatched bytes, truncation at n_max_tokens — and the oracle still gives them too.  Runs on the GPU
    box as well (tests/test_gpu_exactness.py), where /root/reference does not exist."""
    g, path = corpus_golden
    m = libbert.BertModel(path, tokenizer_only=True)
    o = orc.Oracle(path, vocab_only=True)
    texts = [t.encode("latin-1") for t in g["texts"]]
    with silenced_stderr():
        for t, want in zip(texts, g["ids"]):
            assert m.tokenize(t) == want, t[:40]
            assert o.tokenize(t) == want, t[:40]
        for t, want in zip(texts, g["ids_n_max_12"]):
            assert m.tokenize(t, 12) == want, t[:40]
        assert m.tokenize_batch(texts, 4) == g["ids"]
    assert sum(len(i) > 12 for i in g["ids"]) > 100 and max(len(i) for i in g["ids"]) == g["n_max_tokens"]


def test_model_file_validation(tmp_path, capfd):
    hp = gf.MODEL_DIMS["tiny"]
    good = str(tmp_path / "good.bin")
    gf.write_model(good, hp, gf.synthetic_weights(hp, 0), gf.FTYPE_F16)
    data = open(good, "rb").read()
    L = libbert.lib()

    def load(b):
        p = str(tmp_path / "case.bin")
        open(p, "wb").write(b)
        return L.bert_load_from_file(p.encode())

    assert not L.bert_load_from_file(b"/nonexistent/model.bin")
    assert "failed to open" in capfd.readouterr().err
    assert not load(b"\x00" * 100)
    assert "bad magic" in capfd.readouterr().err
    bad_f16 = bytearray(data); bad_f16[28:32] = struct.pack("<i", 7)
    assert not load(bytes(bad_f16))
    assert "bad f16 value" in capfd.readouterr().err
    assert not load(data[:-100])                                  # truncated last tensor
    assert "truncated" in capfd.readouterr().err
    renamed = data.replace(b"encoder.layer.0.output.dense.bias", b"encoder.layer.0.output.dense.bogu")
    assert not load(renamed)
    assert "unknown tensor" in capfd.readouterr().err


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_load_fails_loudly_without_gpu(tmp_path, capfd):
    """No CPU fallback: without a HIP device bert_load_from_file returns NULL with a clear message."""
    hp = gf.MODEL_DIMS["tiny"]
    p = str(tmp_path / "m.bin")
    gf.write_model(p, hp, gf.synthetic_weights(hp, 0), gf.FTYPE_F16)
    with pytest.raises(RuntimeError):
        libbert.BertModel(p)
    assert "no HIP device" in capfd.readouterr().err
    # a tokenizer-only context refuses to evaluate
    m = libbert.BertModel(p, tokenizer_only=True)
    out = m.eval([101, 5, 102])
    assert np.isnan(out).all()
    assert "no device weights" in capfd.readouterr().err


@pytest.mark.parametrize("ftype", ["q4_0", "q4_1"])
def test_legacy_q4_block_layout_is_detected_and_converted(tmp_path, ftype):
    """Model files quantised by early-2023 ggml use 20-byte q4_0 / 24-byte q4_1 blocks (f32 scales, nibbles 2j | 2j+1): the
    sizes the reference's README prints (README.md:105-107).  The records carry no byte counts; the loader tells the layouts
    apart by the length of the tensor section and re-blocks legacy tensors, after which they are byte-identical to the same
    weights written in the current layout."""
    from bert_cpp_amd import ggml_file as gf

    hp = gf.MODEL_DIMS["tiny-h128"]
    w = gf.synthetic_weights(hp, 3)
    cur, leg = str(tmp_path / "cur.bin"), str(tmp_path / "leg.bin")
    gf.write_model(cur, hp, w, gf.FTYPE_BY_NAME[ftype])
    gf.write_model(leg, hp, w, gf.FTYPE_BY_NAME[ftype], legacy_q4=True)
    assert os.path.getsize(leg) > os.path.getsize(cur)
    n0, l0, d0 = libbert.model_digest(cur)
    n1, l1, d1 = libbert.model_digest(leg)
    assert (n0, l0) == (5 + 16 * hp.n_layer, False) and (n1, l1) == (n0, True)
    assert d0 == d1
    # a legacy file cut short is still rejected
    data = open(leg, "rb").read()
    bad = str(tmp_path / "bad.bin")
    open(bad, "wb").write(data[:-100])
    with silenced_stderr():
        with pytest.raises(RuntimeError):
            libbert.model_digest(bad)


def test_bench_line_stays_parseable_for_the_driver():
    """Round 4's line was 26 kB and the driver could not parse it.  The compact line built from that
    very round's full result (profiles/r4_bench_line.json: twelve entries under `also`) must stay under 6 kB, keep the
    contract keys, `roofline` and `cpu_baseline`, and round-trip through json."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "r4_bench_line.json")) as f:
        full = json.load(f)
    contract_keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                     "dtype", "data", "config")
    contract = {k: full[k] for k in contract_keys}
    e = dict(full)
    e["eval_batch_api"] = {"value": 300000.123456, "ms_per_call": 0.85333333, "vs_eval_packed": 0.99123456, "rows_equal_eval_packed": True}
    extras = dict(full["also"])
    extras["broken_entry"] = {"workload": "x", "error": "RuntimeError: " + "y" * 1000}
    line = bench.compact_line(contract, e, extras, "/some/where/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 6144, len(text)
    back = json.loads(text)
    for k in contract_keys + ("roofline", "cpu_baseline", "host_to_host", "eval_batch_api", "also", "mean_cosine_vs_cpu"):
        assert k in back, k
    assert back["roofline"]["bound"] == "mfma" and 0 < back["roofline"]["frac"] < 1 and back["roofline"]["traffic"] > 0
    assert {"value", "unit", "cores", "kind", "sample"} <= set(back["cpu_baseline"])
    assert set(extras) == set(back["also"])
    for k, v in back["also"].items():
        assert len(v) <= 12, (k, v)
    # every value of the headline survives to 6 digits
    assert abs(back["value"] - full["value"]) < 1e-6 * full["value"]
    # a pathological input (every entry an error, long strings) still gives a parseable, short line
    worst = bench.compact_line(contract, e, {f"entry{i}": {"error": "z" * 5000} for i in range(40)}, None)
    assert len(json.dumps(worst)) < bench.LINE_LIMIT


def test_library_exports_nothing_but_the_c_abi():
    """A drop-in for the reference's libbert.so (reference bert.h:8-16: BERT_API on its 11 functions) exports those names and the
    bert_hip_* extensions — no kernel stubs, no __hip_cuid_*, no weak libstdc++ instantiations (bert.cpp_amd/libbert.map)."""
    import subprocess

    def exported(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return sorted(l.split()[-1] for l in out.splitlines() if l.strip())

    assert exported(libbert.LIB_PATH) == sorted(libbert.BERT_H_SYMBOLS + libbert.BERT_HIP_H_SYMBOLS)
    assert exported(libbert.TEST_LIB_PATH) == sorted(libbert.BERT_H_SYMBOLS + libbert.BERT_HIP_H_SYMBOLS + libbert.BERT_HIP_TEST_H_SYMBOLS)
