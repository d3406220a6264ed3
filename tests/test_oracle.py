"""Pins the CPU oracle: reference known-answer token ids, an independent float implementation
(HuggingFace BertModel fixtures), fp16 conversion, and the q4 block formats."""
import json
import os

import numpy as np
import pytest

from bert_cpp_amd import ggml_file as gf
from oracle import oracle as orc

from conftest import GOLDEN


def cos(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def test_tokenizer_known_answers(sparse_vocab_model, tok_golden):
    """reference examples/test_tokenizer.cpp:70-73 — bit-exact ids."""
    o = orc.Oracle(sparse_vocab_model, vocab_only=True)
    for t in tok_golden["tests"]:
        assert o.tokenize(t["text"]) == t["ids"], t["text"][:30]


def test_tokenizer_quirks(sparse_vocab_model):
    o = orc.Oracle(sparse_vocab_model, vocab_only=True)
    # non-ASCII bytes are dropped and act as separators; unknown pieces emit nothing (no [UNK])
    assert o.tokenize("你好") == [101, 102]
    assert o.tokenize("") == [101, 102]
    # truncation: at most n_max_tokens ids, last is [SEP]   (bert.cpp:300-301,323)
    ids = o.tokenize("a " * 600, n_max_tokens=16)
    assert len(ids) == 16 and ids[0] == 101 and ids[-1] == 102 and all(i == 1037 for i in ids[1:-1])
    assert o.id_to_token(1037) == b"a" and o.id_to_token(99999) == b"[UNK TOKEN from bert_vocab]"
    # subword ids map back to their full '##' spelling (bert.cpp:121-134)
    assert o.id_to_token(5358) == b"##om"


def test_fp16_conversion_matches_ieee():
    L = orc.lib()
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.normal(0, 1, 2000), rng.normal(0, 1e-6, 500), rng.normal(0, 3e4, 500),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001,
                  6.1e-5, 6.0e-5, 5.96e-8, 1.0009765625, 1.00048828125, 1.000488281250001])]).astype(np.float32)
    for v in vals:
        want = int(np.float32(v).astype(np.float16).view(np.uint16))
        assert L.oracle_f2h_soft(float(v)) == want, v
        assert L.oracle_f2h(float(v)) == want, v
    for h in list(range(0, 0x7c00, 37)) + [0x8001, 0x83ff, 0xfbff, 0x0001, 0x03ff, 0x0400]:
        assert L.oracle_h2f_soft(h) == float(np.uint16(h).view(np.float16))


def test_quantizer_roundtrip_properties():
    rng = np.random.default_rng(1)
    w = rng.normal(0, 1, (16, 128)).astype(np.float32)
    q0 = gf.quantize_q4_0(w); q1 = gf.quantize_q4_1(w)
    assert q0.shape == (16, 4, 18) and q1.shape == (16, 4, 20)
    d0 = gf.dequantize_q4_0(q0); d1 = gf.dequantize_q4_1(q1)
    # error bounded by one quantisation step per block
    step0 = np.abs(w.reshape(-1, 32)).max(axis=1) / 8
    assert (np.abs(d0 - w).reshape(-1, 32) <= step0[:, None] * 1.01 + 1e-3).all()
    rngb = w.reshape(-1, 32).max(axis=1) - w.reshape(-1, 32).min(axis=1)
    assert (np.abs(d1 - w).reshape(-1, 32) <= rngb[:, None] / 15 * 0.51 + 2e-3).all()
    # idempotence on already-representable data
    assert np.array_equal(gf.quantize_q4_0(d0), q0)
    # all-zero block
    z = gf.quantize_q4_0(np.zeros((1, 32), np.float32))
    assert np.array_equal(gf.dequantize_q4_0(z), np.zeros((1, 32), np.float32))


@pytest.fixture(scope="module")
def hf_golden():
    with open(os.path.join(GOLDEN, "hf_tiny_golden.json")) as f:
        return json.load(f)


def test_oracle_matches_huggingface(hf_golden):
    """Independent implementation check: HF BertModel (gelu_new, eps 1e-5) + mean-pool + L2."""
    o = orc.Oracle(os.path.join(GOLDEN, hf_golden["model"]))
    for ids, want, h8 in zip(hf_golden["sentences"], hf_golden["embeddings"], hf_golden["layer1_tok0_first8"]):
        got, hid = o.eval(ids, orc.MODE_PLAIN, want_hidden=True)
        assert np.allclose(hid[1][0, :8], h8, atol=2e-4), (len(ids), hid[1][0, :8], h8)
        assert np.abs(got - np.asarray(want)).max() < 2e-5, len(ids)
        assert cos(got, want) > 1 - 1e-9
        # f32 file: ggml mode only changes exp/GELU lookups -> fp16-level differences
        got_g = o.eval(ids, orc.MODE_GGML)
        assert cos(got_g, want) > 1 - 1e-5


@pytest.mark.parametrize("ftype,min_cos", [("f32", 1 - 1e-5), ("f16", 1 - 1e-5), ("q4_0", 0.995), ("q4_1", 0.995)])
def test_oracle_modes_agree(model_dir, ftype, min_cos):
    """ggml-faithful vs plain numerics on the same file: the gap is the reference's own noise."""
    path = os.path.join(model_dir, f"tiny_{ftype}.bin")
    hp = gf.make_synthetic_model(path, "tiny", ftype, seed=3)
    o = orc.Oracle(path)
    ids = gf.synthetic_token_ids(4, 24, hp.n_vocab, seed=5)
    for s in ids:
        a = o.eval(s, orc.MODE_GGML); b = o.eval(s, orc.MODE_PLAIN)
        assert abs(np.linalg.norm(a) - 1) < 1e-5
        assert cos(a, b) > min_cos, (ftype, cos(a, b))


def test_oracle_quantized_close_to_f32_source(model_dir):
    """q4 files are noisy versions of the f32 model (what the 0.99 contract is about)."""
    hp = gf.MODEL_DIMS["tiny"]
    w = gf.synthetic_weights(hp, seed=11)
    paths = {}
    for ft in ("f32", "q4_0", "q4_1"):
        paths[ft] = os.path.join(model_dir, f"tinyq_{ft}.bin")
        gf.write_model(paths[ft], hp, w, gf.FTYPE_BY_NAME[ft])
    ids = gf.synthetic_token_ids(3, 32, hp.n_vocab, seed=2)
    ref = orc.Oracle(paths["f32"])
    for ft in ("q4_0", "q4_1"):
        o = orc.Oracle(paths[ft])
        for s in ids:
            assert cos(o.eval(s), ref.eval(s)) > 0.9, ft


def test_oracle_error_paths(model_dir, tmp_path):
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\x00" * 64)
    with pytest.raises(RuntimeError):
        orc.Oracle(str(bad))
    path = os.path.join(model_dir, "tiny_err.bin")
    hp = gf.make_synthetic_model(path, "tiny", "f32", seed=3)
    o = orc.Oracle(path)
    with pytest.raises(RuntimeError):
        o.eval(list(range(hp.n_max_tokens + 1)))          # too many tokens (bert.cpp:765-769)
