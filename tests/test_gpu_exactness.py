"""The integer / byte exactness evidence of the path, in the driver-run GPU set.

BASELINE.json's one bit-exact requirement is the token ids (reference examples/test_tokenizer.cpp:70-73); the q4 block
quantizer must reproduce the reference tool's bytes (reference models/quantize.cpp:213-217).  The tests that prove both are
CPU tests (tests/test_host.py, tests/test_tools.py, tests/test_oracle.py) and `pytest -m gpu` deselects them — so the same
test bodies are re-run here under the `gpu` mark, on the GPU box, against the library that box loads.  On top of them: the
ids the GPU context itself produces (bert_tokenize on a context with device weights, bert_encode's tokenization) are the
known answers too."""
import numpy as np
import pytest

import test_host
import test_oracle
import test_tools
from test_host import corpus_golden, fuzz_vocab_model  # noqa: F401  (fixtures)
from test_tools import tools  # noqa: F401  (fixture)

from bert_cpp_amd import pybert

pytestmark = pytest.mark.gpu


def test_tokenizer_reference_known_answers_on_the_gpu_box(sparse_vocab_model, tok_golden):
    test_host.test_tokenizer_reference_known_answers(sparse_vocab_model, tok_golden)
    test_oracle.test_tokenizer_known_answers(sparse_vocab_model, tok_golden)


def test_tokenizer_fuzz_matches_oracle_on_the_gpu_box(fuzz_vocab_model):  # noqa: F811
    test_host.test_tokenizer_fuzz_matches_oracle(fuzz_vocab_model)
    test_host.test_tokenize_batch_on_threads_equals_one_by_one(fuzz_vocab_model)


def test_tokenizer_on_committed_reference_texts_on_the_gpu_box(corpus_golden):  # noqa: F811
    """(the reference's sample texts: tests/test_host.py skips the mounted-tree form of this test on the GPU box)"""
    test_host.test_tokenizer_on_committed_reference_texts(corpus_golden)


@pytest.mark.parametrize("src", ["f32", "f16"])
@pytest.mark.parametrize("qtype", [2, 3])
def test_quantize_tool_bytes_on_the_gpu_box(tools, make_model, model_dir, src, qtype):  # noqa: F811
    test_tools.test_quantize_tool_matches_numpy_quantizers(tools, make_model, model_dir, src, qtype)


def test_f16_conversions_and_block_quantizers_on_the_gpu_box(tools, model_dir):  # noqa: F811
    test_tools.test_f16_conversions_of_the_tool_round_like_numpy(tools, model_dir)
    test_oracle.test_fp16_conversion_matches_ieee()
    test_oracle.test_quantizer_roundtrip_properties()


def test_gpu_context_tokenizes_the_reference_vectors(make_model, tok_golden, tmp_path):
    """A context WITH device weights (bert_load_from_file on the MI355X, not the tokenizer-only loader) gives the reference's
    ids, and bert_encode's embedding is the embedding of exactly those ids."""
    from bert_cpp_amd import ggml_file as gf

    n = tok_golden["n_vocab"]
    vocab = [f"[unused{i}]" for i in range(n)]
    vocab[0], vocab[100], vocab[101], vocab[102], vocab[103] = "[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"
    for i, p in tok_golden["sparse_vocab"].items():
        vocab[int(i)] = p
    path = str(tmp_path / "minilm_real_vocab.bin")
    hp = gf.MODEL_DIMS["minilm-l6"]
    assert hp.n_vocab == n
    gf.write_model(path, hp, gf.synthetic_weights(hp, 0, "sensitive"), gf.FTYPE_BY_NAME["f16"], vocab=[v.encode("utf-8") for v in vocab])
    m = pybert.BertModel(path)
    for t in tok_golden["tests"]:
        ids = m.tokenize(t["text"])
        assert ids == t["ids"], t["text"][:30]
        assert np.array_equal(m.encode(t["text"]), m.eval(ids))
