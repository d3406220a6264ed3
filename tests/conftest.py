import json
import os
import struct
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("BERT_HIP_QUIET", "1")      # no load-progress text on stdout during tests
# The parity tests are sized for the CPU oracle: batches of a few hundred tokens, meant for the BATCH route's kernels (windows,
# layer tail, all layers in one launch).  Since round 5 the engine sends calls of up to 768 tokens down the latency route (same
# bits, faster for small calls): here the cap stays at one window, and the tests of the route itself
# (test_latency_route_*) set the shipped default explicitly.
os.environ.setdefault("BERT_HIP_LATENCY", "128")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("models"))


@pytest.fixture(scope="session")
def tok_golden():
    with open(os.path.join(GOLDEN, "tokenizer_golden.json"), encoding="utf-8") as f:
        return json.load(f)


def write_vocab_only_model(path, vocab, n_max_tokens=512):
    """Header + vocab of the bert.cpp file format and no tensors (enough for the tokenizer)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 0x67676D6C))
        f.write(struct.pack("<7i", len(vocab), n_max_tokens, 384, 1536, 12, 6, 1))
        for tok in vocab:
            b = tok if isinstance(tok, bytes) else tok.encode("utf-8")
            f.write(struct.pack("<I", len(b)))
            f.write(b)


@pytest.fixture(scope="session")
def sparse_vocab_model(tok_golden, model_dir):
    n = tok_golden["n_vocab"]
    vocab = [f"[unused{i}]" for i in range(n)]
    vocab[0], vocab[100], vocab[101], vocab[102], vocab[103] = "[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"
    for i, p in tok_golden["sparse_vocab"].items():
        vocab[int(i)] = p
    path = os.path.join(model_dir, "sparse_vocab.bin")
    write_vocab_only_model(path, vocab)
    return path


def cosine(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


_MODEL_CACHE = {}


@pytest.fixture(scope="session")
def make_model(model_dir):
    """make_model(dims, ftype, seed) -> (path, hparams); synthetic seeded weights, cached per session."""
    from bert_cpp_amd import ggml_file as gf

    def _make(dims, ftype, seed=0):
        key = (dims, ftype, seed)
        if key not in _MODEL_CACHE:
            path = os.path.join(model_dir, f"{dims}_{ftype}_s{seed}.bin")
            hp = gf.make_synthetic_model(path, dims, ftype, seed=seed)
            _MODEL_CACHE[key] = (path, hp)
        return _MODEL_CACHE[key]

    return _make
