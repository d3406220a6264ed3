"""CPU tests of the host-side tools around the hot path: the native quantizer must write the
same bytes as the NumPy restatement of the block quantizers that manufactures every q4 test model, and the example
programs must fail loudly (no CPU fallback) when there is no HIP device."""
import os
import subprocess

import numpy as np
import pytest

from bert_cpp_amd import ggml_file as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bert.cpp_amd", "bin")


@pytest.fixture(scope="module")
def tools():
    subprocess.run(["make", "-C", os.path.join(ROOT, "bert.cpp_amd"), "tools", "examples"], check=True,
                   stdout=subprocess.DEVNULL)
    return BIN


@pytest.mark.parametrize("src", ["f32", "f16"])
@pytest.mark.parametrize("qtype", [2, 3])
def test_quantize_tool_matches_numpy_quantizers(tools, make_model, model_dir, src, qtype):
    # same seed -> same f32 weights; the python writer quantizes from f32, the tool from the file's tensors
    path_src, hp = make_model("tiny-d64", src, 3)
    out = os.path.join(model_dir, f"tool_{src}_{qtype}.bin")
    r = subprocess.run([os.path.join(tools, "bert-quantize"), path_src, out, str(qtype)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "quantize time" in r.stdout
    got = gf.read_model(out)
    assert got.ftype == qtype
    srcm = gf.read_model(path_src)
    assert got.vocab == srcm.vocab
    quant = gf.quantize_q4_0 if qtype == 2 else gf.quantize_q4_1
    n_quantized = 0
    for name, (ttype, shape, raw) in srcm.raw.items():
        gt, gshape, graw = got.raw[name]
        assert gshape == shape
        if len(shape) == 2 and name.endswith("weight"):
            w = np.frombuffer(raw, dtype=np.float32 if ttype == 0 else np.float16).astype(np.float32).reshape(shape)
            assert gt == qtype
            assert bytes(graw) == quant(w).tobytes(), name
            n_quantized += 1
        else:
            assert gt == ttype and bytes(graw) == bytes(raw), name
    assert n_quantized == 3 + 6 * hp.n_layer     # 3 embedding tables + q,k,v,o,ff_i,ff_o per layer

    if src == "f16":
        # the reference pipeline (HF -> f16 file -> quantize tool): the whole file equals what the python writer,
        # which rounds its f32 weights through f16 first, emits for this ftype from the same seed
        path_py, _ = make_model("tiny-d64", "q4_0" if qtype == 2 else "q4_1", 3)
        assert open(out, "rb").read() == open(path_py, "rb").read()


def test_quantize_tool_rejects_bad_input(tools, make_model, model_dir, tmp_path):
    exe = os.path.join(tools, "bert-quantize")
    path_q, _ = make_model("tiny-d64", "q4_0", 3)
    out = str(tmp_path / "o.bin")
    r = subprocess.run([exe, path_q, out, "2"], capture_output=True, text=True)       # already quantized
    assert r.returncode == 1 and "unsupported source type" in r.stderr
    path_f, _ = make_model("tiny-d64", "f32", 3)
    assert subprocess.run([exe, path_f, out, "7"], capture_output=True).returncode == 1    # unknown type
    assert subprocess.run([exe, path_f, out], capture_output=True).returncode == 1         # usage
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"not a model")
    r = subprocess.run([exe, str(junk), out, "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "bad magic" in r.stderr
    trunc = tmp_path / "trunc.bin"
    trunc.write_bytes(open(path_f, "rb").read()[:-100])
    r = subprocess.run([exe, str(trunc), out, "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "truncated" in r.stderr


def test_f16_conversions_of_the_tool_round_like_numpy(tools, model_dir):
    """Scales are stored as f16: feed blocks whose scale lands on subnormal / tie / overflow-adjacent values."""
    hp = gf.BertHParams(n_vocab=8, n_max_tokens=8, n_embd=64, n_intermediate=64, n_head=1, n_layer=1)
    w = gf.synthetic_weights(hp, seed=1)
    rng = np.random.default_rng(0)
    tbl = w["embeddings.word_embeddings.weight"]
    mags = np.array([1e-7, 3.1e-6, 6.1e-5, 6.2e-5, 1.0 + 2.0**-11, 8 * (1.0 + 3 * 2.0**-11), 300.0, 5e4], dtype=np.float32)
    tbl[:] = (rng.standard_normal(tbl.shape).astype(np.float32) * mags[: tbl.shape[0], None]).astype(np.float32)
    src = os.path.join(model_dir, "edge_f32.bin")
    gf.write_model(src, hp, w, 0)
    for qtype, quant in ((2, gf.quantize_q4_0), (3, gf.quantize_q4_1)):
        out = os.path.join(model_dir, f"edge_q{qtype}.bin")
        subprocess.run([os.path.join(tools, "bert-quantize"), src, out, str(qtype)], check=True, stdout=subprocess.DEVNULL)
        got = gf.read_model(out).raw["embeddings.word_embeddings.weight"][2]
        assert bytes(got) == quant(tbl).tobytes()


@pytest.mark.parametrize("exe", ["bert-main", "bert-server"])
def test_examples_fail_loudly_without_a_device_or_model(tools, make_model, exe):
    import torch
    path, _ = make_model("tiny", "f16", 0)
    r = subprocess.run([os.path.join(tools, exe), "-m", "/nonexistent/model.bin", "--port", "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "failed to load model" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([os.path.join(tools, exe), "-m", path, "--port", "0"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "failed to load model" in r.stderr
