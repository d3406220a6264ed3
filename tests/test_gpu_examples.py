"""GPU tests of the example programs: bert-main's stdout and bert-server's wire protocol
(reference examples/main.cpp:8-77, examples/server.cpp:36-124, examples/sample_client.py:9-22), checked against
the library they wrap and the CPU oracle."""
import os
import re
import socket
import struct
import subprocess
import threading

import numpy as np
import pytest

from bert_cpp_amd import ggml_file as gf
from bert_cpp_amd import pybert
from oracle import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bert.cpp_amd", "bin")

WORDS = ["[PAD]", "[UNK]"] + [f"w{i}" for i in range(2, 101)] + ["[CLS]", "[SEP]"] + list("abcdefghij") + \
        ["hello", "world", "##ing", "##s", "test", ",", ".", "!", "embed", "##ding", "server", "client"]


@pytest.fixture(scope="module")
def text_model(tmp_path_factory):
    subprocess.run(["make", "-C", os.path.join(ROOT, "bert.cpp_amd"), "examples"], check=True, stdout=subprocess.DEVNULL)
    hp = gf.MODEL_DIMS["tiny-h128"]
    vocab = [w.encode() for w in WORDS] + [f"[unused{i}]".encode() for i in range(len(WORDS), hp.n_vocab)]
    path = str(tmp_path_factory.mktemp("ex") / "text_model_q4_0.bin")
    gf.write_model(path, hp, gf.synthetic_weights(hp, 11), gf.FTYPE_Q4_0, vocab=vocab)
    return path, hp


def test_main_prints_ids_pieces_embedding_and_timings(text_model):
    path, hp = text_model
    prompt = "Hello world, testing embeddings!"
    r = subprocess.run([os.path.join(BIN, "bert-main"), "-m", path, "-p", prompt, "-t", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    ref = oracle.Oracle(path)
    want_ids = ref.tokenize(prompt)
    i_n = next(i for i, l in enumerate(lines) if l.startswith("main: number of tokens in prompt = "))
    assert int(lines[i_n].rsplit("=", 1)[1]) == len(want_ids)
    arrays = [l for l in lines if l.startswith("[") and l.endswith("]")]
    ids = [int(x) for x in arrays[0].strip("[]").split(",") if x.strip()]
    assert ids == list(want_ids)
    pieces = [l for l in lines if re.match(r"^\d+ -> ", l)]
    assert [int(p.split(" -> ")[0]) for p in pieces] == ids
    assert pieces[0].endswith("[CLS]") and pieces[-1].endswith("[SEP]") and "101 -> [CLS]" == pieces[0]
    emb = np.array([float(x) for x in arrays[1].strip("[]").split(",") if x.strip()], dtype=np.float32)
    assert emb.shape == (hp.n_embd,)
    want = ref.eval(want_ids)
    assert np.abs(emb - want).max() < 5e-3                 # printed with 4 decimals; q4_0 tolerance of the parity tests
    assert float(emb @ want / np.linalg.norm(emb) / np.linalg.norm(want)) > 0.99
    for tag in ("main:     load time =", "main:  eval time =", "main:    total time ="):
        assert any(l.startswith(tag) for l in lines), tag
    assert " ms per token" in r.stdout


def test_main_usage_and_unknown_argument(text_model):
    path, _ = text_model
    r = subprocess.run([os.path.join(BIN, "bert-main"), "--bogus"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "usage:" in r.stderr          # reference bert.cpp:181-185: usage, exit(0)
    r = subprocess.run([os.path.join(BIN, "bert-main"), "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "--model" in r.stderr


def _recv_exact(s, n):
    buf = b""
    while len(buf) < n:
        c = s.recv(n - len(buf))
        assert c, "server closed the connection"
        buf += c
    return buf


def test_server_protocol_and_concurrent_clients(text_model):
    path, hp = text_model
    srv = subprocess.Popen([os.path.join(BIN, "bert-server"), "-m", path, "--port", "0"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
    try:
        line = srv.stdout.readline()
        m = re.match(r"Server running on port (\d+) with \d+ threads", line)
        assert m, line
        port = int(m.group(1))
        lib = pybert.BertModel(path)
        texts = [["hello world", "testing server!", "a b c d e f g h i j " * 20],
                 ["client test.", "HELLO, embedding", "hello world"],
                 ["world hello", "x", "tests, tests, tests."]]
        got = [[None] * 3 for _ in texts]
        errors = []

        def client(ci):
            try:
                with socket.create_connection(("127.0.0.1", port), timeout=60) as s:
                    n_embd = struct.unpack("<i", _recv_exact(s, 4))[0]
                    assert n_embd == hp.n_embd
                    for ti, t in enumerate(texts[ci]):
                        s.sendall(t.encode())
                        got[ci][ti] = np.frombuffer(_recv_exact(s, 4 * n_embd), dtype="<f4").copy()
            except Exception as e:       # surfaced in the main thread
                errors.append(e)

        threads = [threading.Thread(target=client, args=(ci,)) for ci in range(len(texts))]
        for t in threads: t.start()
        for t in threads: t.join(120)
        assert not errors, errors
        for ci, ts in enumerate(texts):
            for ti, t in enumerate(ts):
                want = lib.encode(t)
                assert np.array_equal(got[ci][ti], want), (ci, ti, float(np.abs(got[ci][ti] - want).max()))   # same library, same kernels: bit-equal
        assert np.array_equal(got[0][0], got[1][2])                            # batched with other requests or not

        # the bundled python client speaks the same protocol
        r = subprocess.run(["python", os.path.join(ROOT, "bert.cpp_amd", "examples", "client.py"), "--port", str(port),
                            "--encode", "hello world"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        import json
        assert np.allclose(np.array(json.loads(r.stdout), dtype=np.float32), got[0][0], atol=0, rtol=0)
        assert srv.poll() is None            # still serving after clients disconnected
    finally:
        srv.terminate()
        try:
            srv.wait(20)
        except subprocess.TimeoutExpired:
            srv.kill()


def test_scale_smoke_script_on_shared_gpu():
    """tools/scale_smoke.sh — the first command for an N-GPU node — at N = 2 on THIS box (BERT_BENCH_SHARED_GPU=1: the torchrun ranks
    share the GPU and exchange over gloo; the in-process arm runs at N = 1 with its exchange step on a 1-rank RCCL communicator): the
    two-rank arms print the digests of N = 1 and the script exits 0; with one rank's row corrupted it exits non-zero."""
    import subprocess
    env = dict(os.environ, BERT_BENCH_SHARED_GPU="1", STEPS="3", BERT_HIP_QUIET="1")
    env.pop("BERT_HIP_LATENCY", None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_smoke.sh"), "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "n=2" in r.stdout and "ranks_in_group=2" in r.stdout and "MISMATCH" not in r.stdout, r.stdout[-2000:]
    assert '"validation_only"' in r.stdout, r.stdout[-1500:]          # (bench.py's two-rank line says what it is)
    bad = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_smoke.sh"), "2"], cwd=ROOT, env=dict(env, SCALE_SMOKE_CORRUPT="1", STEPS="1"),
                         capture_output=True, text=True, timeout=900)
    assert bad.returncode != 0 and "MISMATCH" in bad.stdout, (bad.returncode, bad.stdout[-1500:])
