"""Host logic of the multi-GPU layer inside libbert.so (csrc/multi_device.cpp) and of the sentence windows of the fused
projection+attention kernel, through the test library — no GPU needed; and, on the GPU box (-m gpu), the same code with
real devices: sharded evaluation and the RCCL gather must give the bits of the single-device call."""
import numpy as np
import pytest

from bert_cpp_amd import dist as bdist
from bert_cpp_amd import ggml_file as gf
from bert_cpp_amd import pybert


def _cu(lens):
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


def test_shard_bounds_match_the_python_layer_and_balance_tokens():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 4, 8):
        for n in (1, 2, 7, 100, 1000):
            lens = rng.integers(1, 513, size=n).tolist()
            got = pybert.shard_bounds(_cu(lens), world)
            want = bdist.shard_bounds(lens, world)
            assert got == [want[0][0]] + [e for _, e in want], (world, n)
            assert got[0] == 0 and got[-1] == n and all(a <= b for a, b in zip(got, got[1:]))
            if n >= 4 * world:
                tok = [sum(lens[a:b]) for a, b in zip(got, got[1:])]
                assert max(tok) - min(tok) <= 2 * 512
    assert pybert.shard_bounds(_cu([128] * 1024), 8) == [i * 128 for i in range(9)]
    # a window of a larger batch (prefix sums not starting at 0): the dispatcher passes cu + first
    cu = _cu([5, 9, 100, 3, 64, 64, 7])
    assert pybert.shard_bounds(cu[2:], 2) == [b - 0 for b in pybert.shard_bounds(_cu([100, 3, 64, 64, 7]), 2)]


@pytest.mark.parametrize("slot", [16, 8])
def test_windows_hold_whole_sentences_in_order_and_fit(slot):
    """The host builder with 16-slot places (the default) and with 8-slot places (BERT_HIP_WINDOW_SLOTS=8: denser, but a
    sentence's bits then depend on its place — DESIGN.md §3)."""
    assert pybert.set_window_slots(slot) == slot
    try:
        rng = np.random.default_rng(1)
        total = 0
        for trial in range(50):
            n = int(rng.integers(1, 300))
            lens = np.clip(np.round(rng.lognormal(np.log(21.0), 0.7, n)), 1, 128).astype(int).tolist()
            win = pybert.build_windows(_cu(lens))
            total += len(win)
            assert win[0][0] == 0 and sum(c for _, c in win) == n
            nxt = 0
            for first, count in win:
                assert first == nxt and count >= 1
                nxt = first + count
                fill = 0
                for L in lens[first:first + count]:
                    assert fill + L <= 128                     # the sentence starts at a multiple of `slot` and fits
                    fill = (fill + L + slot - 1) // slot * slot
            # next-fit: a window is only closed when the next sentence does not fit
            for (f0, c0), (f1, _) in zip(win, win[1:]):
                fill = 0
                for L in lens[f0:f0 + c0]:
                    fill = (fill + L + slot - 1) // slot * slot
                assert fill + lens[f1] > 128
            assert len(win) <= pybert.max_windows(n, sum(lens)) <= n
        assert pybert.build_windows(_cu([128] * 5)) == [(i, 1) for i in range(5)]
        assert pybert.build_windows(_cu([16] * 8 + [1])) == [(0, 8), (8, 1)]
        assert pybert.build_windows(_cu([8] * 17)) == ([(0, 16), (16, 1)] if slot == 8 else [(0, 8), (8, 8), (16, 1)])
        _WINDOWS_BUILT[slot] = total
        if len(_WINDOWS_BUILT) == 2:
            assert _WINDOWS_BUILT[8] < 0.93 * _WINDOWS_BUILT[16]          # (mean ~25 tokens: an eighth fewer windows)
    finally:
        pybert.set_window_slots(16)


_WINDOWS_BUILT = {}


def test_the_grid_bound_of_device_built_windows_covers_every_packing():
    """The device API launches the fused attention kernel before it knows how many windows its own builder kernel makes: the
    grid is an upper bound computed from the sentence and token counts alone (two neighbouring next-fit windows hold more
    than 128 slots together).  It must cover the real count whatever the lengths are."""
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(1, 400))
        kind = trial % 4
        lens = (rng.integers(1, 129, n) if kind == 0 else rng.integers(1, 18, n) if kind == 1 else
                rng.choice([1, 15, 16, 17, 63, 64, 65, 112, 113, 128], n) if kind == 2 else
                np.clip(np.round(rng.lognormal(np.log(21.0), 0.7, n)), 1, 128).astype(int))
        cu = _cu([int(x) for x in lens])
        assert len(pybert.build_windows(cu)) <= pybert.max_windows(n, int(cu[-1])) <= n, (trial, n)


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_dispatcher_with_a_stub_evaluator(n_shards):
    """The code path of a multi-GPU bert_eval_batch (shard, one thread per shard, every shard writes the caller's rows)
    with a stub in place of the engines: every sentence is evaluated exactly once, by the shard that owns it."""
    rng = np.random.default_rng(n_shards)
    for n in (1, 2, 5, 64, 500):
        lens = rng.integers(1, 200, size=n)
        cu = _cu(lens)
        toks = rng.integers(0, 30000, size=int(cu[-1])).astype(np.int32)
        out = pybert.dispatch_stub(toks, cu, n_shards, H=4)
        bounds = pybert.shard_bounds(cu, n_shards)
        for b in range(n):
            shard = max(r for r in range(n_shards) if bounds[r] <= b and b < bounds[r + 1])
            want = [float(np.float32(toks[cu[b]:cu[b + 1]].astype(np.int64).sum())), float(lens[b]), float(shard), float(b)]
            assert out[b].tolist() == want, (n, b)


def test_dispatcher_keeps_its_threads_and_survives_exceptions():
    """A context's shard threads are created once: a thousand back-to-back calls create none (a forward pass takes under a
    millisecond; thread creation per call was the first suspect for sub-linear 8-GPU scaling).  An exception inside a
    shard — on a worker or on the caller's thread — comes back as an error code, and the pool keeps working."""
    lens = np.full(64, 32)
    cu = _cu(lens)
    toks = np.arange(int(cu[-1]), dtype=np.int32)
    want = pybert.dispatch_stub(toks, cu, 8)
    before = pybert.shard_threads_created()
    for _ in range(1000):
        assert np.array_equal(pybert.dispatch_stub(toks, cu, 8), want)
    assert pybert.shard_threads_created() == before
    with pytest.raises(RuntimeError, match="-9"):
        pybert.dispatch_stub(toks, cu, 8, throw=True)
    assert np.array_equal(pybert.dispatch_stub(toks, cu, 8), want)
    assert pybert.shard_threads_created() == before


# ------------------------------------------------------------------------------------------------ GPU
class _Hip:
    """Just enough of the HIP runtime through ctypes (the runtime libbert.so itself is linked against)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.lib = C.CDLL("libamdhip64.so")

    def malloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.lib.hipMalloc(self.C.byref(p), self.C.c_size_t(nbytes)) == 0
        return p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        assert self.lib.hipMemcpy(self.C.c_void_p(p), self.C.c_void_p(arr.ctypes.data), self.C.c_size_t(arr.nbytes), 1) == 0
        return p

    def download(self, p, shape, dtype=np.float32, device=None):
        if device is not None:
            assert self.lib.hipSetDevice(device) == 0
        out = np.empty(shape, dtype=dtype)
        assert self.lib.hipDeviceSynchronize() == 0
        assert self.lib.hipMemcpy(self.C.c_void_p(out.ctypes.data), self.C.c_void_p(p), self.C.c_size_t(out.nbytes), 2) == 0
        return out

    def stream(self):
        s = self.C.c_void_p()
        assert self.lib.hipStreamCreate(self.C.byref(s)) == 0
        return s.value


@pytest.mark.gpu
def test_gather_entry_point_matches_the_host_api(make_model, monkeypatch):
    """bert_hip_eval_packed_gather on the devices of this box (one on the test box): the device-resident matrix equals the
    host API's result bit for bit; with the option test_rccl_single the exchange runs through a 1-rank RCCL communicator."""
    hip = _Hip()
    path, hp = make_model("minilm-l6", "f16", 0)
    rng = np.random.default_rng(3)
    lens = rng.integers(1, 129, size=300)
    cu = _cu(lens)
    toks = rng.integers(1000, hp.n_vocab, size=int(cu[-1])).astype(np.int32)
    m = pybert.BertModel(path)
    want = m.eval_packed(toks, cu)
    for force in ("0", "1"):
        m.set_option("test_rccl_single", force)
        ptrs = m.eval_packed_gather(toks, cu)
        assert len(ptrs) == m.n_devices() >= 1
        for d, p in enumerate(ptrs):
            assert np.array_equal(hip.download(p, (len(lens), hp.n_embd), device=d), want), (force, d)
    # SUPER-BATCHES: with few tokens per run the call is cut into many runs, each sharded and exchanged on
    # its own — the exchange of run k on the exchange stream while run k + 1 computes, two shard buffers in turn.  Rows land
    # at their global positions: the same matrix, whatever the cut (1 / 2 / 3 / 7 / 19 runs here).
    for super_tokens in (100000, 10000, 7000, 3000, 1000):
        m.set_option("gather_super_tokens", str(super_tokens))
        ptrs = m.eval_packed_gather(toks, cu)
        for d, p in enumerate(ptrs):
            assert np.array_equal(hip.download(p, (len(lens), hp.n_embd), device=d), want), (super_tokens, d)
    m.set_option("gather_super_tokens", "0")
    # back-to-back calls reuse buffers, streams and events
    for _ in range(3):
        assert np.array_equal(hip.download(m.eval_packed_gather(toks, cu)[0], (len(lens), hp.n_embd), device=0), want)


@pytest.mark.gpu
@pytest.mark.parametrize("one_launch", ["1", "2"])
def test_device_api_guards(make_model, capfd, one_launch):
    """bert_hip_eval_packed_device trusts max_len for kernel selection; a batch that breaks the promise must not produce
    silent garbage: NaN rows for the offending sentences and bert_hip_check() == 1 — on the default route and with all
    layers in one launch whatever the windows' fill (one_launch=2: model_kernel.hip on ragged windows)."""
    hip = _Hip()
    path, hp = make_model("minilm-l6", "f16", 0)
    m = pybert.BertModel(path)
    m.set_option("one_launch", one_launch)
    lens = [20, 100, 64, 7]
    cu = _cu(lens)
    T, H = int(cu[-1]), hp.n_embd
    toks = np.random.default_rng(0).integers(1000, hp.n_vocab, size=T).astype(np.int32)
    want = m.eval_packed(toks, cu)
    d_t, d_cu = hip.upload(toks), hip.upload(cu)
    out = hip.upload(np.full((4, H), 7.0, np.float32))
    m.reserve(T, 4)
    m.eval_packed_device(d_t, d_cu, 4, T, 128, out, 0)
    assert m.check() == 0
    assert np.array_equal(hip.download(out, (4, H)), want)
    # promise 64, deliver 100: sentence 1 is flagged, the others are unaffected
    m.eval_packed_device(d_t, d_cu, 4, T, 64, out, 0)
    assert m.check() == 1 and m.check() == 0
    got = hip.download(out, (4, H))
    assert np.isnan(got[1]).all() and np.array_equal(got[[0, 2, 3]], want[[0, 2, 3]])
    assert "max_len" in capfd.readouterr().err
    # a max_len that cannot hold the tokens at all is refused on the host
    with pytest.raises(RuntimeError):
        m.eval_packed_device(d_t, d_cu, 4, T, 16, out, 0)
    # promise 128, deliver 200 (longer than a window): the same
    lens2 = [20, 200, 64, 7, 128, 90]
    cu2 = _cu(lens2)
    toks2 = np.random.default_rng(1).integers(1000, hp.n_vocab, size=int(cu2[-1])).astype(np.int32)
    want2 = m.eval_batch([toks2[cu2[i]:cu2[i + 1]] for i in (0, 2, 3, 4, 5)])
    out2 = hip.upload(np.full((6, H), 7.0, np.float32))
    m.eval_packed_device(hip.upload(toks2), hip.upload(cu2), 6, int(cu2[-1]), 128, out2, 0)
    assert m.check() == 1
    got2 = hip.download(out2, (6, H))
    assert np.isnan(got2[1]).all() and np.array_equal(got2[[0, 2, 3, 4, 5]], want2)
    capfd.readouterr()
    # a batch that LOOKS like full windows (T = 128 B, max_len = 128) but is not: the form specialised for full windows works
    # 128-token block by block — the sentences that are exactly their block keep their bits, every other one (the offender and
    # the neighbours it shifts across block boundaries) gets a NaN row, none gets numbers made of a neighbour's tokens
    lens3 = [128, 128, 200, 56, 128]
    cu3 = _cu(lens3)
    assert int(cu3[-1]) == 128 * len(lens3)
    toks3 = np.random.default_rng(2).integers(1000, hp.n_vocab, size=int(cu3[-1])).astype(np.int32)
    want3 = m.eval_batch([toks3[cu3[i]:cu3[i + 1]] for i in (0, 1, 4)])
    out3 = hip.upload(np.full((5, H), 7.0, np.float32))
    m.eval_packed_device(hip.upload(toks3), hip.upload(cu3), 5, int(cu3[-1]), 128, out3, 0)
    assert m.check() == 1
    got3 = hip.download(out3, (5, H))
    assert np.isnan(got3[2]).all() and np.isnan(got3[3]).all() and np.array_equal(got3[[0, 1, 4]], want3)
    capfd.readouterr()
    # two streams, back to back: the context serialises its passes itself
    s1, s2 = hip.stream(), hip.stream()
    o1, o2 = hip.malloc(4 * H * 4), hip.malloc(4 * H * 4)
    for _ in range(5):
        m.eval_packed_device(d_t, d_cu, 4, T, 128, o1, s1)
        m.eval_packed_device(d_t, d_cu, 4, T, 128, o2, s2)
    assert np.array_equal(hip.download(o1, (4, H)), want) and np.array_equal(hip.download(o2, (4, H)), want)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,ftype", [("minilm-l6", "f16"), ("minilm-l12", "q4_1"), ("h256-l3", "f16")])
def test_one_launch_on_ragged_windows(make_model, dims, ftype):
    """model_kernel.hip also takes windows that are not full (whole sentences, at most 128 tokens between them: its layer-tail
    phase runs on the window's rows of the packed batch, the stores behind them dropped by a buffer descriptor's range
    check).  Same bits as two launches per layer, with the caller's window list (host API), the list built on the device and
    one sentence per window.  The engine takes the route by itself only for well-filled windows (the tail phase costs 128
    rows' time whatever the window holds: 1.15 M against 1.49 M sentences/s on the bench's mixed-length batch), one_launch=2
    forces it."""
    gf.MODEL_DIMS.setdefault("h256-l3", gf.BertHParams(1000, 128, 256, 1024, 8, 3))
    hip = _Hip()
    path, hp = make_model(dims, ftype, 0)
    m = pybert.BertModel(path)
    m.set_option("latency", "0")
    rng = np.random.default_rng(5)
    cases = {"mixed": rng.integers(1, 129, size=300), "ones": np.ones(70, dtype=np.int64), "16s": np.full(40, 16), "17s": np.full(33, 17),
             "two": np.array([128, 3]), "max 64": rng.integers(40, 65, size=50), "32s": np.full(64, 32), "one": np.array([5]),
             "127s": np.full(9, 127), "128s": np.full(7, 128), "many": rng.integers(3, 129, size=2000)}
    H = hp.n_embd
    for name, lens in cases.items():
        cu = _cu(lens)
        T, B, ml = int(cu[-1]), len(lens), int(np.max(lens))
        toks = rng.integers(0, hp.n_vocab, size=T).astype(np.int32)
        m.set_option("one_launch", "0")
        want = m.eval_packed(toks, cu)
        m.set_option("one_launch", "2")
        m.profile(True)
        got = m.eval_packed(toks, cu)
        names = set(m.profile_report())
        m.profile(False)
        assert "model_kernel" in names and not {"qkv_attention2", "layer_tail"} & names, (name, names)
        assert np.array_equal(got, want), (name, "host windows")
        d_t, d_cu, out = hip.upload(toks), hip.upload(cu), hip.upload(np.full((B, H), 7.0, np.float32))
        m.reserve(T, B)
        for promised in (ml, 128):                          # (windows built on the device, or one sentence per window)
            m.profile(True)
            m.eval_packed_device(d_t, d_cu, B, T, promised, out, 0)
            got = hip.download(out, (B, H))
            names = set(m.profile_report())
            m.profile(False)
            assert m.check() == 0 and "model_kernel" in names and not {"qkv_attention2", "layer_tail"} & names, (name, promised, names)
            assert np.array_equal(got, want), (name, promised)
    # by itself: windows filled to 95 % and more
    lens = rng.integers(123, 129, size=40)
    cu = _cu(lens)
    toks = rng.integers(0, hp.n_vocab, size=int(cu[-1])).astype(np.int32)
    m.set_option("one_launch", "0")
    want = m.eval_packed(toks, cu)
    m.set_option("one_launch", "1")
    m.profile(True)
    got = m.eval_packed(toks, cu)
    names = set(m.profile_report())
    m.profile(False)
    assert "model_kernel" in names and np.array_equal(got, want), names
    short = [toks[cu[i]:cu[i + 1]][:90] for i in range(20)]                      # fill 0.70: two launches per layer
    m.profile(True)
    got = m.eval_batch(short)
    names = set(m.profile_report())
    m.profile(False)
    assert "model_kernel" not in names and {"qkv_attention2", "layer_tail"} <= names, names


@pytest.mark.gpu
@pytest.mark.parametrize("mean_len", [6, 25, 90])
def test_device_api_packs_short_sentences_like_the_host_api(make_model, mean_len):
    """The asynchronous device API sees the lengths only in HBM: it builds the 128-slot windows with a kernel (when the batch
    is short enough for packing to pay) and must give the bits of the host API, which builds them on the CPU."""
    hip = _Hip()
    path, hp = make_model("minilm-l6", "f16", 0)
    m = pybert.BertModel(path)
    rng = np.random.default_rng(mean_len)
    lens = np.clip(rng.gamma(2.0, mean_len / 2.0, 1500).astype(np.int64) + 1, 1, 128)
    cu = _cu(lens)
    T, H = int(cu[-1]), hp.n_embd
    toks = rng.integers(1000, hp.n_vocab, size=T).astype(np.int32)
    want = m.eval_packed(toks, cu)
    d_t, d_cu = hip.upload(toks), hip.upload(cu)
    out = hip.upload(np.full((len(lens), H), 7.0, np.float32))
    m.reserve(T, len(lens))
    for _ in range(3):                                      # (the window list is rebuilt by every pass)
        m.eval_packed_device(d_t, d_cu, len(lens), T, 128, out, 0)
    assert m.check() == 0
    assert np.array_equal(hip.download(out, (len(lens), H)), want)


@pytest.mark.gpu
def test_no_exception_crosses_the_abi(make_model, capfd, monkeypatch):
    path, hp = make_model("tiny", "f16", 1)
    s = np.arange(5, dtype=np.int32)
    m = pybert.BertModel(path)
    m.set_option("test_inject_bad_alloc", "1")
    out = m.eval_batch([s, s])
    assert np.isnan(out).all()                              # outputs untouched, the process is alive
    assert "bert_eval_batch: std::bad_alloc" in capfd.readouterr().err
    n, out = m.encode_batch_count(["a b", "c"])
    assert n == -1 or n == 0
    assert np.isnan(out).all()
    assert np.isfinite(pybert.BertModel(path).eval_batch([s])).all()


@pytest.mark.gpu
def test_gemm_option_toggled_after_load(make_model, capfd, monkeypatch):
    """set_option("gemm", "naive") after a default load used to make the generic kernel read a null image: it
    is refused now unless the images were built at load time; with them both families agree."""
    path, hp = make_model("tiny-h128", "f16", 1)
    s = [np.random.default_rng(0).integers(0, hp.n_vocab, size=n).astype(np.int32) for n in (9, 40, 64)]
    m = pybert.BertModel(path, test_routes=True)
    base = m.eval_batch(s)
    m.set_option("gemm", "naive")
    assert "ignored" in capfd.readouterr().err
    assert np.array_equal(m.eval_batch(s), base)
    # the product library does not know the route at all (a test cross-check: libbert_test.so's engine only)
    monkeypatch.setenv("BERT_HIP_KERNELS", "naive")
    mp = pybert.BertModel(path)
    assert "not a route of libbert.so" in capfd.readouterr().err
    mp.set_option("attn", "naive")
    assert "not a route of libbert.so" in capfd.readouterr().err
    assert np.array_equal(mp.eval_batch(s), base)
    m2 = pybert.BertModel(path, test_routes=True)
    monkeypatch.delenv("BERT_HIP_KERNELS")
    naive = m2.eval_batch(s)
    m2.set_option("gemm", "mfma")
    m2.set_option("attn", "mfma")
    fast = m2.eval_batch(s)
    m2.set_option("gemm", "naive")
    m2.set_option("attn", "naive")
    assert np.array_equal(m2.eval_batch(s), naive)
    for a, b in zip(naive, fast):
        assert float(a @ b) > 1 - 1e-5


def _rank_worker(rank, world, port, model_path, seed, out_dir):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BERT_HIP_DEVICES"] = "0"               # every rank on the one GPU of the test box
    os.environ["BERT_HIP_QUIET"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = pybert.BertModel(model_path)
        rng = np.random.default_rng(seed)              # same sentences on every rank
        sents = [rng.integers(1000, 30000, size=int(n)).astype(np.int32) for n in rng.integers(1, 129, size=37)]
        emb = bdist.encode_sharded(lambda ss: m.eval_batch(ss), sents)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), emb.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "ref.npy"), m.eval_batch(sents))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_path_with_the_hip_engine(make_model, tmp_path, world):
    """The multi-process layer (bert.cpp_amd/dist.py: shard by tokens, evaluate, ONE all-gather) with the real HIP engine
    in every rank — the ranks share the box's GPU and gather over gloo; on a multi-GPU node the same code runs one rank per
    GPU over RCCL (bench.py).  Every rank must hold the bits of the single-process evaluation."""
    import socket

    import torch.multiprocessing as mp

    path, hp = make_model("minilm-l6", "f16", 0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_worker, args=(world, port, path, 11, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"rank{r}.npy"), ref), r


@pytest.mark.gpu
def test_device_selection_environment(make_model, monkeypatch, capfd):
    """BERT_HIP_DEVICES names the context's GPUs; BERT_HIP_DEVICE=<n> (the single-device spelling of the first builds) is read
    as a list of one when BERT_HIP_DEVICES is unset; a bad ordinal fails the load loudly instead of landing on device 0."""
    path, hp = make_model("tiny", "f16", 0)
    monkeypatch.delenv("BERT_HIP_DEVICES", raising=False)
    monkeypatch.setenv("BERT_HIP_DEVICE", "0")
    m = pybert.BertModel(path)
    assert m.n_devices() == 1 and m.lib.bert_hip_device(m.ctx) == 0
    m.close()
    monkeypatch.setenv("BERT_HIP_DEVICE", "63")
    with pytest.raises(RuntimeError):
        pybert.BertModel(path)
    assert "out of range" in capfd.readouterr().err
    monkeypatch.setenv("BERT_HIP_DEVICES", "0")          # (the list wins over the alias)
    m = pybert.BertModel(path)
    assert m.n_devices() == 1
    m.close()
