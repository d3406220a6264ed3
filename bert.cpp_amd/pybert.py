"""ctypes binding of libbert.so — the same way the reference's Python callers bind it
(reference examples/sample_dylib.py:19-59, benchmarks/run_mteb.py:34-72), plus the bert_hip.h
extensions.  There is no fallback: if the shared library (the HIP extension) is missing or does
not load, importing/constructing fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BERT_HIP_LIB") or os.path.join(_HERE, "libbert.so")   # BERT_HIP_LIB: tuning builds

# every symbol include/bert.h and include/bert_hip.h declare
BERT_H_SYMBOLS = [
    "bert_params_parse", "bert_load_from_file", "bert_free", "bert_encode", "bert_encode_batch", "bert_tokenize",
    "bert_eval", "bert_eval_batch", "bert_n_embd", "bert_n_max_tokens", "bert_vocab_id_to_token",
]
BERT_HIP_H_SYMBOLS = [
    "bert_hip_load_tokenizer", "bert_hip_tokenize_batch", "bert_hip_n_layer", "bert_hip_n_head", "bert_hip_n_intermediate", "bert_hip_n_vocab",
    "bert_hip_ftype", "bert_hip_device", "bert_hip_n_devices", "bert_hip_encode_batch", "bert_hip_eval_packed", "bert_hip_eval_packed_gather",
    "bert_hip_eval_packed_device", "bert_hip_reserve", "bert_hip_check", "bert_hip_eval_hidden",
    "bert_hip_profile_enable", "bert_hip_profile_report", "bert_hip_set_option", "bert_hip_version",
]
# include/bert_hip_test.h: the op-level test hooks, exported by libbert_test.so only
BERT_HIP_TEST_H_SYMBOLS = [
    "bert_hip_test_gemm", "bert_hip_test_gemm_lnfold", "bert_hip_test_attention", "bert_hip_test_qkv_attention",
    "bert_hip_test_layer_tail", "bert_hip_test_shard_bounds", "bert_hip_test_build_windows",
    "bert_hip_test_build_windows_device", "bert_hip_test_max_windows", "bert_hip_test_set_window_slots",
    "bert_hip_test_dispatch", "bert_hip_test_shard_threads_created", "bert_hip_test_embed_ln", "bert_hip_test_pool_normalize",
    "bert_hip_test_model_digest",
]
TEST_LIB_PATH = LIB_PATH[:-3] + "_test.so"


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile libbert.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", _HERE, f"-j{jobs}", "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def _declare_product_abi(L):
    """argument / result types of include/bert.h + include/bert_hip.h (libbert.so; libbert_test.so holds the same entry points)"""
    vp, i32, f32p, i32p = C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.bert_load_from_file.restype = vp; L.bert_load_from_file.argtypes = [C.c_char_p]
    L.bert_hip_load_tokenizer.restype = vp; L.bert_hip_load_tokenizer.argtypes = [C.c_char_p]
    L.bert_free.restype = None; L.bert_free.argtypes = [vp]
    for fn in ("bert_n_embd", "bert_n_max_tokens", "bert_hip_n_layer", "bert_hip_n_head", "bert_hip_n_intermediate",
               "bert_hip_n_vocab", "bert_hip_ftype", "bert_hip_device"):
        getattr(L, fn).restype = i32; getattr(L, fn).argtypes = [vp]
    L.bert_vocab_id_to_token.restype = C.c_char_p; L.bert_vocab_id_to_token.argtypes = [vp, i32]
    L.bert_tokenize.restype = None; L.bert_tokenize.argtypes = [vp, C.c_char_p, i32p, i32p, i32]
    L.bert_eval.restype = None; L.bert_eval.argtypes = [vp, i32, i32p, i32, f32p]
    L.bert_eval_batch.restype = None
    L.bert_eval_batch.argtypes = [vp, i32, i32, C.POINTER(i32p), i32p, C.POINTER(f32p)]
    L.bert_encode.restype = None; L.bert_encode.argtypes = [vp, i32, C.c_char_p, f32p]
    L.bert_encode_batch.restype = None
    L.bert_encode_batch.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_char_p), C.POINTER(f32p)]
    L.bert_hip_eval_packed.restype = i32; L.bert_hip_eval_packed.argtypes = [vp, i32p, i32p, i32, f32p]
    L.bert_hip_eval_packed_device.restype = i32
    L.bert_hip_eval_packed_device.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    L.bert_hip_eval_hidden.restype = i32; L.bert_hip_eval_hidden.argtypes = [vp, i32p, i32, f32p, f32p]
    L.bert_hip_profile_enable.restype = None; L.bert_hip_profile_enable.argtypes = [vp, i32]
    L.bert_hip_profile_report.restype = i32; L.bert_hip_profile_report.argtypes = [vp, C.c_char_p, i32]
    L.bert_hip_set_option.restype = None; L.bert_hip_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.bert_hip_n_devices.restype = i32; L.bert_hip_n_devices.argtypes = [vp]
    L.bert_hip_encode_batch.restype = i32
    L.bert_hip_encode_batch.argtypes = [vp, i32, i32, C.POINTER(C.c_char_p), C.POINTER(f32p)]
    L.bert_hip_eval_packed_gather.restype = i32
    L.bert_hip_eval_packed_gather.argtypes = [vp, i32p, i32p, i32, C.POINTER(vp)]
    L.bert_hip_reserve.restype = i32; L.bert_hip_reserve.argtypes = [vp, i32, i32]
    L.bert_hip_check.restype = i32; L.bert_hip_check.argtypes = [vp]
    L.bert_hip_tokenize_batch.restype = i32
    L.bert_hip_tokenize_batch.argtypes = [vp, i32, i32, C.POINTER(C.c_char_p), i32p, i32p]
    L.bert_hip_version.restype = C.c_char_p


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build the HIP extension first "
                           f"(python -c 'import __graft_entry__ as g; g.build()' or make -C bert.cpp_amd)")
    L = C.CDLL(LIB_PATH)
    _declare_product_abi(L)
    _lib = L
    return L


_test_lib = None


def test_lib() -> C.CDLL:
    """libbert_test.so: the product's objects plus the op-level test hooks of include/bert_hip_test.h."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    if not os.path.exists(TEST_LIB_PATH):
        raise RuntimeError(f"{TEST_LIB_PATH} not found: build it first (make -C bert.cpp_amd)")
    L = C.CDLL(TEST_LIB_PATH)
    _declare_product_abi(L)
    vp, i32, i32p = C.c_void_p, C.c_int32, C.POINTER(C.c_int32)
    L.bert_hip_test_gemm.restype = i32
    L.bert_hip_test_gemm.argtypes = [i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, vp]
    L.bert_hip_test_gemm_lnfold.restype = i32
    L.bert_hip_test_gemm_lnfold.argtypes = [i32, i32, i32, i32] + [vp] * 10 + [i32, vp, vp, vp]
    L.bert_hip_test_attention.restype = i32
    L.bert_hip_test_attention.argtypes = [i32, i32p, i32, i32, vp, i32, vp]
    L.bert_hip_test_qkv_attention.restype = i32
    L.bert_hip_test_qkv_attention.argtypes = [i32, i32p, i32, i32, vp, vp, i32, vp, i32, vp]
    L.bert_hip_test_layer_tail.restype = i32
    L.bert_hip_test_layer_tail.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    L.bert_hip_test_embed_ln.restype = i32
    L.bert_hip_test_embed_ln.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, i32p, i32p, i32, vp]
    L.bert_hip_test_pool_normalize.restype = i32
    L.bert_hip_test_pool_normalize.argtypes = [i32, vp, i32p, i32, i32, vp, i32p]
    L.bert_hip_test_model_digest.restype = i32
    L.bert_hip_test_model_digest.argtypes = [C.c_char_p, i32p, C.POINTER(C.c_uint64)]
    L.bert_hip_test_shard_bounds.restype = None
    L.bert_hip_test_shard_bounds.argtypes = [i32p, i32, i32, i32p]
    L.bert_hip_test_build_windows.restype = i32
    L.bert_hip_test_build_windows.argtypes = [i32p, i32, i32p]
    L.bert_hip_test_max_windows.restype = i32
    L.bert_hip_test_max_windows.argtypes = [i32, i32]
    L.bert_hip_test_set_window_slots.restype = i32
    L.bert_hip_test_set_window_slots.argtypes = [i32]
    L.bert_hip_test_build_windows_device.restype = i32
    L.bert_hip_test_build_windows_device.argtypes = [i32p, i32, i32p]
    L.bert_hip_test_shard_threads_created.restype = C.c_int64
    L.bert_hip_test_shard_threads_created.argtypes = []
    L.bert_hip_test_dispatch.restype = i32
    L.bert_hip_test_dispatch.argtypes = [i32p, i32p, i32, i32, i32, C.POINTER(C.c_float)]
    _test_lib = L
    return L


def test_embed_ln(table_type: int, word_bytes, type_bytes, pos_bytes, H: int, gamma, beta, tokens, cu_seqlens) -> np.ndarray:
    wb, tb, pb = (np.ascontiguousarray(a) for a in (word_bytes, type_bytes, pos_bytes))
    rb = {0: 4 * H, 1: 2 * H, 2: H // 32 * 18, 3: H // 32 * 20}[table_type]
    g = np.ascontiguousarray(gamma, dtype=np.float32); b = np.ascontiguousarray(beta, dtype=np.float32)
    toks = np.ascontiguousarray(tokens, dtype=np.int32); cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    out = np.zeros((len(toks), H), dtype=np.float16)
    r = test_lib().bert_hip_test_embed_ln(table_type, H, wb.nbytes // rb, pb.nbytes // rb, wb.ctypes.data, tb.ctypes.data, pb.ctypes.data,
                                          g.ctypes.data, b.ctypes.data, _i32p(toks), _i32p(cu), len(cu) - 1, out.ctypes.data)
    if r != 0:
        raise RuntimeError(f"bert_hip_test_embed_ln failed: {r}")
    return out


def test_pool_normalize(x: np.ndarray, cu_seqlens, max_len: int):
    x = np.ascontiguousarray(x, dtype=np.float16); cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    out = np.zeros((len(cu) - 1, x.shape[1]), dtype=np.float32)
    st = np.zeros(1, dtype=np.int32)
    r = test_lib().bert_hip_test_pool_normalize(x.shape[1], x.ctypes.data, _i32p(cu), len(cu) - 1, max_len, out.ctypes.data, _i32p(st))
    if r != 0:
        raise RuntimeError(f"bert_hip_test_pool_normalize failed: {r}")
    return out, int(st[0])


def model_digest(path: str):
    """(n_tensors, legacy_q4, digest) of a model file as the product's parser sees it (no GPU needed)."""
    leg = C.c_int32(0)
    dig = C.c_uint64(0)
    n = test_lib().bert_hip_test_model_digest(path.encode(), C.byref(leg), C.byref(dig))
    if n < 0:
        raise RuntimeError("model file rejected (see stderr)")
    return n, bool(leg.value), int(dig.value)


def shard_bounds(cu_seqlens: np.ndarray, n_shards: int) -> List[int]:
    cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    out = np.zeros(n_shards + 1, dtype=np.int32)
    test_lib().bert_hip_test_shard_bounds(_i32p(cu), len(cu) - 1, n_shards, _i32p(out))
    return out.tolist()


def build_windows(cu_seqlens: np.ndarray, device: bool = False) -> List[tuple]:
    """{first sentence, count} windows of 128 token slots: the host builder, or (device=True, needs a GPU) the kernel."""
    cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    out = np.zeros(2 * max(len(cu) - 1, 1), dtype=np.int32)
    f = test_lib().bert_hip_test_build_windows_device if device else test_lib().bert_hip_test_build_windows
    n = f(_i32p(cu), len(cu) - 1, _i32p(out))
    if n < 0:
        raise RuntimeError("bert_hip_test_build_windows_device failed")
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def set_window_slots(slots: int) -> int:
    """Place granularity of the windows in libbert_test.so (its own copy of the setting; a context's: set_option "window_slots")."""
    return int(test_lib().bert_hip_test_set_window_slots(slots))


def max_windows(n_sentences: int, n_tokens: int) -> int:
    return int(test_lib().bert_hip_test_max_windows(n_sentences, n_tokens))


def dispatch_stub(tokens: np.ndarray, cu_seqlens: np.ndarray, n_shards: int, H: int = 4, throw: bool = False) -> np.ndarray:
    tokens = np.ascontiguousarray(tokens, dtype=np.int32)
    cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    out = np.full((len(cu) - 1, abs(H)), np.nan, dtype=np.float32)
    r = test_lib().bert_hip_test_dispatch(_i32p(tokens), _i32p(cu), len(cu) - 1, n_shards, -H if throw else H, _f32p(out))
    if r != 0:
        raise RuntimeError(f"bert_hip_test_dispatch failed: {r}")
    return out


def shard_threads_created() -> int:
    return int(test_lib().bert_hip_test_shard_threads_created())


def _f32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class BertModel:
    """Mirror of the reference's Python `BertModel` wrapper (examples/sample_dylib.py:12-59)."""

    def __init__(self, fname: str, tokenizer_only: bool = False, test_routes: bool = False):
        # test_routes: the context lives in libbert_test.so, whose engine also understands the whole-model "naive" cross-check route
        self.lib = test_lib() if test_routes else lib()
        load = self.lib.bert_hip_load_tokenizer if tokenizer_only else self.lib.bert_load_from_file
        self.ctx = load(fname.encode("utf-8"))
        if not self.ctx:
            raise RuntimeError(f"bert_load_from_file('{fname}') failed (see stderr)")
        self.n_embd = self.lib.bert_n_embd(self.ctx)
        self.n_max_tokens = self.lib.bert_n_max_tokens(self.ctx)
        self.n_layer = self.lib.bert_hip_n_layer(self.ctx)
        self.n_vocab = self.lib.bert_hip_n_vocab(self.ctx)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.bert_free(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- bert.h ---------------------------------------------------------------------------
    def tokenize(self, text: str | bytes, n_max_tokens: Optional[int] = None) -> List[int]:
        n_max = n_max_tokens or self.n_max_tokens
        buf = (C.c_int32 * max(n_max, 2))()
        n = C.c_int32(0)
        data = text if isinstance(text, bytes) else text.encode("utf-8")
        self.lib.bert_tokenize(self.ctx, data, buf, C.byref(n), n_max)
        return list(buf[: n.value])

    def tokenize_batch(self, texts: Sequence[str | bytes], n_threads: int = 6) -> List[List[int]]:
        n, N = len(texts), self.n_max_tokens
        arr = (C.c_char_p * n)(*[t if isinstance(t, bytes) else t.encode("utf-8") for t in texts])
        toks = np.zeros((n, N), dtype=np.int32)
        cnt = np.zeros(n, dtype=np.int32)
        if self.lib.bert_hip_tokenize_batch(self.ctx, n_threads, n, arr, _i32p(toks), _i32p(cnt)) != 0:
            raise RuntimeError("bert_hip_tokenize_batch failed")
        return [toks[i, : cnt[i]].tolist() for i in range(n)]

    def id_to_token(self, i: int) -> bytes:
        return self.lib.bert_vocab_id_to_token(self.ctx, i)

    def eval(self, tokens: Sequence[int], n_threads: int = 6) -> np.ndarray:
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.full(self.n_embd, np.nan, dtype=np.float32)
        self.lib.bert_eval(self.ctx, n_threads, _i32p(toks), len(toks), _f32p(out))
        return out

    def eval_batch(self, sentences: Sequence[Sequence[int]], n_threads: int = 6, scattered: bool = False) -> np.ndarray:
        """bert_eval_batch through per-sentence host pointers, exactly like a C caller.  scattered: the result rows are
        every other row of a wider matrix (rows that are NOT the rows of one [B][n_embd] matrix)."""
        B = len(sentences)
        arrs = [np.ascontiguousarray(s, dtype=np.int32) for s in sentences]
        wide = np.full((B, 2 * self.n_embd + 3 if scattered else self.n_embd), np.nan, dtype=np.float32)
        out = wide[:, : self.n_embd]
        tok_ptrs = (C.POINTER(C.c_int32) * B)(*[_i32p(a) for a in arrs])
        lens = np.array([len(a) for a in arrs], dtype=np.int32)
        out_ptrs = (C.POINTER(C.c_float) * B)(*[C.cast(wide[i].ctypes.data, C.POINTER(C.c_float)) for i in range(B)])
        self.lib.bert_eval_batch(self.ctx, n_threads, B, tok_ptrs, _i32p(lens), out_ptrs)
        if scattered:
            assert np.isnan(wide[:, self.n_embd:]).all()          # nothing written beside the rows
        return np.ascontiguousarray(out)

    def encode(self, text: str, n_threads: int = 6) -> np.ndarray:
        out = np.full(self.n_embd, np.nan, dtype=np.float32)
        self.lib.bert_encode(self.ctx, n_threads, text.encode("utf-8"), _f32p(out))
        return out

    def encode_batch(self, texts: Sequence[str], n_threads: int = 6, batch_size: int = 16) -> np.ndarray:
        n = len(texts)
        out = np.full((n, self.n_embd), np.nan, dtype=np.float32)
        out_ptrs = (C.POINTER(C.c_float) * n)(*[_f32p(out[i]) for i in range(n)])
        txt = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        self.lib.bert_encode_batch(self.ctx, n_threads, batch_size, n, txt, out_ptrs)
        return out

    # ---- bert_hip.h -----------------------------------------------------------------------
    def eval_packed(self, tokens: np.ndarray, cu_seqlens: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        """out: the caller's result rows, as in the C ABI (bert.h: `float **batch_embeddings`); a fresh NaN-filled array if None."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        B = len(cu) - 1
        if out is None:
            out = np.full((B, self.n_embd), np.nan, dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (B, self.n_embd)
        r = self.lib.bert_hip_eval_packed(self.ctx, _i32p(tokens), _i32p(cu), B, _f32p(out))
        if r != 0:
            raise RuntimeError(f"bert_hip_eval_packed failed: {r}")
        return out

    def n_devices(self) -> int:
        return self.lib.bert_hip_n_devices(self.ctx)

    def eval_packed_gather(self, tokens: np.ndarray, cu_seqlens: np.ndarray) -> List[int]:
        """Evaluates on all devices of the context and gathers on every device: returns the device pointers of the
        [n_sentences][n_embd] f32 matrices, one per device (owned by the context)."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        ptrs = (C.c_void_p * self.n_devices())()
        r = self.lib.bert_hip_eval_packed_gather(self.ctx, _i32p(tokens), _i32p(cu), len(cu) - 1, ptrs)
        if r != 0:
            raise RuntimeError(f"bert_hip_eval_packed_gather failed: {r}")
        return [int(p) for p in ptrs]

    def reserve(self, n_tokens: int, n_sentences: int) -> None:
        if self.lib.bert_hip_reserve(self.ctx, n_tokens, n_sentences) != 0:
            raise RuntimeError("bert_hip_reserve failed")

    def check(self) -> int:
        return self.lib.bert_hip_check(self.ctx)

    def encode_batch_count(self, texts: Sequence[str], n_threads: int = 6):
        n = len(texts)
        out = np.full((n, self.n_embd), np.nan, dtype=np.float32)
        out_ptrs = (C.POINTER(C.c_float) * n)(*[_f32p(out[i]) for i in range(n)])
        txt = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        return self.lib.bert_hip_encode_batch(self.ctx, n_threads, n, txt, out_ptrs), out

    def eval_packed_device(self, d_tokens_ptr: int, d_cu_ptr: int, n_sentences: int, n_tokens: int, max_len: int,
                           d_out_ptr: int, stream: int = 0) -> None:
        r = self.lib.bert_hip_eval_packed_device(self.ctx, d_tokens_ptr, d_cu_ptr, n_sentences, n_tokens, max_len,
                                                 d_out_ptr, stream)
        if r != 0:
            raise RuntimeError(f"bert_hip_eval_packed_device failed: {r}")

    def eval_hidden(self, tokens: Sequence[int]):
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        hid = np.empty((self.n_layer + 1, len(toks), self.n_embd), dtype=np.float32)
        out = np.empty(self.n_embd, dtype=np.float32)
        r = self.lib.bert_hip_eval_hidden(self.ctx, _i32p(toks), len(toks), _f32p(hid), _f32p(out))
        if r != 0:
            raise RuntimeError(f"bert_hip_eval_hidden failed: {r}")
        return out, hid

    def profile(self, on: bool) -> None:
        self.lib.bert_hip_profile_enable(self.ctx, int(on))

    def profile_report(self, families: bool = False) -> dict:
        """{kernel: {launches, total_ms, flops_per_launch}}; families=True adds the "family:<kernel>_<weights>" lines (which
        mat-mul kernel served the launches: gemm256_f16 / gemm256_q4 / gemm_mfma_f16 / gemm_mfma_q4 / gemm_naive)."""
        buf = C.create_string_buffer(1 << 16)
        self.lib.bert_hip_profile_report(self.ctx, buf, len(buf))
        out = {}
        for line in buf.value.decode().splitlines():
            name, launches, ms, flops = line.split()
            if families or not name.startswith("family:"):
                out[name] = {"launches": int(launches), "total_ms": float(ms), "flops_per_launch": float(flops)}
        return out

    def set_option(self, key: str, value: str) -> None:
        self.lib.bert_hip_set_option(self.ctx, key.encode(), value.encode())


def test_gemm(A: np.ndarray, W_bytes: np.ndarray, wtype: int, N: int, bias: np.ndarray,
              resid: Optional[np.ndarray], epilogue: int, impl: int) -> np.ndarray:
    """A: float16 [M, K]; W_bytes: file-layout bytes of W[N][K]; returns float16 [M, N]."""
    L = test_lib()
    A = np.ascontiguousarray(A, dtype=np.float16)
    M, K = A.shape
    Wb = np.ascontiguousarray(W_bytes)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    out = np.zeros((M, N), dtype=np.float16)
    rp = None
    if resid is not None:
        resid = np.ascontiguousarray(resid, dtype=np.float16)
        rp = resid.ctypes.data
    r = L.bert_hip_test_gemm(M, N, K, A.ctypes.data, Wb.ctypes.data, wtype, bias.ctypes.data, rp, epilogue, impl,
                             out.ctypes.data)
    if r != 0:
        raise RuntimeError(f"bert_hip_test_gemm failed: {r}")
    return out


def test_gemm_lnfold(A1, W1, b1, r, rg, rb, W2, b2, g, be, epi2):
    """The LayerNorm-folded mat-mul pair of the H = 768 route (include/bert_hip_test.h): returns (u [M,H] f16, out2 [M,N2] f16, rows [M,4] f32)."""
    L = test_lib()
    A1 = np.ascontiguousarray(A1, dtype=np.float16); W1 = np.ascontiguousarray(W1, dtype=np.float16); W2 = np.ascontiguousarray(W2, dtype=np.float16)
    r = np.ascontiguousarray(r, dtype=np.float16)
    f = lambda v: None if v is None else np.ascontiguousarray(v, dtype=np.float32)
    b1, rg, rb, b2, g, be = f(b1), f(rg), f(rb), f(b2), f(g), f(be)
    M, K1 = A1.shape
    H, N2 = W1.shape[0], W2.shape[0]
    u = np.zeros((M, H), dtype=np.float16); out = np.zeros((M, N2), dtype=np.float16); rows = np.zeros((M, 4), dtype=np.float32)
    p = lambda v: None if v is None else v.ctypes.data
    rc = L.bert_hip_test_gemm_lnfold(M, K1, H, N2, p(A1), p(W1), p(b1), p(r), p(rg), p(rb), p(W2), p(b2), p(g), p(be), epi2, p(u), p(out), p(rows))
    if rc != 0:
        raise RuntimeError(f"bert_hip_test_gemm_lnfold failed: {rc}")
    return u, out, rows


def test_attention(qkv: np.ndarray, cu_seqlens: np.ndarray, n_head: int, d_head: int, impl: int) -> np.ndarray:
    L = test_lib()
    qkv = np.ascontiguousarray(qkv, dtype=np.float16)
    cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    T = qkv.shape[0]
    out = np.zeros((T, n_head * d_head), dtype=np.float16)
    r = L.bert_hip_test_attention(len(cu) - 1, _i32p(cu), n_head, d_head, qkv.ctypes.data, impl, out.ctypes.data)
    if r != 0:
        raise RuntimeError(f"bert_hip_test_attention failed: {r}")
    return out


def test_qkv_attention(x: np.ndarray, cu_seqlens: np.ndarray, n_head: int, d_head: int, W_bytes: np.ndarray, wtype: int,
                       bias: np.ndarray, fused: bool) -> np.ndarray:
    """x [T][H] f16, Wqkv [3H][H] in file layout of wtype, bias [3H] -> attention context [T][H] f16."""
    L = test_lib()
    x = np.ascontiguousarray(x, dtype=np.float16)
    cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    w = np.ascontiguousarray(W_bytes)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    out = np.zeros((x.shape[0], n_head * d_head), dtype=np.float16)
    r = L.bert_hip_test_qkv_attention(len(cu) - 1, _i32p(cu), n_head, d_head, x.ctypes.data, w.ctypes.data, wtype,
                                      bias.ctypes.data, int(fused), out.ctypes.data)
    if r != 0:
        raise RuntimeError(f"bert_hip_test_qkv_attention failed: {r}")
    return out


def test_layer_tail(ctx: np.ndarray, x: np.ndarray, Wo_bytes, W1_bytes, W2_bytes, wtype: int, I: int, bo, g1, be1, b1, b2,
                    g2, be2, impl: int) -> np.ndarray:
    """ctx, x [M][H] f16 -> layer output [M][H] f16 (out-projection + LN + FFN + LN); impl see bert_hip.h."""
    L = test_lib()
    ctx = np.ascontiguousarray(ctx, dtype=np.float16)
    x = np.ascontiguousarray(x, dtype=np.float16)
    M, H = ctx.shape
    ws = [np.ascontiguousarray(w) for w in (Wo_bytes, W1_bytes, W2_bytes)]
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    ps = [f(v) for v in (bo, g1, be1, b1, b2, g2, be2)]
    out = np.zeros((M, H), dtype=np.float16)
    r = L.bert_hip_test_layer_tail(M, H, I, ctx.ctypes.data, x.ctypes.data, ws[0].ctypes.data, ws[1].ctypes.data,
                                   ws[2].ctypes.data, wtype, *[p.ctypes.data for p in ps], impl, out.ctypes.data)
    if r != 0:
        raise RuntimeError(f"bert_hip_test_layer_tail failed: {r}")
    return out
