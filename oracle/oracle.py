"""ctypes binding of the CPU oracle (oracle/bert_oracle.cpp).

TEST INFRASTRUCTURE ONLY — may be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libbert_oracle.so")
MODE_GGML, MODE_PLAIN = 0, 1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bert_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def usable_cores() -> int:
    """CPU cores this process may actually use: min(affinity, cgroup quota).  The GPU box reports
    256 logical CPUs; spinning up one OpenMP thread per logical CPU for 128-row loops is slower
    than a handful of threads, so callers cap this further (default_threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def default_threads() -> int:
    """Thread count the oracle uses unless told otherwise (intra-op, like ggml's n_threads)."""
    env = os.environ.get("ORACLE_THREADS")
    if env:
        return max(1, int(env))
    return min(usable_cores(), 16)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # before libgomp initialises
        os.environ.setdefault("OMP_PROC_BIND", "false")
        L = C.CDLL(_LIB_PATH)
        L.oracle_load.restype = C.c_void_p
        L.oracle_load.argtypes = [C.c_char_p, C.c_int]
        L.oracle_free.argtypes = [C.c_void_p]
        for fn in ("oracle_n_embd", "oracle_n_max_tokens", "oracle_n_layer", "oracle_ftype"):
            getattr(L, fn).restype = C.c_int32
            getattr(L, fn).argtypes = [C.c_void_p]
        L.oracle_vocab_id_to_token.restype = C.c_char_p
        L.oracle_vocab_id_to_token.argtypes = [C.c_void_p, C.c_int32]
        L.oracle_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32]
        L.oracle_eval.restype = C.c_int
        L.oracle_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int32,
                                  C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_f2h_soft.restype = C.c_uint16
        L.oracle_f2h_soft.argtypes = [C.c_float]
        L.oracle_f2h.restype = C.c_uint16
        L.oracle_f2h.argtypes = [C.c_float]
        L.oracle_h2f_soft.restype = C.c_float
        L.oracle_h2f_soft.argtypes = [C.c_uint16]
        L.oracle_num_threads.restype = C.c_int
        _lib = L
    return _lib


class Oracle:
    """CPU oracle context for one model file."""

    def __init__(self, path: str, vocab_only: bool = False):
        self._L = lib()
        self._h = self._L.oracle_load(path.encode(), int(vocab_only))
        if not self._h:
            raise RuntimeError(f"oracle_load failed for {path}")
        self.n_embd = self._L.oracle_n_embd(self._h)
        self.n_max_tokens = self._L.oracle_n_max_tokens(self._h)
        self.n_layer = self._L.oracle_n_layer(self._h)
        self.ftype = self._L.oracle_ftype(self._h)

    def close(self):
        if self._h:
            self._L.oracle_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tokenize(self, text: str | bytes, n_max_tokens: Optional[int] = None) -> List[int]:
        n_max = n_max_tokens or self.n_max_tokens
        buf = (C.c_int32 * max(n_max, 2))()
        n = C.c_int32(0)
        data = text if isinstance(text, bytes) else text.encode("utf-8")
        self._L.oracle_tokenize(self._h, data, buf, C.byref(n), n_max)
        return list(buf[: n.value])

    def id_to_token(self, i: int) -> bytes:
        return self._L.oracle_vocab_id_to_token(self._h, i)

    def eval(self, tokens: Sequence[int], mode: int = MODE_GGML, n_threads: int = 0,
             want_hidden: bool = False):
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.n_embd, dtype=np.float32)
        hid = None
        hp = None
        if want_hidden:
            hid = np.empty((self.n_layer + 1, len(toks), self.n_embd), dtype=np.float32)
            hp = hid.ctypes.data_as(C.POINTER(C.c_float))
        if n_threads <= 0:
            n_threads = default_threads()
        r = self._L.oracle_eval(self._h, mode, n_threads, toks.ctypes.data_as(C.POINTER(C.c_int32)), len(toks),
                                out.ctypes.data_as(C.POINTER(C.c_float)), hp)
        if r != 0:
            raise RuntimeError(f"oracle_eval failed: {r}")
        return (out, hid) if want_hidden else out

    def eval_batch(self, sentences: Sequence[Sequence[int]], mode: int = MODE_GGML, n_threads: int = 0) -> np.ndarray:
        return np.stack([self.eval(s, mode, n_threads) for s in sentences])
