#!/usr/bin/env python3
"""Kernel-tuning helper: time the fused FFN kernel alone for several (M, H, I)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert_cpp_amd import pybert
L = pybert.lib()
L.bert_hip_bench_ffn.restype = C.c_float
L.bert_hip_bench_ffn.argtypes = [C.c_int32] * 4
for (M, H, I) in [(32768, 384, 128), (32768, 384, 768), (32768, 384, 1536), (32768, 384, 3072), (131072, 384, 1536)]:
    ms = L.bert_hip_bench_ffn(M, H, I, 20)
    print(f"M={M} H={H} I={I}: {ms*1e3:8.1f} us  {4.0*M*H*I/ms/1e9:8.1f} TFLOP/s")
