#!/usr/bin/env python3
"""gemm256 with q4 planes against the f16 form on the expanded matrix: EQUAL BITS over a sweep of shapes chosen for the corners of
the block-request pipeline — 2 .. 6 reduction tiles per output tile (requests for the tile after next cross output-tile
boundaries at once), one to many output tiles per persistent workgroup, both block types, all three epilogues.
usage: stress_gemm256_q4.py [n_random_shapes]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bert_cpp_amd import ggml_file as gf, pybert  # noqa: E402
from test_gpu_parity import _q4_image_f16  # noqa: E402

fixed = [(140000, 256, 128), (70000, 512, 192), (66000, 256, 256), (40000, 768, 320), (33000, 1024, 384), (300, 256, 128), (257, 2304, 768)]
rng = np.random.default_rng(2024)
n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shapes = fixed + [(int(rng.integers(1, 90000)), 256 * int(rng.integers(1, 7)), 64 * int(rng.integers(2, 13))) for _ in range(n_rand)]
bad = 0
for M, N, K in shapes:
    A = rng.normal(0, 1, (M, K)).astype(np.float16)
    W = (rng.normal(0, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.normal(0, 0.5, N).astype(np.float32)
    resid = rng.normal(0, 1, (M, N)).astype(np.float16)
    for wtype in (2, 3):
        q = gf.quantize_q4_0(W) if wtype == 2 else gf.quantize_q4_1(W)
        img = _q4_image_f16(q, wtype, (N, K))
        for epi in (0, 1, 2):
            r = resid if epi == 2 else None
            got = pybert.test_gemm(A, q.reshape(-1), wtype, N, bias, r, epi, 3)
            want = pybert.test_gemm(A, img.view(np.uint8).reshape(-1), 1, N, bias, r, epi, 3)
            n = int((got.view(np.uint16) != want.view(np.uint16)).sum())
            if n:
                bad += 1
                print(f"MISMATCH M={M} N={N} K={K} wtype={wtype} epi={epi}: {n} elements, first {np.argwhere(got.view(np.uint16) != want.view(np.uint16))[:3].tolist()}", flush=True)
    print(f"M={M:6d} N={N:5d} K={K:5d} ok" if not bad else f"M={M} N={N} K={K} checked", flush=True)
print("stress_gemm256_q4:", len(shapes), "shapes,", bad, "mismatching cases")
sys.exit(1 if bad else 0)
