#!/usr/bin/env python3
"""What the vendor GEMM library (hipBLASLt behind torch.matmul) sustains on this board for the matrix shapes of the forward
pass — a calibration line for the hand-written kernels, not part of the product (nothing in libbert.so calls a library GEMM).
Random normal f16 operands (the power draw of the MFMA pipes depends on the operand bits; zeros run much faster)."""
import sys, torch

dev = torch.device("cuda:0")
shapes = [  # (M tokens, N out features, K in features, label)
    (32768, 1152, 384, "MiniLM QKV, 256x128 tokens"),
    (32768, 384, 384, "MiniLM out-proj"),
    (32768, 1536, 384, "MiniLM FFN up"),
    (32768, 384, 1536, "MiniLM FFN down"),
    (131072, 1536, 384, "MiniLM FFN up, 1024x128 tokens"),
    (262144, 2304, 768, "bert-base QKV, 512x512 tokens"),
    (262144, 3072, 768, "bert-base FFN up"),
    (262144, 768, 3072, "bert-base FFN down"),
    (8192, 8192, 8192, "square 8192"),
]
for zeros in (False, True):
    for M, N, K, label in shapes:
        if zeros and M != 8192:
            continue
        a = torch.zeros(M, K, device=dev, dtype=torch.float16) if zeros else torch.randn(M, K, device=dev, dtype=torch.float16)
        w = torch.zeros(N, K, device=dev, dtype=torch.float16) if zeros else torch.randn(N, K, device=dev, dtype=torch.float16) / K ** 0.5
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        for _ in range(5):
            torch.matmul(a, w.t(), out=out)
        iters = 50 if M * N * K < 4e12 else 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            torch.matmul(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{label:34s} M={M:6d} N={N:5d} K={K:5d} {'zeros ' if zeros else 'random'} {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms * 1e-9:7.0f} TFLOP/s", flush=True)
