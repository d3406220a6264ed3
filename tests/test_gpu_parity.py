"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel and the whole bert_eval path,
called through the C ABI, against the CPU oracle / numpy on the same seeded inputs.

Tolerances (BASELINE.json north_star): token ids bit-exact (test_host.py); embeddings cosine
>= 1 - 1e-4 for f32/f16 files and >= 0.99 for q4_0/q4_1 files versus the oracle in ggml-faithful
mode.  Tighter bounds are asserted where the numerics allow it."""
import json
import os

import numpy as np
import pytest

from bert_cpp_amd import ggml_file as gf
from bert_cpp_amd import pybert
from oracle import oracle as orc

from conftest import GOLDEN, cosine

pytestmark = pytest.mark.gpu

WT = {"f32": 0, "f16": 1, "q4_0": 2, "q4_1": 3}


# ------------------------------------------------------------------------------------------------
# op level
# ------------------------------------------------------------------------------------------------
def _weight_bytes(w, ftype):
    if ftype == "f32":
        return w.astype(np.float32).view(np.uint8).reshape(-1), w.astype(np.float32)
    if ftype == "f16":
        h = w.astype(np.float16)
        return h.view(np.uint8).reshape(-1), h.astype(np.float32)
    if ftype == "q4_0":
        q = gf.quantize_q4_0(w)
        return q.reshape(-1), gf.dequantize_q4_0(q)
    q = gf.quantize_q4_1(w)
    return q.reshape(-1), gf.dequantize_q4_1(q)


def _gelu(x):
    return 0.5 * x * (1 + np.tanh(0.7978845608028654 * x * (1 + 0.044715 * x * x)))


@pytest.mark.parametrize("impl", [0, 1, 3], ids=["mfma", "naive", "tile256"])
@pytest.mark.parametrize("ftype", ["f16", "q4_0", "q4_1", "f32"])
@pytest.mark.parametrize("shape", [(200, 192, 128), (256, 384, 384), (130, 64, 64), (512, 1536, 384), (384, 384, 1536),
                                   (256, 1152, 384), (129, 2304, 768), (300, 768, 3072), (513, 3072, 768), (256, 256, 64)])
def test_gemm_kernel(impl, ftype, shape):
    M, N, K = shape
    rng = np.random.default_rng(hash((M, N, K, ftype)) % 2 ** 31)
    A = rng.normal(0, 1, (M, K)).astype(np.float16)
    W = (rng.normal(0, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
    # asymmetric structure so a transposed / permuted tile cannot pass
    W[:, : K // 2] *= 1.5
    W[: N // 3] += 0.02
    bias = rng.normal(0, 0.5, N).astype(np.float32)
    resid = rng.normal(0, 1, (M, N)).astype(np.float16)
    wb, wdeq = _weight_bytes(W, ftype)
    base = A.astype(np.float64) @ wdeq.astype(np.float64).T + bias
    for epi in (0, 1, 2):
        want = base if epi == 0 else _gelu(base) if epi == 1 else base + resid.astype(np.float64)
        try:
            got = pybert.test_gemm(A, wb, WT[ftype], N, bias, resid if epi == 2 else None, epi, impl).astype(np.float64)
        except RuntimeError as e:
            if impl == 3 and "-2" in str(e):
                pytest.skip("shape / weight type not handled by this kernel")
            raise
        err = np.abs(got - want)
        tol = 2e-3 * np.abs(want) + 4e-3          # f16 output rounding + f16 weight rounding of the q4 dequant
        bad = err > tol
        assert not bad.any(), (ftype, shape, epi, impl, int(bad.sum()), float(err.max()), np.argwhere(bad)[:5])


@pytest.mark.parametrize("M,N,K", [(20480, 2304, 768), (33000, 768, 3072), (70000, 768, 768), (20000, 3072, 768), (16640, 1536, 384)])
def test_gemm_persistent_workgroups_walk_several_tiles(M, N, K, impl=3):
    """The 256 x 256 tile kernel runs one persistent workgroup per CU: with more output tiles than CUs a workgroup streams
    its reduction tiles across output tiles and finishes a tile's epilogue behind the next tile's first barrier.
    All three epilogues against numpy (float32 BLAS here: the product is too large for a float64 matmul in a test)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.normal(0, 1, (M, K)).astype(np.float16)
    W = (rng.normal(0, 1, (N, K)) / np.sqrt(K)).astype(np.float16)
    W[:, : K // 2] *= 1.5
    W[: N // 3] += 0.02
    bias = rng.normal(0, 0.5, N).astype(np.float32)
    resid = rng.normal(0, 1, (M, N)).astype(np.float16)
    base = A.astype(np.float32) @ W.astype(np.float32).T + bias
    for epi in (0, 1, 2):
        want = base if epi == 0 else _gelu(base.astype(np.float64)).astype(np.float32) if epi == 1 else base + resid.astype(np.float32)
        got = pybert.test_gemm(A, W.view(np.uint8), 1, N, bias, resid if epi == 2 else None, epi, impl).astype(np.float32)
        err = np.abs(got - want)
        bad = err > 2e-3 * np.abs(want) + 4e-3
        assert not bad.any(), (impl, epi, int(bad.sum()), float(err.max()), np.argwhere(bad)[:5].tolist())


@pytest.mark.parametrize("wtype", [2, 3], ids=["q4_0", "q4_1"])
@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (512, 2304, 768), (257, 768, 3072), (20480, 3072, 768), (33000, 768, 3072), (9000, 256, 128)])
def test_gemm256_q4_tile_load_gives_the_f16_form_s_bits(M, N, K, wtype):
    """gemm256 with 4-bit-resident weights (SURVEY §8 row g1 at H = 768: q4_0 / q4_1 blocks dequantised in the GEMM's tile
    load): a thread fetches its 32-weight block two reduction tiles ahead and expands it into the weight tile's LDS image —
    the image the f16 form loads by LDS-DMA from the matrix expanded at load.  Same image, same MFMA sequence: EQUAL BITS with
    the f16 form on the expanded matrix, all three epilogues; sizes with one output tile per workgroup, with several
    (persistent walk, block requests crossing output tiles), and with the minimum of two reduction tiles."""
    rng = np.random.default_rng(M + N + K + wtype)
    A = rng.normal(0, 1, (M, K)).astype(np.float16)
    W = (rng.normal(0, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
    W[:, : K // 2] *= 1.5
    W[: N // 3] += 0.02
    bias = rng.normal(0, 0.5, N).astype(np.float32)
    resid = rng.normal(0, 1, (M, N)).astype(np.float16)
    q = gf.quantize_q4_0(W) if wtype == 2 else gf.quantize_q4_1(W)
    img = _q4_image_f16(q, wtype, (N, K))
    for epi in (0, 1, 2):
        r = resid if epi == 2 else None
        got = pybert.test_gemm(A, q.reshape(-1), wtype, N, bias, r, epi, 3)
        want = pybert.test_gemm(A, img.view(np.uint8).reshape(-1), 1, N, bias, r, epi, 3)
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (epi, int((got != want).sum()), np.argwhere(got != want)[:5].tolist())
    # and the f16 form itself is right (float32 BLAS: the big cases are too large for a float64 product in a test)
    base = A.astype(np.float32) @ img.astype(np.float32).T + bias
    err = np.abs(want.astype(np.float32) - (base + resid.astype(np.float32)))
    assert not (err > 2e-3 * np.abs(base) + 6e-3).any(), float(err.max())


@pytest.mark.parametrize("rebuild", [False, True], ids=["plain-residual", "rebuilt-residual"])
@pytest.mark.parametrize("M,K1,H,N2,epi2", [(300, 768, 768, 3072, 1), (1000, 3072, 768, 2304, 0), (20000, 256, 768, 768, 1), (257, 128, 256, 512, 0)])
def test_layernorm_folded_into_the_gemms(M, K1, H, N2, epi2, rebuild):
    """The H = 768 route's LayerNorms live inside the mat-muls around them (kernels.h GemmLnFold; reference bert.cpp:866-875, :892-901):
    a residual mat-mul writes the UN-normalised sum u and per-row partial statistics of its rounded values — its own residual plain, or
    LayerNorm(r) REBUILT per element from r, r's row statistics and packed (gamma, beta + bias) —; the mat-mul that consumes
    LayerNorm(u) reads u itself: gamma folded into its weights, ONE extra k-step carrying - mean s[n] + std c[n] (hi / lo f16 pairs),
    rows scaled by 1 / std in the epilogue.  Against float64 on the same f16 inputs: u, the row statistics, and
    epi(LayerNorm(u) W2^T + b2) (bias, bias + GELU) — with shifted rows (mean != 0), feature-dependent gamma / beta, several output
    tiles per workgroup."""
    rng = np.random.default_rng(M + K1 + N2 + rebuild)
    A1 = rng.normal(0, 1, (M, K1)).astype(np.float16)
    W1 = (rng.normal(0, 1, (H, K1)) / np.sqrt(K1)).astype(np.float16)
    b1 = rng.normal(0, 0.3, H).astype(np.float32)
    r = (rng.normal(0, 1, (M, H)) + rng.normal(0, 0.7, (M, 1))).astype(np.float16)        # (rows with a mean of their own)
    rg = (1 + rng.normal(0, 0.2, H)).astype(np.float32); rb = rng.normal(0, 0.2, H).astype(np.float32)
    W2 = (rng.normal(0, 1, (N2, H)) / np.sqrt(H)).astype(np.float16)
    b2 = rng.normal(0, 0.5, N2).astype(np.float32)
    g = (1 + rng.normal(0, 0.2, H)).astype(np.float32); be = rng.normal(0, 0.3, H).astype(np.float32)

    def ln(v, gamma, beta):
        v = v.astype(np.float64)
        mu = v.mean(axis=1, keepdims=True)
        var = ((v - mu) ** 2).mean(axis=1, keepdims=True)
        return (v - mu) / np.sqrt(var + 1e-5) * gamma + beta

    u, out, rows = pybert.test_gemm_lnfold(A1, W1, b1, r, rg if rebuild else None, rb if rebuild else None, W2, b2, g, be, epi2)
    resid = ln(r, rg, rb) if rebuild else r.astype(np.float64)
    u_ref = A1.astype(np.float64) @ W1.astype(np.float64).T + b1 + resid
    err = np.abs(u.astype(np.float64) - u_ref)
    assert not (err > 2e-3 * np.abs(u_ref) + 6e-3).any(), ("u", float(err.max()))
    # the statistics are those of the ROUNDED u (what the consumer reads)
    uf = u.astype(np.float64)
    mu, sd = uf.mean(axis=1), np.sqrt(uf.var(axis=1) + 1e-5)
    assert np.abs(rows[:, 3] - sd).max() < 2e-5 * sd.max() + 1e-6 and np.abs(-rows[:, 2] - mu).max() < 1e-5
    assert np.allclose(rows[:, 0] * rows[:, 3], 1, atol=1e-6) and np.allclose(rows[:, 1], rows[:, 2] * rows[:, 0], atol=1e-6)
    base = ln(u, g, be) @ W2.astype(np.float64).T + b2
    want = _gelu(base) if epi2 == 1 else base
    err = np.abs(out.astype(np.float64) - want)
    assert not (err > 3e-3 * np.abs(want) + 8e-3).any(), ("out", float(err.max()), np.argwhere(err > 3e-3 * np.abs(want) + 8e-3)[:5].tolist())
    assert err.mean() < 6e-4, float(err.mean())


@pytest.mark.parametrize("M,H,I", [(200, 128, 256), (256, 256, 512), (130, 384, 1536), (384, 384, 256), (128, 128, 128), (1000, 256, 1024)])
def test_layer_tail_kernel(M, H, I, impl=1):
    """Out-projection + LN + FFN + LN in one launch (layer_tail.hip) and as five kernels (three GEMMs, two LayerNorms: the
    path of the shapes the one-launch kernel does not take) against a float64 reference of reference bert.cpp:859-901."""
    rng = np.random.default_rng(M + H + I)
    ctx = rng.normal(0, 1, (M, H)).astype(np.float16)
    x = rng.normal(0, 1, (M, H)).astype(np.float16)
    Wo = (rng.normal(0, 1, (H, H)) / np.sqrt(H)).astype(np.float16)
    W1 = (rng.normal(0, 1, (I, H)) / np.sqrt(H)).astype(np.float16)
    W2 = (rng.normal(0, 1, (H, I)) / np.sqrt(I)).astype(np.float16)
    bo, b2 = rng.normal(0, 0.2, H), rng.normal(0, 0.2, H)
    b1 = rng.normal(0, 0.5, I)
    g1, g2 = 1 + rng.normal(0, 0.1, H), 1 + rng.normal(0, 0.1, H)
    be1, be2 = rng.normal(0, 0.1, H), rng.normal(0, 0.1, H)

    def ln(v, g, b):
        mu = v.mean(axis=1, keepdims=True)
        var = ((v - mu) ** 2).mean(axis=1, keepdims=True)
        return (v - mu) / np.sqrt(var + 1e-5) * g + b

    f8 = lambda a: a.astype(np.float64)
    y = ln(f8(ctx) @ f8(Wo).T + bo + f8(x), g1, be1)
    y16 = f8(y.astype(np.float16))                       # the device keeps y in f16 (GEMM input and residual)
    u = y16 @ f8(W1).T + b1
    gl = 0.5 * u * (1 + np.tanh(0.7978845608028654 * u * (1 + 0.044715 * u * u)))
    want = ln(f8(gl.astype(np.float16)) @ f8(W2).T + b2 + y16, g2, be2)

    args = (ctx, x, Wo.view(np.uint8), W1.view(np.uint8), W2.view(np.uint8), 1, I, bo, g1, be1, b1, b2, g2, be2)
    base = pybert.test_layer_tail(*args, 0).astype(np.float64)
    err = np.abs(base - want)
    assert err.max() < 2.5e-2 and err.mean() < 2e-3, ("five kernels", M, H, I, float(err.max()), float(err.mean()))
    try:
        got = pybert.test_layer_tail(*args, impl).astype(np.float64)
    except RuntimeError as e:
        if "-2" in str(e):
            assert H not in (256, 384)                   # (the engine falls back to the five kernels)
            return
        raise
    err = np.abs(got - want)
    assert err.max() < 2.5e-2 and err.mean() < 2e-3, (impl, M, H, I, float(err.max()), float(err.mean()))
    assert np.abs(got - base).max() < 2.5e-2


def _q4_image_f16(q, wtype, shape):
    """The f16 image of q4 blocks as the device builds it: (q - 8) d, or q d + m with ONE rounding (an f16 fma)."""
    bs = 18 if wtype == 2 else 20
    flat = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, bs)
    d = flat[:, 0:2].copy().view(np.float16).astype(np.float64)
    qs = flat[:, bs - 16:]
    el = np.concatenate([qs & 0x0F, qs >> 4], axis=1).astype(np.float64)
    if wtype == 2:
        vals = (el - 8.0) * d
    else:
        vals = el * d + flat[:, 2:4].copy().view(np.float16).astype(np.float64)
    return vals.astype(np.float16).reshape(shape)


@pytest.mark.parametrize("wtype", [2, 3], ids=["q4_0", "q4_1"])
@pytest.mark.parametrize("M,H,I", [(130, 384, 1536), (256, 256, 512), (1000, 256, 1024), (384, 384, 256), (3000, 384, 1536)])
def test_layer_tail_kernel_q4(wtype, M, H, I):
    """The layer tail with the weights 4-bit in HBM (BERT_HIP_Q4=fused): raw blocks fetched into registers one interval
    ahead and expanded into the same LDS ring slots the f16 form fills by DMA — so the MFMAs see the f16 image the engine
    builds at load by default, and the output has the bits of the f16 form run on that image."""
    rng = np.random.default_rng(M + H + I + wtype)
    ctx = rng.normal(0, 1, (M, H)).astype(np.float16)
    x = rng.normal(0, 1, (M, H)).astype(np.float16)
    quant = gf.quantize_q4_0 if wtype == 2 else gf.quantize_q4_1
    Ws = [(rng.normal(0, 1, shp) / np.sqrt(shp[1])).astype(np.float32) for shp in ((H, H), (I, H), (H, I))]
    if wtype == 3:
        Ws = [w + 0.03 for w in Ws]                      # a minimum worth storing
    qs = [quant(w) for w in Ws]
    imgs = [_q4_image_f16(q, wtype, w.shape) for q, w in zip(qs, Ws)]
    bo, b2 = rng.normal(0, 0.2, H), rng.normal(0, 0.2, H)
    b1 = rng.normal(0, 0.5, I)
    g1, g2 = 1 + rng.normal(0, 0.1, H), 1 + rng.normal(0, 0.1, H)
    be1, be2 = rng.normal(0, 0.1, H), rng.normal(0, 0.1, H)
    tail = (I, bo, g1, be1, b1, b2, g2, be2)
    got = pybert.test_layer_tail(ctx, x, *[q.view(np.uint8) for q in qs], wtype, *tail, 1)
    want = pybert.test_layer_tail(ctx, x, *[w.view(np.uint8) for w in imgs], 1, *tail, 1)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    base = pybert.test_layer_tail(ctx, x, *[q.view(np.uint8) for q in qs], wtype, *tail, 0)     # five kernels, q4 GEMMs
    assert np.abs(got.astype(np.float64) - base.astype(np.float64)).max() < 2.5e-2


def _attention_ref(qkv, cu, n_head, d):
    T = qkv.shape[0]
    H = n_head * d
    out = np.zeros((T, H))
    q, k, v = qkv[:, :H].astype(np.float64), qkv[:, H:2 * H].astype(np.float64), qkv[:, 2 * H:].astype(np.float64)
    for b in range(len(cu) - 1):
        s, e = cu[b], cu[b + 1]
        for h in range(n_head):
            sl = slice(h * d, (h + 1) * d)
            sc = q[s:e, sl] @ k[s:e, sl].T / np.sqrt(d)
            sc -= sc.max(axis=1, keepdims=True)
            p = np.exp(sc)
            p /= p.sum(axis=1, keepdims=True)
            out[s:e, sl] = p @ v[s:e, sl]
    return out


@pytest.mark.parametrize("impl", [0, 1], ids=["mfma", "naive"])
@pytest.mark.parametrize("d_head,n_head", [(32, 3), (64, 2)])
@pytest.mark.parametrize("lens", [[1, 2, 5, 31, 32, 33, 64, 100, 127, 128], [129, 200, 7, 256], [512, 300]])
def test_attention_kernel(impl, d_head, n_head, lens):
    rng = np.random.default_rng(sum(lens) + d_head)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    T, H = int(cu[-1]), n_head * d_head
    qkv = rng.normal(0, 1, (T, 3 * H)).astype(np.float16)
    qkv[:, :H] *= 1.7          # spread-out scores: softmax far from uniform
    # one dominant key per sentence exercises the running-max path of the long-sequence variant
    for b in range(len(lens)):
        qkv[cu[b + 1] - 1, H:2 * H] *= 4.0
    want = _attention_ref(qkv, cu, n_head, d_head)
    got = pybert.test_attention(qkv, cu, n_head, d_head, impl).astype(np.float64)
    err = np.abs(got - want)
    assert err.max() < 6e-3, (impl, d_head, lens, float(err.max()), np.argwhere(err > 6e-3)[:5])


Q2_CASES = [
    [128, 128, 128],
    [1, 2, 5, 31, 32, 33, 64, 100, 127, 128],
    [96, 97, 48],
    list(range(1, 48)),                       # short sentences: several per 128-slot window
    [16, 16, 16, 16, 16, 16, 16, 16, 15, 17, 1, 1, 1, 1, 1, 1, 1, 1, 1, 112],
    [40, 3, 77, 128, 9, 9, 64, 64, 63, 65, 20],
]


@pytest.mark.parametrize("mode", [2, 3, 4], ids=["next-fit", "uniform", "next-fit-on-device"])
@pytest.mark.parametrize("n_head", [4, 8, 12])
@pytest.mark.parametrize("lens", Q2_CASES, ids=[f"case{i}" for i in range(len(Q2_CASES))])
def test_qkv_attention2_kernel(n_head, lens, mode):
    """Second-generation fused projection + attention (qkv_attention2.hip: windows of 128 token slots holding whole
    sentences, projection waves own token blocks): equal bits with the two-kernel path whatever window a sentence lands
    in, and within f16 rounding of a float64 reference of reference bert.cpp:822-856."""
    d_head, H = 32, 32 * n_head
    rng = np.random.default_rng(sum(lens) + n_head)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    T = int(cu[-1])
    x = rng.normal(0, 1, (T, H)).astype(np.float16)
    W = (rng.normal(0, 1, (3 * H, H)) / np.sqrt(H)).astype(np.float16)
    W[:H] *= 1.7
    W[:, : H // 2] *= 1.3                      # asymmetric in k: a permuted k-tile cannot pass
    bias = rng.normal(0, 0.3, 3 * H).astype(np.float32)
    got = pybert.test_qkv_attention(x, cu, n_head, d_head, W.view(np.uint8), 1, bias, mode)
    split = pybert.test_qkv_attention(x, cu, n_head, d_head, W.view(np.uint8), 1, bias, 0)
    qkv = (x.astype(np.float64) @ W.astype(np.float64).T + bias).astype(np.float16)
    want = _attention_ref(qkv, cu, n_head, d_head)
    err = np.abs(got.astype(np.float64) - want)
    bad = np.argwhere(err > 6e-3)
    assert err.max() < 6e-3, (n_head, mode, float(err.max()), bad[:8].tolist(), len(bad))
    neq = np.argwhere(got.view(np.uint16) != split.view(np.uint16))
    assert len(neq) == 0, (n_head, mode, len(neq), neq[:8].tolist())


@pytest.mark.parametrize("wtype", [2, 3], ids=["q4_0", "q4_1"])
@pytest.mark.parametrize("mode", [2, 3], ids=["next-fit", "uniform"])
@pytest.mark.parametrize("n_head", [4, 8, 12])
@pytest.mark.parametrize("lens", [Q2_CASES[0] * 20, Q2_CASES[1], Q2_CASES[3], Q2_CASES[5]], ids=["full", "edges", "short", "mixed"])
def test_qkv_attention2_kernel_q4(n_head, lens, mode, wtype):
    """The window kernel with the Q|K|V weights 4-bit in HBM (BERT_HIP_Q4=fused): the projection waves fetch raw blocks one
    slab period ahead and expand them into the ring slots the f16 form fills by DMA — the bits of the f16 form run on
    the image the engine builds at load by default."""
    d_head, H = 32, 32 * n_head
    rng = np.random.default_rng(sum(lens) + n_head + wtype)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = rng.normal(0, 1, (int(cu[-1]), H)).astype(np.float16)
    W = (rng.normal(0, 1, (3 * H, H)) / np.sqrt(H)).astype(np.float32)
    W[:H] *= 1.7
    W[:, : H // 2] *= 1.3
    if wtype == 3:
        W += 0.02
    bias = rng.normal(0, 0.3, 3 * H).astype(np.float32)
    q = gf.quantize_q4_0(W) if wtype == 2 else gf.quantize_q4_1(W)
    img = _q4_image_f16(q, wtype, W.shape)
    got = pybert.test_qkv_attention(x, cu, n_head, d_head, q.view(np.uint8), wtype, bias, mode)
    want = pybert.test_qkv_attention(x, cu, n_head, d_head, img.view(np.uint8), 1, bias, mode)
    neq = np.argwhere(got.view(np.uint16) != want.view(np.uint16))
    assert len(neq) == 0, (n_head, mode, wtype, len(neq), neq[:8].tolist())
    split = pybert.test_qkv_attention(x, cu, n_head, d_head, q.view(np.uint8), wtype, bias, 0)     # q4 GEMM + attention kernel
    assert np.abs(got.astype(np.float64) - split.astype(np.float64)).max() < 6e-3


@pytest.mark.parametrize("slot", [16, 8])
@pytest.mark.parametrize("n_sentences", [1, 2, 7, 511, 512, 513, 1024, 5000, 40000])
@pytest.mark.parametrize("dist", ["short", "mixed", "full", "sixteens"])
def test_windows_built_on_the_device_equal_the_host_builder(n_sentences, dist, slot):
    """The device API packs sentences into 128-slot windows with a kernel (a scan over the next-fit automaton's nine
    states — seventeen with 8-slot places —, qkv_attention2.hip build_windows_kernel); the host path with a loop (engine.hip
    build_windows): same list."""
    if slot == 8 and n_sentences in (2, 511, 513):
        pytest.skip("8-slot places: a subset of the sizes")
    pybert.set_window_slots(slot)
    try:
        _windows_equal(n_sentences, dist)
    finally:
        pybert.set_window_slots(16)


def _windows_equal(n_sentences, dist):
    rng = np.random.default_rng(n_sentences * 7 + len(dist))
    lens = {"short": lambda: rng.integers(1, 20, n_sentences),
            "mixed": lambda: np.clip(rng.gamma(2.0, 14.0, n_sentences).astype(np.int64) + 1, 1, 128),
            "full": lambda: rng.integers(120, 129, n_sentences),
            "sixteens": lambda: rng.choice([15, 16, 17, 32, 48, 64, 112, 128], n_sentences)}[dist]()
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    host = pybert.build_windows(cu)
    dev = pybert.build_windows(cu, device=True)
    assert dev == host, (len(dev), len(host), next((i, a, b) for i, (a, b) in enumerate(zip(dev, host)) if a != b) if len(dev) == len(host) else None)
    assert sum(c for _, c in dev) == n_sentences


def test_eight_slot_places_option(make_model):
    """bert_hip_set_option "window_slots" = "8" (BERT_HIP_WINDOW_SLOTS): sentences start at multiples of 8 slots inside the
    attention windows — fewer windows for short sentences.  What holds: the device-built window list equals the host's (test
    above), embeddings agree with the oracle as before, batches whose lengths are all multiples of 16 keep the 16-slot form's
    bits (every place is a multiple of 16 again).  What does NOT hold any more, and why the option is off by default: a
    sentence's last bits depend on where it sits (v_mfma's sum over a 16-key step depends on the k-slot: tools/ubench/mfma_shift.hip)."""
    path, hp = make_model("minilm-l6", "f16", 0)
    m = pybert.BertModel(path)
    rng = np.random.default_rng(21)
    lens = np.clip(np.round(rng.lognormal(np.log(21.0), 0.55, 2000)), 3, 128).astype(np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    toks = rng.integers(1000, hp.n_vocab, size=int(cu[-1])).astype(np.int32)
    lens16 = rng.choice([16, 32, 48, 64, 128], 1500).astype(np.int32)
    cu16 = np.concatenate([[0], np.cumsum(lens16)]).astype(np.int32)
    toks16 = rng.integers(1000, hp.n_vocab, size=int(cu16[-1])).astype(np.int32)
    base, base16 = m.eval_packed(toks, cu), m.eval_packed(toks16, cu16)
    try:
        m.set_option("window_slots", "8")
        m.profile(True)
        got = m.eval_packed(toks, cu)
        assert "qkv_attention2" in m.profile_report()
        m.profile(False)
        got16 = m.eval_packed(toks16, cu16)
    finally:
        m.set_option("window_slots", "16")
    assert np.array_equal(got16, base16)                                  # places at multiples of 16 again: the same bits
    cos = np.einsum("ij,ij->i", got, base)
    assert cos.min() >= 1 - 1e-6 and np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-3
    assert not np.array_equal(got, base)                                  # (the documented price: last bits move with the place)
    o = orc.Oracle(path)
    for i in (0, 7, 999, 1999):
        assert cosine(got[i], o.eval(toks[cu[i]:cu[i + 1]], orc.MODE_GGML)) >= 1 - 1e-4
    assert np.array_equal(m.eval_packed(toks, cu), base)                  # back at 16: the default's bits


def test_qkv_attention2_same_bits_in_any_window():
    """A sentence gives the same bits alone in a window, behind other sentences, or at another 16-slot offset."""
    n_head, d_head, H = 12, 32, 384
    rng = np.random.default_rng(77)
    W = (rng.normal(0, 1, (3 * H, H)) / np.sqrt(H)).astype(np.float16)
    bias = rng.normal(0, 0.3, 3 * H).astype(np.float32)
    sents = [rng.normal(0, 1, (n, H)).astype(np.float16) for n in (23, 40, 7, 16, 33)]

    def run(order):
        xs = np.concatenate([sents[i] for i in order])
        cu = np.concatenate([[0], np.cumsum([len(sents[i]) for i in order])]).astype(np.int32)
        out = pybert.test_qkv_attention(xs, cu, n_head, d_head, W.view(np.uint8), 1, bias, 2)
        return {i: out[cu[k]:cu[k + 1]] for k, i in enumerate(order)}

    alone = {i: run([i])[i] for i in range(len(sents))}
    for order in ([0, 1, 2, 3, 4], [4, 3, 2, 1, 0], [2, 0, 4, 1, 3]):
        got = run(order)
        for i in order:
            assert np.array_equal(got[i].view(np.uint16), alone[i].view(np.uint16)), (order, i)


def test_qkv_attention2_kernel_limits():
    rng = np.random.default_rng(0)
    H = 128
    x = rng.normal(0, 1, (130, H)).astype(np.float16)
    W = rng.normal(0, 0.1, (3 * H, H)).astype(np.float32)
    bias = np.zeros(3 * H, np.float32)
    cu = np.array([0, 130], np.int32)
    with pytest.raises(RuntimeError, match="-2"):                      # longer than a window's 128 slots
        pybert.test_qkv_attention(x, cu, 4, 32, W.astype(np.float16).view(np.uint8), 1, bias, 2)
    H = 192
    q = gf.quantize_q4_0(rng.normal(0, 0.1, (3 * H, H)).astype(np.float32))
    with pytest.raises(RuntimeError, match="-2"):                      # H = 128 / 256 / 384 only
        pybert.test_qkv_attention(rng.normal(0, 1, (64, H)).astype(np.float16), np.array([0, 64], np.int32), 6, 32,
                                  q.view(np.uint8), 2, np.zeros(3 * H, np.float32), 2)


def test_attention_generic_head_dim():
    """d_head outside {32, 64} must route to the generic kernel (mfma entry reports unsupported)."""
    rng = np.random.default_rng(3)
    cu = np.array([0, 9, 40], dtype=np.int32)
    qkv = rng.normal(0, 1, (40, 3 * 4 * 16)).astype(np.float16)
    with pytest.raises(RuntimeError):
        pybert.test_attention(qkv, cu, 4, 16, 0)
    got = pybert.test_attention(qkv, cu, 4, 16, 1).astype(np.float64)
    assert np.abs(got - _attention_ref(qkv, cu, 4, 16)).max() < 6e-3


@pytest.mark.parametrize("ftype", ["f32", "f16", "q4_0", "q4_1"])
@pytest.mark.parametrize("H", [64, 384, 768])
def test_embed_layernorm_kernel(ftype, H):
    """word[id] + type[0] + pos[p] -> LayerNorm(eps 1e-5) * gamma + beta (reference bert.cpp:796-814) against float64 on
    the dequantised tables: every table type the file format has, positions up to 511, ragged sentences."""
    rng = np.random.default_rng(H + len(ftype))
    V, P = 300, 512
    lens = [512, 1, 2, 3, 4, 5, 130, 64, 511]
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    toks = rng.integers(0, V, size=int(cu[-1])).astype(np.int32)
    tabs = [rng.normal(0, 1, (n, H)).astype(np.float32) for n in (V, 2, P)]
    tabs[2] += np.linspace(-2, 2, P)[:, None].astype(np.float32)           # positions are distinguishable
    enc = [_weight_bytes(t, ftype) for t in tabs]
    gamma = rng.normal(1, 0.2, H).astype(np.float32); beta = rng.normal(0, 0.3, H).astype(np.float32)
    got = pybert.test_embed_ln(WT[ftype], enc[0][0], enc[1][0], enc[2][0], H, gamma, beta, toks, cu).astype(np.float64)
    word, typ, pos = (e[1].reshape(-1, H).astype(np.float64) for e in enc)
    p = np.concatenate([np.arange(n) for n in lens])
    pre = word[toks] + typ[0] + pos[p]
    mu = pre.mean(axis=1, keepdims=True)
    want = (pre - mu) / np.sqrt(((pre - mu) ** 2).mean(axis=1, keepdims=True) + 1e-5) * gamma + beta
    err = np.abs(got - want)
    assert err.max() < 4e-3 + 1e-3 * np.abs(want).max(), (ftype, H, float(err.max()), np.argwhere(err > 4e-3)[:4].tolist())


@pytest.mark.parametrize("H", [64, 384, 768, 130])
def test_pool_normalize_kernel(H):
    """mean over ALL tokens of the sentence, then y / ||y||_2 without epsilon (reference bert.cpp:904-913) against float64;
    a sentence outside [1, max_len] gets a NaN row and raises the status word."""
    rng = np.random.default_rng(H)
    lens = [1, 2, 3, 5, 31, 64, 128, 300, 512]
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = rng.normal(0.1, 1, (int(cu[-1]), H)).astype(np.float16)
    got, st = pybert.test_pool_normalize(x, cu, 512)
    assert st == 0
    for b, n in enumerate(lens):
        m = x[cu[b]:cu[b + 1]].astype(np.float64).mean(axis=0)
        want = m / np.sqrt((m * m).sum())
        assert np.abs(got[b] - want).max() < 2e-6, (H, n, float(np.abs(got[b] - want).max()))
    got, st = pybert.test_pool_normalize(x, cu, 128)
    assert st == 1 and np.isnan(got[7]).all() and np.isnan(got[8]).all() and np.isfinite(got[:7]).all()


# ------------------------------------------------------------------------------------------------
# end to end vs the oracle
# ------------------------------------------------------------------------------------------------
MIN_COS = {"f32": 1 - 1e-4, "f16": 1 - 1e-4, "q4_0": 0.99, "q4_1": 0.99}
# On the BASELINE models' dimensions the q4 paths sit far inside the north_star's 0.99: measured 0.99993 against the oracle's
# ggml-faithful mode (the gap is the reference's 8-bit activation blocks, which the GPU path does not have) and 1 - 2e-6 against
# its plain mode (f32 arithmetic on the SAME dequantised weights).  Asserted at what is measured, so that an error growing
# a hundredfold fails.
TIGHT_COS_GGML = {"f32": 1 - 1e-4, "f16": 1 - 1e-4, "q4_0": 0.9995, "q4_1": 0.9995}
TIGHT_COS_PLAIN = 1 - 1e-4
LENS = [1, 2, 3, 17, 31, 32, 33, 48, 64]


@pytest.mark.parametrize("ftype", ["f32", "f16", "q4_0", "q4_1"])
@pytest.mark.parametrize("dims", ["tiny", "tiny-d64", "tiny-d16", "tiny-h128"])
def test_eval_matches_oracle_small(make_model, dims, ftype):
    path, hp = make_model(dims, ftype, 1)
    m = pybert.BertModel(path)
    o = orc.Oracle(path)
    rng = np.random.default_rng(5)
    sents = [rng.integers(0, hp.n_vocab, size=min(n, hp.n_max_tokens)).astype(np.int32) for n in LENS]
    got = m.eval_batch(sents)
    coss = []
    for s, g in zip(sents, got):
        want = o.eval(s, orc.MODE_GGML)
        plain = o.eval(s, orc.MODE_PLAIN)
        assert abs(np.linalg.norm(g) - 1) < 1e-3
        c = cosine(g, want)
        coss.append(c)
        assert c >= MIN_COS[ftype], (dims, ftype, len(s), c)
        # the dequantise-to-f16 GPU path must be at least as close to exact arithmetic on the stored
        # weights as the reference's own 8-bit-activation CPU path is
        if ftype in ("q4_0", "q4_1"):
            assert cosine(g, plain) >= min(cosine(want, plain), 1 - 1e-4) - 1e-4
        else:
            assert cosine(g, plain) >= 1 - 1e-4
    assert np.mean(coss) >= MIN_COS[ftype]


@pytest.mark.parametrize("ftype", ["f16", "q4_0"])
def test_hidden_states_match_oracle(make_model, ftype):
    """Layer-by-layer tap (bert.cpp:806-901) localises any divergence."""
    path, hp = make_model("tiny", ftype, 2)
    m = pybert.BertModel(path)
    o = orc.Oracle(path)
    s = np.random.default_rng(9).integers(0, hp.n_vocab, size=40).astype(np.int32)
    emb, hid = m.eval_hidden(s)
    want_emb, want_hid = o.eval(s, orc.MODE_PLAIN, want_hidden=True)
    tol = 6e-3 if ftype == "f16" else 0.25
    for layer in range(hp.n_layer + 1):
        err = np.abs(hid[layer] - want_hid[layer]).max()
        assert err < tol * (1 + layer), (ftype, layer, err)
    assert cosine(emb, want_emb) > (1 - 1e-4 if ftype == "f16" else 0.99)


def test_eval_matches_huggingface_golden():
    """Independent float implementation (tests/golden/make_golden.py) through the HIP path."""
    with open(os.path.join(GOLDEN, "hf_tiny_golden.json")) as f:
        g = json.load(f)
    m = pybert.BertModel(os.path.join(GOLDEN, g["model"]))
    got = m.eval_batch(g["sentences"])
    for e, want in zip(got, g["embeddings"]):
        assert cosine(e, want) >= 1 - 1e-4


@pytest.mark.parametrize("dims,ftype", [("minilm-l6", "f16"), ("minilm-l6", "q4_0"), ("bert-base-l2", "f16"), ("bert-base-l2", "q4_1")])
def test_baseline_dims_match_huggingface_live(tmp_path, dims, ftype):
    """The fused kernels of the BASELINE models against an INDEPENDENT float implementation built on the box: a HuggingFace
    `BertModel` (random init, fixed seed; tanh GELU and LayerNorm eps 1e-5, the numerics of the reference's ggml ops — SURVEY
    Appendix C) at all-MiniLM-L6-v2's dimensions (one launch for all layers / two per layer) and at bert-base's (two layers of
    it: gemm256 + attention at 512 tokens), exported the way the reference's convert-to-ggml.py does, with its matrices replaced
    by what the file's weight type stores (f16 rounding, q4 block dequantisation) so that only the ARITHMETIC differs: f16
    activations with f32 accumulation against f32 throughout.  Cosine of the mean-pooled, normalised embeddings >= 1 - 1e-4."""
    # (torch runs in a process of its own: `import torch` brings the wheel's own ROCm libraries — a second librccl among them —
    # into a process whose later tests use the system's RCCL through libbert.so)
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_reference.py")
    r = subprocess.run([sys.executable, script, dims, ftype, str(tmp_path)], capture_output=True, text=True, timeout=900)
    if r.returncode == 77:
        pytest.skip("torch / transformers not importable")
    assert r.returncode == 0, r.stderr[-2000:]
    path = str(tmp_path / f"hf_{dims}_{ftype}.bin")
    ref = np.load(str(tmp_path / "hf_reference.npz"))
    sents = [ref[f"ids{i}"] for i in range(int(ref["n"]))]
    want = [ref[f"want{i}"] for i in range(int(ref["n"]))]
    m = pybert.BertModel(path)
    m.profile(True)
    got = m.eval_batch(sents)
    names = set(m.profile_report())
    m.profile(False)
    assert ({"qkv_attention2", "layer_tail"} <= names or "model_kernel" in names) if dims == "minilm-l6" else {"gemm_qkv", "attention", "gemm_ffn_up"} <= names, names
    coss = [cosine(a, b) for a, b in zip(got, want)]
    assert min(coss) >= 1 - 1e-4, coss


@pytest.mark.parametrize("dims,ftype,n,lens", [
    ("minilm-l6", "f16", 6, [128, 128, 5, 77, 128, 1]),
    ("minilm-l6", "q4_0", 4, [128, 128, 64, 9]),
    ("minilm-l6", "q4_1", 2, [128, 33]),
    ("minilm-l6", "f32", 2, [128, 20]),
    ("bert-base", "q4_1", 2, [512, 130]),
    ("mpnet-dims", "q4_0", 2, [128, 514]),
])
def test_eval_matches_oracle_baseline_models(make_model, dims, ftype, n, lens):
    """BASELINE.json configs' model dimensions, sentence counts the oracle finishes in seconds."""
    path, hp = make_model(dims, ftype, 0)
    m = pybert.BertModel(path)
    o = orc.Oracle(path)
    ids = [gf.synthetic_token_ids(1, L, hp.n_vocab, seed=100 + i)[0] if L >= 2 else np.array([101], np.int32)
           for i, L in enumerate(lens)]
    got = m.eval_batch(ids)
    coss = [cosine(g, o.eval(s, orc.MODE_GGML)) for s, g in zip(ids, got)]
    assert min(coss) >= (TIGHT_COS_GGML[ftype] if min(lens) >= 8 else MIN_COS[ftype]), (dims, ftype, coss)
    plain = [cosine(g, o.eval(s, orc.MODE_PLAIN)) for s, g in zip(ids, got)]
    assert min(plain) >= TIGHT_COS_PLAIN, (dims, ftype, plain)


@pytest.mark.parametrize("dims,lens", [
    ("tiny", LENS), ("tiny-d64", LENS), ("tiny-d16", LENS), ("tiny-h128", LENS),
    ("minilm-l6", [128, 20, 1, 77]), ("bert-base", [300, 17]),
])
def test_f32_files_run_in_f32_arithmetic(make_model, dims, lens):
    """f32 model files (ftype 0) take ggml's f32 mat-mul in the reference (bert.cpp:825 with GGML_TYPE_F32 tensors, typed at
    :407-429): the engine's f32 route (f32_route.hip: f32 activations, v_mfma_f32_32x32x2_f32, f32 softmax / GELU / LayerNorm)
    must reproduce the oracle's plain f32 arithmetic to max-abs 2e-5 per embedding component — an f16-operand pass is 50x
    off that.  The route is what runs by default (kernel family asserted); "f32" = "f16" selects f16 operands and the fused
    kernels, which stay within the cosine tolerance of every other file type."""
    path, hp = make_model(dims, "f32", 1)
    m = pybert.BertModel(path)
    o = orc.Oracle(path)
    rng = np.random.default_rng(11)
    sents = [rng.integers(0, hp.n_vocab, size=min(n, hp.n_max_tokens)).astype(np.int32) for n in lens]
    m.profile(True)
    got = m.eval_batch(sents)
    rep = m.profile_report(families=True)
    m.profile(False)
    assert rep.get("family:gemm_f32", {}).get("launches") == 4 * hp.n_layer and not any(k.startswith("family:gemm") and k != "family:gemm_f32" for k in rep), rep
    worst = 0.0
    for s, g in zip(sents, got):
        plain = o.eval(s, orc.MODE_PLAIN)
        worst = max(worst, float(np.abs(g - plain).max()))
        assert abs(np.linalg.norm(g) - 1) < 1e-5
        assert cosine(g, o.eval(s, orc.MODE_GGML)) >= 1 - 1e-4          # (ggml mode: fp16 exp / GELU tables on f32 mat-muls)
    assert worst <= 2e-5, (dims, worst)
    # per-sentence results do not depend on the batch or the entry point
    assert np.array_equal(np.stack([m.eval(s) for s in sents]), got)
    # the f16-operand route on the same file
    m.set_option("f32", "f16")
    m.profile(True)
    fast = m.eval_batch(sents)
    rep = m.profile_report(families=True)
    m.profile(False)
    assert "family:gemm_f32" not in rep, rep
    assert min(cosine(a, b) for a, b in zip(fast, got)) >= 1 - 1e-4
    m.set_option("f32", "exact")
    assert np.array_equal(m.eval_batch(sents), got)


def test_f32_route_hidden_states(make_model):
    """Layer-by-layer tap of the f32 route against the oracle's plain mode (bert.cpp:806-901)."""
    path, hp = make_model("tiny", "f32", 2)
    m = pybert.BertModel(path)
    o = orc.Oracle(path)
    s = np.random.default_rng(9).integers(0, hp.n_vocab, size=40).astype(np.int32)
    emb, hid = m.eval_hidden(s)
    want_emb, want_hid = o.eval(s, orc.MODE_PLAIN, want_hidden=True)
    for layer in range(hp.n_layer + 1):
        assert np.abs(hid[layer] - want_hid[layer]).max() < 2e-4 * (1 + layer), layer
    assert np.abs(emb - want_emb).max() <= 2e-5


# ------------------------------------------------------------------------------------------------
# API semantics
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims,ftype", [("tiny", "f16"), ("tiny-h128", "f16"), ("tiny-h128", "q4_0"), ("tiny-d64", "q4_1")])
def test_api_equivalences_and_batch_independence(make_model, dims, ftype):
    path, hp = make_model(dims, ftype, 1)
    m = pybert.BertModel(path)
    rng = np.random.default_rng(0)
    sents = [rng.integers(0, hp.n_vocab, size=n).astype(np.int32) for n in (7, 64, 1, 33, 20, 64, 50, 3, 64, 11)]
    batch = m.eval_batch(sents)
    for _ in range(3):                      # the same launch sequence gives the same bits every time
        assert np.array_equal(m.eval_batch(sents), batch)
    single = np.stack([m.eval(s) for s in sents])
    cu = np.concatenate([[0], np.cumsum([len(s) for s in sents])]).astype(np.int32)
    packed = m.eval_packed(np.concatenate(sents), cu)
    # per-sentence results do not depend on what else is in the batch, nor on the entry point: bit-exact
    assert np.array_equal(batch, single)
    assert np.array_equal(batch, packed)
    rev = m.eval_batch(sents[::-1])[::-1]
    assert np.array_equal(batch, rev)
    # small device chunks give the same bits
    m.set_option("chunk_tokens", "40")
    assert np.array_equal(m.eval_batch(sents), batch)


def test_eval_batch_rows_in_place_or_scattered(make_model, capfd):
    """bert_eval_batch's result pointers (bert.h: `float **batch_embeddings`): rows of one matrix are written in place by the
    engine, scattered rows go through a matrix of the library's own — the same embeddings; and a sentence that cannot be
    evaluated leaves its row and the rows behind it untouched either way (reference bert.cpp:765-769)."""
    path, hp = make_model("tiny-h128", "f16", 1)
    m = pybert.BertModel(path)
    rng = np.random.default_rng(9)
    sents = [rng.integers(0, hp.n_vocab, size=int(n)).astype(np.int32) for n in rng.integers(1, hp.n_max_tokens + 1, size=70)]
    a = m.eval_batch(sents)
    b = m.eval_batch(sents, scattered=True)
    assert np.isfinite(a).all() and np.array_equal(a, b)
    bad = sents[:5] + [np.zeros(hp.n_max_tokens + 1, dtype=np.int32)] + sents[5:9]
    for scattered in (False, True):
        out = m.eval_batch(bad, scattered=scattered)
        assert np.array_equal(out[:5], a[:5]) and np.isnan(out[5:]).all()
    capfd.readouterr()


def test_host_path_pipelines_chunks(make_model):
    """eval_packed_host keeps two chunks in flight (stage / compute / unpack overlap, engine.hip): many chunks of
    uneven size, buffers growing between calls, a single-chunk call in between — always the bits of the unchunked call."""
    path, hp = make_model("tiny-h128", "f16", 1)
    m = pybert.BertModel(path)
    rng = np.random.default_rng(42)
    sents = [rng.integers(0, hp.n_vocab, size=int(n)).astype(np.int32) for n in rng.integers(1, hp.n_max_tokens + 1, size=300)]
    want = m.eval_batch(sents)
    for chunk in (64, 500, 129, 4096, 70):
        m.set_option("chunk_tokens", str(chunk))
        assert np.array_equal(m.eval_batch(sents[:17]), want[:17]), chunk          # small call first: buffers grow afterwards
        assert np.array_equal(m.eval_batch(sents), want), chunk
    m2 = pybert.BertModel(path)                                                     # cold context, chunked from the first call
    m2.set_option("chunk_tokens", "200")
    assert np.array_equal(m2.eval_batch(sents), want)


def test_api_error_behaviour(make_model, capfd):
    path, hp = make_model("tiny", "f16", 1)
    m = pybert.BertModel(path)
    ok = np.arange(5, dtype=np.int32)
    too_long = np.zeros(hp.n_max_tokens + 1, dtype=np.int32)
    out = m.eval_batch([ok, too_long, ok])
    err = capfd.readouterr().err
    assert f"Too many tokens, maximum is {hp.n_max_tokens}" in err       # reference bert.cpp:767
    assert np.isfinite(out[0]).all()                                      # evaluated before the failure
    assert np.isnan(out[1]).all() and np.isnan(out[2]).all()              # untouched, as in the reference
    out = m.eval_batch([np.array([hp.n_vocab], dtype=np.int32)])
    assert np.isnan(out).all() and "out of range" in capfd.readouterr().err
    # maximum length works
    full = np.random.default_rng(1).integers(0, hp.n_vocab, size=hp.n_max_tokens).astype(np.int32)
    assert abs(np.linalg.norm(m.eval(full)) - 1) < 1e-3


def test_encode_equals_tokenize_plus_eval(make_model, tmp_path):
    hp = gf.MODEL_DIMS["tiny"]
    words = ["[PAD]", "[UNK]"] + [f"w{i}" for i in range(2, 101)] + ["[CLS]", "[SEP]"] + list("abcdefghij") + \
            ["hello", "world", "##ing", "##s", "test", ",", ".", "!"]
    vocab = [w.encode() for w in words] + [f"[unused{i}]".encode() for i in range(len(words), hp.n_vocab)]
    path = str(tmp_path / "vocab_model.bin")
    gf.write_model(path, hp, gf.synthetic_weights(hp, 4), gf.FTYPE_F16, vocab=vocab)
    m = pybert.BertModel(path)
    texts = ["hello world!", "testing tests, a b c.", "", "HELLO hello Hello"]
    enc = m.encode_batch(texts)
    for t, e in zip(texts, enc):
        ids = m.tokenize(t)
        assert ids[0] == 101 and ids[-1] == 102
        assert np.array_equal(e, m.eval(ids))
        assert np.array_equal(e, m.encode(t))
    # more inputs than one group of bert_encode_batch (groups of 2048, 4096, 8192, then 16384 texts: group g+1 is tokenized while
    # group g is on the GPU; the groups' id buffers belong to the context and are reused by later calls)
    rng = np.random.default_rng(8)
    pool = ["hello", "world", "testing", "tests", "a", "b", "c", ",", ".", "!", "HELLO", "xyzzy"]
    many = [" ".join(rng.choice(pool, size=int(rng.integers(0, 12)))) for _ in range(15001)]
    enc = m.encode_batch(many, n_threads=5)
    ids = m.tokenize_batch(many, n_threads=3)
    want = m.eval_batch(ids)
    assert np.array_equal(enc, want)
    for i in (0, 2047, 2048, 6143, 6144, 14335, 14336, 15000):
        assert np.array_equal(enc[i], m.encode(many[i])), i
    # a later, smaller call on the same context: nothing of the earlier groups' ids shows through
    few = many[7000:7000 + 2500][::-1]
    assert np.array_equal(m.encode_batch(few, n_threads=2), want[7000:7000 + 2500][::-1])


@pytest.mark.parametrize("dims,ftype", [("tiny-h128", "q4_0"), ("tiny-d64", "q4_1"), ("minilm-l6", "q4_0"), ("tiny", "q4_1")])
def test_q4_expanded_at_load_equals_fused_dequant(make_model, dims, ftype, monkeypatch):
    """q4 weight matrices are expanded to f16 images once at load by default (BERT_HIP_Q4=expand) and then run the
    f16 kernels; BERT_HIP_Q4=fused keeps the 4-bit planes and dequantises in the kernels' tile loads.  Same weight
    values either way: the embeddings agree to f16-accumulation-order noise and both match the oracle."""
    path, hp = make_model(dims, ftype, 5)
    rng = np.random.default_rng(3)
    lens = [5, 64, 17, 33, 64] if hp.n_max_tokens < 128 else [128, 90, 7, 128, 64]
    sents = [rng.integers(0, hp.n_vocab, size=min(n, hp.n_max_tokens)).astype(np.int32) for n in lens]
    a = pybert.BertModel(path).eval_batch(sents)
    monkeypatch.setenv("BERT_HIP_Q4", "fused")
    b = pybert.BertModel(path).eval_batch(sents)
    monkeypatch.delenv("BERT_HIP_Q4")
    ref = orc.Oracle(path)
    if dims == "minilm-l6":
        # the window kernel and the layer tail expand the blocks into the tile image the default path loads: the same bits
        assert np.array_equal(a, b), float(np.abs(a - b).max())
    for i, s in enumerate(sents):
        assert cosine(a[i], b[i]) > 1 - 2e-5, (i, cosine(a[i], b[i]))
        want = ref.eval(s)
        assert cosine(a[i], want) >= TIGHT_COS_GGML[ftype] and cosine(b[i], want) >= TIGHT_COS_GGML[ftype]


@pytest.mark.parametrize("ftype", ["q4_0", "q4_1"])
def test_legacy_q4_files_load_and_give_the_same_embeddings(tmp_path, ftype):
    hp = gf.MODEL_DIMS["tiny-h128"]
    w = gf.synthetic_weights(hp, 3)
    cur, leg = str(tmp_path / "cur.bin"), str(tmp_path / "leg.bin")
    gf.write_model(cur, hp, w, gf.FTYPE_BY_NAME[ftype])
    gf.write_model(leg, hp, w, gf.FTYPE_BY_NAME[ftype], legacy_q4=True)
    sents = [np.random.default_rng(1).integers(0, hp.n_vocab, size=n).astype(np.int32) for n in (5, 64, 33)]
    assert np.array_equal(pybert.BertModel(cur).eval_batch(sents), pybert.BertModel(leg).eval_batch(sents))


@pytest.mark.parametrize("dims,ftype", [("minilm-l6", "f16"), ("minilm-l6", "q4_0"), ("h256-l3", "f16")])
def test_latency_route_gives_the_batch_route_s_bits(make_model, dims, ftype):
    """Batches of at most 128 tokens take the latency route (skinny.hip: every mat-mul of a layer split by output features
    over many workgroups instead of one workgroup per 128 tokens).  A sentence's embedding must not depend on what it is
    batched with, so the route has to reproduce the fused kernels' arithmetic bit for bit: a sentence alone, a few short
    sentences together, and the same sentences inside a large batch give identical bits."""
    gf.MODEL_DIMS.setdefault("h256-l3", gf.BertHParams(1000, 128, 256, 1024, 8, 3))      # (H = 256: the NT = 2 forms of the kernels)
    path, hp = make_model(dims, ftype, 0)
    m = pybert.BertModel(path)
    rng = np.random.default_rng(11)
    lens = [128, 25, 1, 77, 33, 96, 64, 5, 127, 32, 31]
    sents = [rng.integers(0, hp.n_vocab, size=n).astype(np.int32) for n in lens]
    m.profile(True)
    batch = m.eval_batch(sents)                              # 619 tokens: the fused batch route
    names_batch = set(m.profile_report())
    alone = [m.eval_batch([s])[0] for s in sents]
    names_alone = set(m.profile_report())
    few = m.eval_batch([sents[1], sents[2], sents[7], sents[4], sents[10]])      # 95 tokens, five sentences
    m.profile(False)
    assert {"qkv_attention2", "layer_tail"} <= names_batch and not any(k.startswith("skinny") for k in names_batch), names_batch
    assert {"skinny_qkv", "skinny_proj", "skinny_ffn_up", "skinny_ffn_down", "skinny_layernorm", "attention"} <= names_alone, names_alone
    for i, a in enumerate(alone):
        assert np.array_equal(a, batch[i]), (ftype, lens[i], float(np.abs(a - batch[i]).max()))
    for k, i in enumerate([1, 2, 7, 4, 10]):
        assert np.array_equal(few[k], batch[i]), (ftype, "few", lens[i])
    want = orc.Oracle(path).eval(sents[3])
    assert cosine(alone[3], want) >= TIGHT_COS_GGML[ftype]
    # the shipped cap (768 tokens: small batches of a polling server): the 619-token batch itself on the route, and batches
    # around the cap on either side
    m.set_option("latency_tokens", "768")
    m.profile(True)
    routed = m.eval_batch(sents)
    names = set(m.profile_report())
    assert {"skinny_qkv", "skinny_ffn_down", "attention"} <= names and "layer_tail" not in names and "model_kernel" not in names, names
    assert np.array_equal(routed, batch)
    more = sents + sents[:3]                                  # 619 + 154 = 773 tokens: one token past the cap -> the batch route
    out = m.eval_batch(more)
    names = set(m.profile_report())
    m.profile(False)
    assert not any(k.startswith("skinny") for k in names), names
    assert np.array_equal(out[:len(sents)], batch) and np.array_equal(out[len(sents):], batch[:3])
    six = [sents[0]] * 6                                      # 768 tokens of full windows: still the route (not the one-launch kernel)
    assert np.array_equal(m.eval_batch(six), np.stack([batch[0]] * 6))


@pytest.mark.parametrize("dims,ftype,q4", [("minilm-l6", "f16", None), ("minilm-l6", "q4_0", None), ("minilm-l6", "q4_1", "fused"),
                                            ("h256-l3", "f16", None), ("h256-l3", "q4_0", "fused"), ("tiny-d64", "f16", None)])
def test_shipped_latency_cap_through_every_entry_point(make_model, dims, ftype, q4, tmp_path, monkeypatch):
    """The suite runs with BERT_HIP_LATENCY=128 (tests/conftest.py: the parity tests are meant for the batch kernels); the SHIPPED
    default sends calls of up to 768 tokens down the latency route.  Here the default is what is loaded: calls of 129 .. 768
    tokens through bert_eval_batch (row pointers in place and scattered), bert_hip_eval_packed and bert_encode_batch — f16, q4 expanded at load, q4 planes in HBM, d_head 32 (H = 384 / 256) and a d_head 64 model
    the route declines — give the bits of the batch route (the same files loaded with the one-window cap), and match the oracle."""
    gf.MODEL_DIMS.setdefault("h256-l3", gf.BertHParams(1000, 128, 256, 1024, 8, 3))
    path, hp = make_model(dims, ftype, 3)
    if q4:
        monkeypatch.setenv("BERT_HIP_Q4", q4)
    m_batch = pybert.BertModel(path)                         # conftest's cap: one window
    monkeypatch.delenv("BERT_HIP_LATENCY")
    m = pybert.BertModel(path)                               # the shipped default
    rng = np.random.default_rng(21)
    cap = min(hp.n_max_tokens, 128)
    for lens in ([cap, 25, 1, 77 % cap + 1, 33, cap - 2, 64, 5][: 8], [cap] * 5, [7] * 40, [cap, cap, 3]):
        sents = [rng.integers(0, hp.n_vocab, size=n).astype(np.int32) for n in lens]
        total = sum(lens)
        assert 128 < total <= 768, total
        want = m_batch.eval_batch(sents)
        m.profile(True)
        got = m.eval_batch(sents)
        names = set(m.profile_report())
        m.profile(False)
        # (the route under test is the one that ran; 4-bit planes in HBM and H < 256 are shapes the route declines: the batch kernels)
        assert any(k.startswith("skinny") for k in names) == (dims != "tiny-d64" and q4 != "fused"), (dims, q4, names)
        assert np.array_equal(got, want), (dims, ftype, lens, float(np.abs(got - want).max()))
        assert np.array_equal(m.eval_batch(sents, scattered=True), want)
        ids = np.concatenate(sents)
        cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        assert np.array_equal(m.eval_packed(ids, cu), want)
    o = orc.Oracle(path)
    assert cosine(got[0], o.eval(sents[0])) >= TIGHT_COS_GGML[ftype]
    # strings in: a vocabulary for the same weights
    if dims == "minilm-l6" and ftype == "f16":
        words = ["[PAD]"] * 100 + ["[UNK]", "[CLS]", "[SEP]"] + ["hello", "world", "##ing", "##s", "test", ",", ".", "!", "a", "b", "c"]
        vocab = [w.encode() for w in words] + [f"[unused{i}]".encode() for i in range(len(words), hp.n_vocab)]
        vpath = str(tmp_path / "vocab_model.bin")
        gf.write_model(vpath, hp, gf.synthetic_weights(hp, 3), gf.FTYPE_F16, vocab=vocab)
        mv = pybert.BertModel(vpath)
        texts = ["hello world! " * 9, "testing tests, a b c.", "a", "hello " * 30] * 3         # ~500 tokens: the route
        enc = mv.encode_batch(texts)
        idl = [mv.tokenize(t) for t in texts]
        assert 128 < sum(len(i) for i in idl) <= 768
        monkeypatch.setenv("BERT_HIP_LATENCY", "128")
        assert np.array_equal(enc, pybert.BertModel(vpath).eval_batch(idl))


@pytest.mark.parametrize("ftype", ["f16", "q4_0"])
@pytest.mark.parametrize("knob", ["tail=0", "qkv2=0", "kernels=tiled", "kernels=naive"])
def test_kernel_families_agree_end_to_end(make_model, knob, ftype, monkeypatch):
    """The two fused kernels each have a fallback: the one-launch layer tail -> tiled GEMMs + LayerNorm kernels, the window
    kernel (projection + attention) -> QKV GEMM + attention kernel; BERT_HIP_KERNELS=tiled takes both, =naive the generic
    kernels.  On the benchmark's dimensions every family must give the same embeddings up to accumulation-order noise, and
    match the oracle; q4 files with the blocks kept 4-bit on the device (every family dequantises in its tile load)."""
    path, hp = make_model("minilm-l6", ftype, 2)
    rng = np.random.default_rng(4)
    sents = [rng.integers(0, hp.n_vocab, size=n).astype(np.int32) for n in (128, 128, 96, 128, 77, 128)]
    if ftype != "f16":
        monkeypatch.setenv("BERT_HIP_Q4", "fused")
    base = pybert.BertModel(path).eval_batch(sents)
    key, value = knob.split("=")
    if key == "kernels":
        monkeypatch.setenv("BERT_HIP_KERNELS", value)
        m_alt = pybert.BertModel(path, test_routes=value == "naive")     # (the generic kernels as a whole-model route: libbert_test.so only)
        monkeypatch.delenv("BERT_HIP_KERNELS")
    else:
        m_alt = pybert.BertModel(path)
        m_alt.set_option(key, value)
    alt = m_alt.eval_batch(sents)
    m_alt.profile(True); m_alt.eval_batch(sents); names = set(m_alt.profile_report()); m_alt.profile(False)
    assert ("layer_tail" in names) == (knob == "qkv2=0"), (knob, names)
    assert ("qkv_attention2" in names) == (knob == "tail=0"), (knob, names)
    want = orc.Oracle(path).eval(sents[2])
    assert cosine(base[2], want) >= TIGHT_COS_GGML[ftype] and cosine(alt[2], want) >= TIGHT_COS_GGML[ftype]
    for i in range(len(sents)):
        assert cosine(base[i], alt[i]) > 1 - 1e-6, (knob, i, cosine(base[i], alt[i]))


@pytest.mark.parametrize("ftype", ["f16", "q4_0"])
@pytest.mark.parametrize("B", [1, 3, 130, 512])
def test_one_launch_gives_the_two_kernel_route_s_bits(make_model, ftype, B):
    """Batches of full windows (every sentence exactly 128 tokens) run all layers in ONE launch (model_kernel.hip: a workgroup
    carries its window through every layer, the window kernel and the layer tail as phases).  Same arithmetic per sentence
    as two launches per layer — equal bits — and a sentence gives the same bits in a full-window batch and a ragged one."""
    path, hp = make_model("minilm-l6", ftype, 0)
    m = pybert.BertModel(path)
    m.set_option("latency", "0")                             # (B = 1: the latency route would take it)
    ids = gf.synthetic_token_ids(B, 128, hp.n_vocab, seed=77 + B)
    cu = (np.arange(B + 1) * 128).astype(np.int32)
    m.profile(True)
    got = m.eval_packed(ids.reshape(-1), cu)
    names = set(m.profile_report())
    m.profile(False)
    assert names == {"embed_ln", "model_kernel"}, names     # (the workgroups pool their sentences themselves)
    m.set_option("one_launch", "0")
    m.profile(True)
    want = m.eval_packed(ids.reshape(-1), cu)
    names = set(m.profile_report())
    m.profile(False)
    assert {"qkv_attention2", "layer_tail"} <= names and "model_kernel" not in names, names
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    m.set_option("one_launch", "1")
    ragged = m.eval_batch([ids[0], ids[0][:77], ids[B - 1]])               # not all full: two launches per layer
    assert np.array_equal(ragged[0], got[0]) and np.array_equal(ragged[2], got[B - 1])
    want0 = orc.Oracle(path).eval(ids[0])
    assert cosine(got[0], want0) >= TIGHT_COS_GGML[ftype]


def test_workspace_growth_does_not_race_with_the_forward_pass(make_model):
    """Growing batches make the engine reallocate (and zero-fill) its output / workspace buffers between
    evaluations; the fill must be complete before kernels of the next pass write them (it once was not: the
    null-stream memset overtook the pooling kernel now and then)."""
    path, hp = make_model("tiny-h128", "q4_0", 11)
    rng = np.random.default_rng(0)
    sents = [rng.integers(0, hp.n_vocab, size=n).astype(np.int32) for n in (1, 2, 3, 4, 5, 7, 9, 12, 17, 25, 33, 48, 63, 64)]
    m0 = pybert.BertModel(path)
    ref = [m0.eval(s).copy() for s in sents]
    for trial in range(12):
        m = pybert.BertModel(path)                       # fresh context: every growth step happens again
        for k in (1, 2, 3, 5, 8, 14):
            idx = rng.choice(len(sents), size=k, replace=True)
            out = m.eval_batch([sents[i] for i in idx])
            for j, i in enumerate(idx):
                assert np.array_equal(out[j], ref[i]), (trial, k, int(i))
        m.close()


@pytest.mark.parametrize("ftype", ["f16", "q4_1"])
def test_bert_large_dims_both_layernorm_routes(make_model, ftype, monkeypatch):
    """H = 1024 (bert-large's width: sixteen heads of 64, I = 4096, two layers here): wider than anything BASELINE.json names — eight
    statistics partials per row in the folded LayerNorms, four feature tiles in the residual mat-muls.  Mixed lengths up to 300
    tokens (the long-sentence attention form), enough tokens for the 256 x 256-tile mat-muls: the folded route and the plain one
    agree with each other to rounding noise and with the oracle in both of its modes."""
    gf.MODEL_DIMS.setdefault("h1024-l2", gf.BertHParams(2000, 512, 1024, 4096, 16, 2))
    path, hp = make_model("h1024-l2", ftype, 0)
    rng = np.random.default_rng(11)
    lens = np.concatenate([rng.integers(3, 301, size=30), [300, 1, 128, 129]])
    sents = [rng.integers(0, hp.n_vocab, size=int(n)).astype(np.int32) for n in lens]
    outs = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("BERT_HIP_LN_FOLD", fold)
        m = pybert.BertModel(path)
        m.profile(True)
        outs[fold] = m.eval_batch(sents)
        rep = m.profile_report(families=True)
        m.profile(False)
        assert ("ln_rows_finalize" in rep) == (fold == "1"), sorted(rep)
        assert rep["layernorm"]["launches"] == (1 if fold == "1" else 2 * hp.n_layer), rep["layernorm"]
        assert not any(k.startswith("family:gemm_mfma") or k.startswith("family:gemm_naive") for k in rep), sorted(rep)
        m.close()
    assert np.isfinite(outs["1"]).all() and np.abs(np.linalg.norm(outs["1"], axis=1) - 1).max() < 1e-3
    assert min(cosine(a, b) for a, b in zip(outs["1"], outs["0"])) > 1 - 2e-6
    o = orc.Oracle(path)
    for i in (0, 7, 30, 31, 33):
        assert cosine(outs["1"][i], o.eval(sents[i], orc.MODE_GGML)) >= TIGHT_COS_GGML[ftype], i
        assert cosine(outs["0"][i], o.eval(sents[i], orc.MODE_GGML)) >= TIGHT_COS_GGML[ftype], i
    assert cosine(outs["1"][33], o.eval(sents[33], orc.MODE_PLAIN)) >= (TIGHT_COS_PLAIN if ftype == "f16" else 0.999)


# ------------------------------------------------------------------------------------------------
# BASELINE sizes through size-independent properties
# ------------------------------------------------------------------------------------------------
_FULL_SIZE_OUT = {}       # (config, q4 mode) -> embeddings: the fused run is compared with the default's bits


@pytest.mark.parametrize("q4", ["expand", "expand-plain-layernorm", "fused"])
def test_full_size_batch_properties_bert_base(make_model, q4, monkeypatch):
    """configs[3] at full size: bert-base dims q4_1, 512 sentences of 512 tokens.  Unit norm, duplicates give identical
    bits wherever they sit, the kernels the H = 768 path is meant to use are the ones that ran, and eight sentences spread
    over the batch agree with the oracle (ggml-faithful mode).  Both ways: the q4 matrices expanded to f16 at load (the
    default) and as BASELINE.json writes the config — 4-bit in HBM, dequantised in gemm256's tile load (kernel family
    asserted: every weight mat-mul of the pass ran on gemm256 with q4 planes) — with EQUAL BITS between the two."""
    path, hp = make_model("bert-base", "q4_1", 0)
    # (the default folds the H = 768 LayerNorms into the mat-muls around them — kernels.h GemmLnFold; "expand-plain-layernorm": the
    # LayerNorm kernels of their own, the arithmetic the 4-bit-planes form shares bit for bit)
    plain_ln = q4 == "expand-plain-layernorm"
    q4 = q4.split("-")[0]
    monkeypatch.setenv("BERT_HIP_Q4", q4)
    if plain_ln:
        monkeypatch.setenv("BERT_HIP_LN_FOLD", "0")
    m = pybert.BertModel(path)
    folded = q4 == "expand" and not plain_ln
    B, N = 512, 512
    ids = gf.synthetic_token_ids(B, N, hp.n_vocab, seed=1234 + 3)
    ids[B // 2] = ids[3]
    ids[B - 1] = ids[3]
    cu = (np.arange(B + 1) * N).astype(np.int32)
    m.profile(True)
    out = m.eval_packed(ids.reshape(-1), cu)
    rep = m.profile_report(families=True)
    m.profile(False)
    assert {"gemm_qkv", "gemm_ffn_up", "gemm_ffn_down", "gemm_attn_out", "attention", "layernorm"} <= set(rep), sorted(rep)
    fam = {k: v["launches"] for k, v in rep.items() if k.startswith("family:")}
    # (plain LayerNorms: the Q | K | V mat-mul runs on the 4-bit planes too — its f16 image does not fit an XCD's L2 — the other three on
    # f16 images; folded: only the first layer's, whose input the embedding kernel normalised, the later ones on the folded f16 image)
    want_fam = {"family:gemm256_q4": 4 * hp.n_layer} if q4 == "fused" else \
               {"family:gemm256_q4": 1, "family:gemm256_f16": 4 * hp.n_layer - 1} if folded else \
               {"family:gemm256_q4": hp.n_layer, "family:gemm256_f16": 3 * hp.n_layer}
    assert fam == want_fam, fam
    # (folded: ONE LayerNorm launch per pass — the last layer's, for the pooling — and a row-statistics launch per LayerNorm)
    assert rep["layernorm"]["launches"] == (1 if folded else 2 * hp.n_layer), rep["layernorm"]
    assert ("ln_rows_finalize" in rep) == folded and (not folded or rep["ln_rows_finalize"]["launches"] == 2 * hp.n_layer)
    assert np.isfinite(out).all()
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-3
    assert np.array_equal(out[3], out[B // 2]) and np.array_equal(out[3], out[B - 1])
    _FULL_SIZE_OUT[("bert-base", q4, plain_ln)] = out
    if q4 == "fused" and ("bert-base", "expand", True) in _FULL_SIZE_OUT:
        assert np.array_equal(out, _FULL_SIZE_OUT[("bert-base", "expand", True)])
    if folded and ("bert-base", "expand", True) in _FULL_SIZE_OUT:        # (folding changes roundings, not values)
        assert min(cosine(a, b) for a, b in zip(out[::17], _FULL_SIZE_OUT[("bert-base", "expand", True)][::17])) > 1 - 2e-6
    o = orc.Oracle(path)
    sample = [0, 3, 77, B // 3, B // 2 + 5, 400, B - 2, B - 1]
    coss = [cosine(out[i], o.eval(ids[i], orc.MODE_GGML)) for i in sample]
    assert min(coss) >= TIGHT_COS_GGML["q4_1"], coss
    assert min(cosine(out[i], o.eval(ids[i], orc.MODE_PLAIN)) for i in sample[:2]) >= TIGHT_COS_PLAIN


@pytest.mark.parametrize("ftype,B,q4", [("f16", 256, None), ("q4_0", 1024, "expand"), ("q4_0", 1024, "fused")])
def test_full_size_batch_properties(make_model, ftype, B, q4, monkeypatch):
    """configs[1] / configs[2]: MiniLM-L6 dims, seq_len 128.  Unit norm, duplicate sentences give
    identical bits wherever they sit, and a sample agrees with the oracle.  configs[2] both ways: the q4 matrices expanded
    to f16 at load (the default), and as written — 4-bit in HBM, dequantised in the tile loads of the same two kernels."""
    path, hp = make_model("minilm-l6", ftype, 0)
    if q4:
        monkeypatch.setenv("BERT_HIP_Q4", q4)
    m = pybert.BertModel(path)
    ids = gf.synthetic_token_ids(B, 128, hp.n_vocab, seed=1234 + (1 if ftype == "f16" else 2))
    ids[B // 2] = ids[3]
    ids[B - 1] = ids[3]
    cu = (np.arange(B + 1) * 128).astype(np.int32)
    m.profile(True)
    out = m.eval_packed(ids.reshape(-1), cu)
    rep = m.profile_report()
    m.profile(False)
    # full windows (every sentence 128 tokens) and f16 images: ONE launch for all layers, a workgroup per window (the two fused
    # kernels' bodies as phases); 4-bit-resident weights: two launches per layer (the window kernel and the layer tail)
    if q4 == "fused":
        assert set(rep) == {"embed_ln", "qkv_attention2", "layer_tail", "pool_normalize"}, sorted(rep)
        assert rep["qkv_attention2"]["launches"] == hp.n_layer and rep["layer_tail"]["launches"] == hp.n_layer
    else:
        assert set(rep) == {"embed_ln", "model_kernel"}, sorted(rep)
        assert rep["model_kernel"]["launches"] == 1
        m.set_option("one_launch", "0")
        assert np.array_equal(m.eval_packed(ids.reshape(-1), cu), out)          # the same bits as two launches per layer
    assert np.isfinite(out).all()
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-3
    assert np.array_equal(out[3], out[B // 2]) and np.array_equal(out[3], out[B - 1])
    o = orc.Oracle(path)
    sample = [0, 3, B // 3, B - 2]
    coss = [cosine(out[i], o.eval(ids[i], orc.MODE_GGML)) for i in sample]
    assert min(coss) >= TIGHT_COS_GGML[ftype], coss
    assert min(cosine(out[i], o.eval(ids[i], orc.MODE_PLAIN)) for i in sample) >= TIGHT_COS_PLAIN


@pytest.mark.parametrize("q4", ["expand", "expand-plain-layernorm", "fused"])
def test_full_size_batch_properties_mpnet_dims(make_model, q4, monkeypatch):
    """configs[4]'s model (BERT architecture at mpnet-base dimensions, q4_0), 1024 sentences of 128 tokens.  Unit norm,
    duplicates give identical bits, the H = 768 kernel family ran (with BERT_HIP_Q4=fused: on 4-bit planes, equal bits with
    the default), and six sentences spread over the batch agree with both oracle modes."""
    path, hp = make_model("mpnet-dims", "q4_0", 0)
    plain_ln = q4 == "expand-plain-layernorm"
    q4 = q4.split("-")[0]
    monkeypatch.setenv("BERT_HIP_Q4", q4)
    if plain_ln:
        monkeypatch.setenv("BERT_HIP_LN_FOLD", "0")
    m = pybert.BertModel(path)
    folded = q4 == "expand" and not plain_ln
    B, N = 1024, 128
    ids = gf.synthetic_token_ids(B, N, hp.n_vocab, seed=1234 + 4)
    ids[B // 2] = ids[3]
    ids[B - 1] = ids[3]
    cu = (np.arange(B + 1) * N).astype(np.int32)
    m.profile(True)
    out = m.eval_packed(ids.reshape(-1), cu)
    rep = m.profile_report(families=True)
    m.profile(False)
    assert {"gemm_qkv", "gemm_ffn_up", "gemm_ffn_down", "gemm_attn_out", "attention"} <= set(rep), sorted(rep)
    fam = {k: v["launches"] for k, v in rep.items() if k.startswith("family:")}
    want_fam = {"family:gemm256_q4": 4 * hp.n_layer} if q4 == "fused" else \
               {"family:gemm256_q4": 1, "family:gemm256_f16": 4 * hp.n_layer - 1} if folded else \
               {"family:gemm256_q4": hp.n_layer, "family:gemm256_f16": 3 * hp.n_layer}
    assert fam == want_fam, fam
    assert rep["layernorm"]["launches"] == (1 if folded else 2 * hp.n_layer), rep["layernorm"]
    assert np.isfinite(out).all()
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-3
    assert np.array_equal(out[3], out[B // 2]) and np.array_equal(out[3], out[B - 1])
    _FULL_SIZE_OUT[("mpnet-dims", q4, plain_ln)] = out
    if q4 == "fused" and ("mpnet-dims", "expand", True) in _FULL_SIZE_OUT:
        assert np.array_equal(out, _FULL_SIZE_OUT[("mpnet-dims", "expand", True)])
    if folded and ("mpnet-dims", "expand", True) in _FULL_SIZE_OUT:
        assert min(cosine(a, b) for a, b in zip(out[::17], _FULL_SIZE_OUT[("mpnet-dims", "expand", True)][::17])) > 1 - 2e-6
    o = orc.Oracle(path)
    sample = [0, 3, B // 3, B // 2 + 5, B - 2, B - 1]
    coss = [cosine(out[i], o.eval(ids[i], orc.MODE_GGML)) for i in sample]
    assert min(coss) >= TIGHT_COS_GGML["q4_0"], coss
    assert min(cosine(out[i], o.eval(ids[i], orc.MODE_PLAIN)) for i in sample[:3]) >= TIGHT_COS_PLAIN


def test_config5_share_one_call_of_125000_sentences(make_model):
    """BASELINE configs[4] as SURVEY §8(d) writes it ("Config 5"): 1,000,000 sentences of 128 tokens over 8 GPUs = 125,000 per
    GPU, through ONE host-to-host call (bert_hip_eval_packed: what bert_eval_batch runs) — 61 chunks of 2048 sentences through
    the two-slot staging pipeline of engine.hip, 64 MB of ids in, 384 MB of embeddings out.  The pipeline gives the bits of
    single-chunk calls (chunks from the start, the middle, the end and the ragged last one), duplicates placed in far-apart
    chunks come back identical, every row has unit norm, and a fixed sample agrees with the oracle.  The same call through
    bert_hip_eval_packed_gather (the path's exchange step, a 1-rank RCCL communicator here) returns the same matrix."""
    from test_multi_device import _Hip
    path, hp = make_model("mpnet-dims", "q4_0", 0)
    m = pybert.BertModel(path)
    B, N, H = 125000, 128, hp.n_embd
    ids = gf.synthetic_token_ids(B, N, hp.n_vocab, seed=1234 + 4)
    for dup in (70000, B - 1):
        ids[dup] = ids[3]
    cu = (np.arange(B + 1, dtype=np.int64) * N).astype(np.int32)
    out = m.eval_packed(ids.reshape(-1), cu)
    assert np.isfinite(out).all()
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-3
    assert np.array_equal(out[3], out[70000]) and np.array_equal(out[3], out[B - 1])
    per_chunk = 262144 // N
    n_chunks = (B + per_chunk - 1) // per_chunk
    assert n_chunks == 62 and B - 61 * per_chunk == 72          # 61 full chunks + a last one of 72 sentences
    for c in (0, 1, 30, 60, 61):
        b0, b1 = c * per_chunk, min(B, (c + 1) * per_chunk)
        single = m.eval_packed(ids[b0:b1].reshape(-1), cu[: b1 - b0 + 1])
        assert np.array_equal(single, out[b0:b1]), c
    o = orc.Oracle(path)
    sample = np.random.default_rng(5).choice(B, size=6, replace=False)
    coss = [cosine(out[i], o.eval(ids[i], orc.MODE_GGML)) for i in sample]
    assert min(coss) >= TIGHT_COS_GGML["q4_0"], coss
    hip = _Hip()
    m.set_option("test_rccl_single", "1")
    ptrs = m.eval_packed_gather(ids.reshape(-1), cu)
    assert np.array_equal(hip.download(ptrs[0], (B, H)), out)
