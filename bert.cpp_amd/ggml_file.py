"""On-disk model format of bert.cpp (`ggml-model-{f32,f16,q4_0,q4_1}.bin`): writer, reader and
the q4_0 / q4_1 block quantizers, in numpy.

This is the data format on the input side of the hot path.  It
follows, without copying code:
  * header / vocab / tensor records ......... reference models/convert-to-ggml.py:68-108,
                                              read back by bert.cpp:342-669
  * which tensors get quantized ............. reference models/quantize.cpp:154-167
                                              (2-D tensors whose name ends in "weight")
  * q4_0 / q4_1 block layout and rounding ... upstream ggml `quantize_row_q4_0_reference` /
                                              `quantize_row_q4_1_reference` of the mid-2023 API era
                                              the reference targets (ggml is NOT vendored in the
                                              reference tree)

Layouts (little endian, no padding):
  block_q4_0 = { f16 d; u8 qs[16] }        18 B / 32 weights, w = (q - 8) * d
  block_q4_1 = { f16 d; f16 m; u8 qs[16] } 20 B / 32 weights, w = q * d + m
  qs[j] low nibble = element j, high nibble = element j + 16.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MAGIC = 0x67676D6C
FTYPE_F32, FTYPE_F16, FTYPE_Q4_0, FTYPE_Q4_1 = 0, 1, 2, 3
FTYPE_NAMES = {0: "f32", 1: "f16", 2: "q4_0", 3: "q4_1"}
FTYPE_BY_NAME = {v: k for k, v in FTYPE_NAMES.items()}
QK = 32


# --------------------------------------------------------------------------------------------
# block quantizers
# --------------------------------------------------------------------------------------------
def quantize_q4_0(w: np.ndarray) -> np.ndarray:
    """f32 [..., K] (K % 32 == 0) -> uint8 [..., K/32, 18] in block_q4_0 layout."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    assert w.shape[-1] % QK == 0
    blk = w.reshape(-1, QK)
    idx = np.argmax(np.abs(blk), axis=1)             # first element of largest magnitude
    mx = blk[np.arange(blk.shape[0]), idx]           # signed
    d = (mx / np.float32(-8.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    x = blk * inv[:, None]
    q = np.minimum(15, (x + np.float32(8.5)).astype(np.int8)).astype(np.uint8)   # C cast truncates
    out = np.empty((blk.shape[0], 18), dtype=np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
    return out.reshape(*w.shape[:-1], w.shape[-1] // QK, 18)


def quantize_q4_1(w: np.ndarray) -> np.ndarray:
    """f32 [..., K] -> uint8 [..., K/32, 20] in block_q4_1 layout."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    assert w.shape[-1] % QK == 0
    blk = w.reshape(-1, QK)
    mn = blk.min(axis=1)
    mx = blk.max(axis=1)
    d = ((mx - mn) / np.float32(15.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    x = (blk - mn[:, None]) * inv[:, None]
    q = np.minimum(15, (x + np.float32(0.5)).astype(np.int8)).astype(np.uint8)
    out = np.empty((blk.shape[0], 20), dtype=np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:4] = mn.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 4:] = q[:, :16] | (q[:, 16:] << 4)
    return out.reshape(*w.shape[:-1], w.shape[-1] // QK, 20)


def dequantize_q4_0(b: np.ndarray) -> np.ndarray:
    """uint8 [..., nb, 18] -> f32 [..., nb*32] (exact f32 product, as ggml's dequantize_row)."""
    b = np.ascontiguousarray(b, dtype=np.uint8)
    flat = b.reshape(-1, 18)
    d = flat[:, 0:2].copy().view(np.float16).astype(np.float32)       # [nblk,1]
    qs = flat[:, 2:]
    lo = (qs & 0x0F).astype(np.int32) - 8
    hi = (qs >> 4).astype(np.int32) - 8
    vals = np.concatenate([lo, hi], axis=1).astype(np.float32) * d
    return vals.reshape(*b.shape[:-2], b.shape[-2] * QK)


def dequantize_q4_1(b: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint8)
    flat = b.reshape(-1, 20)
    d = flat[:, 0:2].copy().view(np.float16).astype(np.float32)
    m = flat[:, 2:4].copy().view(np.float16).astype(np.float32)
    qs = flat[:, 4:]
    lo = (qs & 0x0F).astype(np.float32)
    hi = (qs >> 4).astype(np.float32)
    vals = np.concatenate([lo, hi], axis=1) * d + m
    return vals.reshape(*b.shape[:-2], b.shape[-2] * QK)


# Legacy block layouts (ggml before May 2023; the model sizes the reference's README prints, README.md:105-107, are
# these): block_q4_0 = { f32 d; u8 qs[16] } = 20 B, block_q4_1 = { f32 d; f32 m; u8 qs[16] } = 24 B, and byte qs[j] holds
# elements 2j (low nibble) and 2j+1 (high nibble) instead of j and j+16.
def to_legacy_q4(b: np.ndarray, ftype: int) -> np.ndarray:
    """uint8 [..., nb, 18|20] (current layout) -> uint8 [..., nb, 20|24] (legacy layout), same q's, d / m widened to f32."""
    bs = 18 if ftype == FTYPE_Q4_0 else 20
    flat = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, bs)
    nsc = 1 if ftype == FTYPE_Q4_0 else 2
    sc = flat[:, : 2 * nsc].copy().view(np.float16).astype(np.float32)         # [nblk, nsc]
    qs = flat[:, 2 * nsc:]
    el = np.concatenate([qs & 0x0F, qs >> 4], axis=1)                          # elements 0..31
    out = np.empty((flat.shape[0], 4 * nsc + 16), dtype=np.uint8)
    out[:, : 4 * nsc] = sc.view(np.uint8).reshape(-1, 4 * nsc)
    out[:, 4 * nsc:] = el[:, 0::2] | (el[:, 1::2] << 4)
    return out.reshape(*b.shape[:-1], 4 * nsc + 16)


# --------------------------------------------------------------------------------------------
# model description
# --------------------------------------------------------------------------------------------
@dataclass
class BertHParams:
    n_vocab: int = 30522
    n_max_tokens: int = 512
    n_embd: int = 384
    n_intermediate: int = 1536
    n_head: int = 12
    n_layer: int = 6

    @property
    def d_head(self) -> int:
        return self.n_embd // self.n_head


# Named dimension sets used by BASELINE.json's configs.
MODEL_DIMS: Dict[str, BertHParams] = {
    "minilm-l6": BertHParams(30522, 512, 384, 1536, 12, 6),       # all-MiniLM-L6-v2
    "minilm-l12": BertHParams(30522, 512, 384, 1536, 12, 12),     # all-MiniLM-L12-v2
    "bert-base": BertHParams(30522, 512, 768, 3072, 12, 12),      # bert-base-uncased
    "mpnet-dims": BertHParams(30527, 514, 768, 3072, 12, 12),     # BERT-arch at mpnet-base dims
    "tiny": BertHParams(256, 64, 64, 128, 2, 2),                  # unit-test size (d_head 32)
    "tiny-d64": BertHParams(300, 96, 128, 256, 2, 2),             # d_head = 64 variant
    "tiny-h128": BertHParams(256, 64, 128, 256, 4, 2),            # H % 128 == 0: fused-FFN kernel path
    "tiny-d16": BertHParams(256, 64, 64, 192, 4, 1),              # d_head = 16 (generic-kernel path)
}


def tensor_names(n_layer: int) -> List[str]:
    """The 5 + 16*L tensor names bert.cpp maps (reference bert.cpp:503-553), HF state-dict order."""
    names = [
        "embeddings.word_embeddings.weight",
        "embeddings.position_embeddings.weight",
        "embeddings.token_type_embeddings.weight",
        "embeddings.LayerNorm.weight",
        "embeddings.LayerNorm.bias",
    ]
    for i in range(n_layer):
        p = f"encoder.layer.{i}."
        names += [
            p + "attention.self.query.weight", p + "attention.self.query.bias",
            p + "attention.self.key.weight", p + "attention.self.key.bias",
            p + "attention.self.value.weight", p + "attention.self.value.bias",
            p + "attention.output.dense.weight", p + "attention.output.dense.bias",
            p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias",
            p + "intermediate.dense.weight", p + "intermediate.dense.bias",
            p + "output.dense.weight", p + "output.dense.bias",
            p + "output.LayerNorm.weight", p + "output.LayerNorm.bias",
        ]
    return names


def tensor_shape(name: str, hp: BertHParams) -> Tuple[int, ...]:
    """numpy (row-major, [out, in]) shape of each tensor."""
    H, I = hp.n_embd, hp.n_intermediate
    if name == "embeddings.word_embeddings.weight":
        return (hp.n_vocab, H)
    if name == "embeddings.position_embeddings.weight":
        return (hp.n_max_tokens, H)
    if name == "embeddings.token_type_embeddings.weight":
        return (2, H)
    if name.endswith("intermediate.dense.weight"):
        return (I, H)
    if name.endswith("intermediate.dense.bias"):
        return (I,)
    if name.endswith(".output.dense.weight") and "attention" not in name:
        return (H, I)
    if name.endswith(".weight") and "LayerNorm" not in name:
        return (H, H)
    return (H,)


def synthetic_weights(hp: BertHParams, seed: int = 0, style: str = "sensitive") -> Dict[str, np.ndarray]:
    """Seeded random f32 weights of the BERT architecture.

    style="sensitive": fan-in scaled matrices with O(1) activations, spread-out attention scores,
    non-trivial biases and LayerNorm affine terms, so that a wrong softmax / bias / LN path moves
    the output (used by the parity tests and the bench).
    style="hf": HuggingFace BERT default init, N(0, 0.02) matrices, gamma=1, beta=0, bias=0.
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name in tensor_names(hp.n_layer):
        shp = tensor_shape(name, hp)
        if style == "hf":
            if len(shp) == 2:
                a = rng.normal(0.0, 0.02, shp)
            elif name.endswith("LayerNorm.weight"):
                a = np.ones(shp)
            else:
                a = np.zeros(shp)
        else:
            if name.startswith("embeddings.") and len(shp) == 2:
                a = rng.normal(0.0, 1.0, shp)
            elif len(shp) == 2:
                gain = 2.0 if (".query." in name or ".key." in name) else 1.0
                a = rng.normal(0.0, gain / np.sqrt(shp[1]), shp)
            elif name.endswith("LayerNorm.weight"):
                a = rng.normal(1.0, 0.1, shp)
            else:
                a = rng.normal(0.0, 0.1, shp)
        out[name] = a.astype(np.float32)
    return out


def synthetic_vocab(n_vocab: int) -> List[bytes]:
    """A vocab whose strings do not matter for eval tests; ids 101/102 keep their BERT meaning."""
    v = [f"[unused{i}]".encode() for i in range(n_vocab)]
    if n_vocab > 103:
        v[0], v[100], v[101], v[102], v[103] = b"[PAD]", b"[UNK]", b"[CLS]", b"[SEP]", b"[MASK]"
    return v


# --------------------------------------------------------------------------------------------
# writer / reader
# --------------------------------------------------------------------------------------------
def _encode_2d(a: np.ndarray, ftype: int) -> bytes:
    if ftype == FTYPE_F32:
        return a.astype(np.float32).tobytes()
    if ftype == FTYPE_F16:
        return a.astype(np.float16).tobytes()
    if ftype == FTYPE_Q4_0:
        return quantize_q4_0(a).tobytes()
    if ftype == FTYPE_Q4_1:
        return quantize_q4_1(a).tobytes()
    raise ValueError(ftype)


def write_model(path: str, hp: BertHParams, weights: Dict[str, np.ndarray], ftype: int,
                vocab: Optional[Sequence[bytes]] = None, from_f16: bool = True, legacy_q4: bool = False) -> None:
    """Write a bert.cpp model file.  2-D tensors named '*weight' take `ftype`; 1-D stay f32.

    For q4 types the source is first rounded through f16 when `from_f16` (the reference pipeline
    is HF -> f16 file -> quantize tool, models/run_conversions.sh + quantize.cpp:175-181).
    """
    if vocab is None:
        vocab = synthetic_vocab(hp.n_vocab)
    assert len(vocab) == hp.n_vocab
    with open(path, "wb") as f:
        f.write(struct.pack("<I", MAGIC))
        f.write(struct.pack("<7i", hp.n_vocab, hp.n_max_tokens, hp.n_embd, hp.n_intermediate,
                            hp.n_head, hp.n_layer, ftype))
        for tok in vocab:
            f.write(struct.pack("<I", len(tok)))
            f.write(tok)
        for name in tensor_names(hp.n_layer):
            a = np.asarray(weights[name], dtype=np.float32)
            assert a.shape == tensor_shape(name, hp), (name, a.shape)
            two_d = a.ndim == 2
            t = ftype if two_d else FTYPE_F32
            nm = name.encode()
            f.write(struct.pack("<3i", a.ndim, len(nm), t))
            for i in range(a.ndim):
                f.write(struct.pack("<i", a.shape[a.ndim - 1 - i]))
            f.write(nm)
            if two_d:
                src = a
                if t in (FTYPE_Q4_0, FTYPE_Q4_1) and from_f16:
                    src = a.astype(np.float16).astype(np.float32)
                data = _encode_2d(src, t)
                if legacy_q4 and t in (FTYPE_Q4_0, FTYPE_Q4_1):
                    bs = 18 if t == FTYPE_Q4_0 else 20
                    data = to_legacy_q4(np.frombuffer(data, dtype=np.uint8).reshape(-1, bs), t).tobytes()
                f.write(data)
            else:
                f.write(a.tobytes())


@dataclass
class ModelFile:
    hp: BertHParams
    ftype: int
    vocab: List[bytes]
    raw: Dict[str, Tuple[int, Tuple[int, ...], bytes]] = field(default_factory=dict)

    def dequantized(self, name: str) -> np.ndarray:
        t, shp, data = self.raw[name]
        if t == FTYPE_F32:
            return np.frombuffer(data, dtype=np.float32).reshape(shp).copy()
        if t == FTYPE_F16:
            return np.frombuffer(data, dtype=np.float16).astype(np.float32).reshape(shp)
        blk = 18 if t == FTYPE_Q4_0 else 20
        b = np.frombuffer(data, dtype=np.uint8).reshape(shp[0], shp[1] // QK, blk)
        return dequantize_q4_0(b) if t == FTYPE_Q4_0 else dequantize_q4_1(b)


def read_model(path: str) -> ModelFile:
    with open(path, "rb") as f:
        buf = f.read()
    off = 0
    (magic,) = struct.unpack_from("<I", buf, off); off += 4
    assert magic == MAGIC, "bad magic"
    vals = struct.unpack_from("<7i", buf, off); off += 28
    hp = BertHParams(*vals[:6])
    ftype = vals[6]
    vocab = []
    for _ in range(hp.n_vocab):
        (n,) = struct.unpack_from("<I", buf, off); off += 4
        vocab.append(buf[off:off + n]); off += n
    mf = ModelFile(hp, ftype, vocab)
    while off < len(buf):
        n_dims, name_len, t = struct.unpack_from("<3i", buf, off); off += 12
        ne = struct.unpack_from(f"<{n_dims}i", buf, off); off += 4 * n_dims
        name = buf[off:off + name_len].decode(); off += name_len
        shp = tuple(reversed(ne))
        nel = int(np.prod(shp))
        nbytes = {0: nel * 4, 1: nel * 2, 2: nel // QK * 18, 3: nel // QK * 20}[t]
        mf.raw[name] = (t, shp, buf[off:off + nbytes]); off += nbytes
    return mf


def make_synthetic_model(path: str, dims: str | BertHParams, ftype: str | int, seed: int = 0,
                         style: str = "sensitive") -> BertHParams:
    hp = MODEL_DIMS[dims] if isinstance(dims, str) else dims
    ft = FTYPE_BY_NAME[ftype] if isinstance(ftype, str) else ftype
    write_model(path, hp, synthetic_weights(hp, seed, style), ft)
    return hp


def synthetic_token_ids(n_sentences: int, seq_len: int, n_vocab: int, seed: int) -> np.ndarray:
    """[CLS] + (seq_len-2 ids uniform in [1000, n_vocab)) + [SEP]   (SURVEY.md §8d)."""
    rng = np.random.default_rng(seed)
    lo = 1000 if n_vocab > 2000 else min(104, n_vocab - 1)
    ids = rng.integers(lo, n_vocab, size=(n_sentences, seq_len), dtype=np.int32)
    ids[:, 0] = 101 if n_vocab > 102 else 1
    ids[:, -1] = 102 if n_vocab > 102 else 2
    return ids
